"""bench.py --gpus N must start N ranks itself: the parent spawns the rank processes before anything
touches the GPU and never imports torch; n_gpus of the JSON line is the world size the process group reports."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=e)


def test_gpus2_launches_two_ranks_and_parent_stays_torch_free():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "0", "--backend", "gloo", "--dry-run"])
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                        # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3
    assert "parent has torch loaded: False" in r.stderr


def test_gpus_flag_must_match_torchrun_world_size():
    r = _run(["--gpus", "4", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0
    assert "does not match WORLD_SIZE" in r.stderr


def test_default_is_one_rank_without_launcher():
    r = _run(["--dry-run", "--steps", "1"])
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    assert "launcher" not in r.stderr


def test_driver_launch_form_eight_ranks_under_torchrun():
    """The driver's N > 1 command line: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 8 ...` - every rank reads RANK / WORLD_SIZE / MASTER_* from the environment, joins the group
    (gloo here: no GPU), runs the barrier + max-over-ranks timing plumbing; exactly one JSON line, n_gpus = 8."""
    port = 29900 + (os.getpid() % 90)
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    e["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--backend", "gloo", "--dry-run"], capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
