#!/usr/bin/env python3
"""bench.py - fMRI volumes/sec (fwd+bwd+AdamW) of ViT3D-base on MI355X, BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = the reference's train step (src/Trainer.py:65-79: forward, CrossEntropyLoss, backward, AdamW)
on one per-GPU batch of synthetic ADNI-shaped volumes already resident in HBM.  Workload = BASELINE.json
configs[1]: ViT3D-base (128^3, patch 16, dim 768, depth 12, heads 12, mlp 3072), bf16 MFMA operands with
fp32 accumulate / master weights, batch 4 per GPU (weak scaling: configs[2] = 32 over 8 GPUs).

Prints ONE JSON line on rank 0 (fields documented in DESIGN.md "Measurement").
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# torch is imported by the RANK processes only (see launch_ranks): the launcher parent never loads it, so it cannot
# create a HIP context that a child would inherit or that an exec would trip over.
torch = None

PEAK_BF16_TFLOPS = 2500.0       # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)
PEAK_F32_TFLOPS = 157.3         # fp32-input MFMA (v_mfma_f32_16x16x4_f32) peak = the fp32 vector rate, same guide ("Matrix cores")
# nv_prof kinds -> the rocprofv3 kernel names of the same launches (profiles/r02_*kernel_stats.csv)
KIND_NAMES = {0: "gemm_ws_kernel<64,128,...,false,false,*> (NT: out-proj, FC2, patch embed)", 1: "gemm_ws_kernel<64,128,...,false,true,*> (NN: dxn1, dxn2, dAO)",
              2: "gemm_ws_kernel<...,true,true,1> (TN: patch-embed weight gradient)", 3: "attn_fwd_res_kernel", 4: "attn_bwd_dq_res_kernel + attn_bwd_dkv_res_kernel",
              5: "gemm_pp_f8_kernel / gemm_pq_kernel<...,true> (fp8 e4m3 operands: qkv, FC1, FC2 of an --fp8 run; priced against the bf16 peak here)",
              10: "gemm_pp_kernel<256,128,4,2,false,false,*> (NT: qkv, FC1)", 11: "gemm_pp_kernel<256,128,4,2,false,true,*> (NN: dU with fused GELU' and bias column sums)",
              12: "gemm_pp_kernel<256,128,4,2,true,true,1> (TN)", 13: "gemm_pp_grouped_tn_kernel (four weight gradients of a layer, auxiliary stream)",
              14: "gemm_pp_grouped_tn_adamw_kernel (the same + AdamW of those weights in the epilogue: 26 B/param of optimizer traffic inside the launch)",
              20: "gemm_pq_kernel<false,false,*> (NT, 256x256 tiles)", 21: "gemm_pq_kernel<false,true,*> (NN, 256x256 tiles)", 22: "gemm_pq_kernel<true,true,*> (TN, 256x256 tiles)"}
GEMM_KINDS = (0, 1, 2, 5, 10, 11, 12, 13, 14, 20, 21, 22)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4, help="volumes per GPU (BASELINE.json configs[1]: 4)")
    ap.add_argument("--preset", default="base", choices=["tiny", "base", "large", "reference"])
    ap.add_argument("--buckets", type=int, default=7,
                    help="gradient all-reduce buckets (7 = two transformer layers each: 28 MB bf16 messages, large enough for RCCL to "
                         "run near its bandwidth, and only the last one (layer 0 + embedding) is exposed after backward)")
    ap.add_argument("--overlap-optimizer", action="store_true", help="AdamW per gradient bucket on the side stream")
    ap.add_argument("--grad-comm", default="bf16", choices=["bf16", "fp32"],
                    help="dtype of the gradient all-reduce messages (N > 1 only; compute and optimizer are unaffected)")
    ap.add_argument("--grad-algo", default=None, choices=["allreduce", "rs_ag", "one_hop"],
                    help="how a gradient bucket is summed over the ranks (N > 1): one all-reduce (default), reduce-scatter + all-gather, or the "
                         "one-hop all-to-all form for a fully connected xGMI node (neurovit_amd/parallel.py::GradSync)")
    ap.add_argument("--dp-path", default="native", choices=["native", "general"],
                    help="N > 1: native = the whole step stays one nv_vit_train_step call with the RCCL all-reduce of every gradient bucket issued from native "
                         "code (falls back to general, with a warning, if the communicator cannot be built); general = Python-driven staged backward + torch.distributed")
    ap.add_argument("--no-probe", action="store_true", help="N > 1: skip the start-up stream-placement probes (streams_beside_collectives): the step "
                                                            "runs on the current stream, the engine picks its auxiliary stream itself")
    ap.add_argument("--dropout", type=float, default=0.0, help="TRAINING_DROPOUT of the timed model (headline: 0, SURVEY 8d)")
    ap.add_argument("--operands", default="bf16", choices=["bf16", "fp16"],
                    help="16-bit MFMA operand format of the timed model (TRAINING_VIT_OPERANDS): bf16 = BASELINE.json's dtype (headline); fp16 = the "
                         "reference's autocast arithmetic (logits within 1e-3 of its fp32 CPU forward), trained with the device-side dynamic loss scale")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary lines (forward-only, fwd+bwd, dropout)")
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--cpu-leg", type=int, default=0, help="internal: run only the CPU baseline's train steps on this many threads and print a JSON value")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--same-data", action="store_true", help="rehearsal: every rank gets rank 0's batch (with --grad-comm fp32 the "
                                                              "averaged gradients, hence the loss curve, must equal the 1-GPU run bit for bit)")
    ap.add_argument("--forward-only", action="store_true", help="time inference forwards (validate path, Trainer.py:101-118) instead of train steps")
    ap.add_argument("--fp8", action="store_true", help="the fp8 (OCP e4m3) path of BASELINE.json configs[4], activation scales calibrated on the bench batch: with "
                                                       "--forward-only the inference forward (all four linears in e4m3); without it the train step whose FORWARD runs "
                                                       "qkv / FC1 / FC2 in e4m3 (bf16 backward; needs --dropout 0)")
    ap.add_argument("--precise", action="store_true", help="with --forward-only: the fp32 inference path (every operand fp32 on the fp32 MFMA: the "
                    "reference's fp32 validate, Trainer.py:101-118; logits within 1e-5 of its CPU forward)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous rehearsal without a GPU: every rank joins the process group, "
                                                            "runs the barrier + max-over-ranks timing plumbing around an empty step and rank 0 "
                                                            "prints the JSON line with value null (tests/test_bench_launcher_cpu.py)")
    return ap.parse_args()


def launch_ranks(a):
    """`python bench.py --gpus N` without a torchrun environment: start N rank processes (one per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set the way `python -m torch.distributed.run` sets them) BEFORE anything in this process touches
    the GPU, relay rank 0's JSON line, exit with the worst return code.  This parent never imports torch."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    log(f"launcher: starting {a.gpus} ranks (127.0.0.1:{port}); parent has torch loaded: {'torch' in sys.modules}")
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread while the parent polls every child: the first rank that fails takes the others down
    # with it (a rank that died before the rendezvous or inside a collective would otherwise leave its peers - and this parent,
    # holding the GPU lease - waiting for the store timeout, or for ever)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        failed = next((i for i, p in enumerate(procs) if p.poll() not in (None, 0)), None)
        if failed is None:
            time.sleep(0.2)
    if failed is None:
        failed = next((i for i, p in enumerate(procs) if p.poll() not in (None, 0)), None)
    if failed is not None:
        log(f"launcher: rank {failed} exited with code {procs[failed].returncode}; stopping the other ranks")
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
    rcs = [p.wait() for p in procs]
    reader.join(timeout=5)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    rc = max((abs(c) for c in rcs), default=0)
    if rc:
        log(f"launcher: rank return codes {rcs}")
    sys.exit(rc)


def make_batch(B, S, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(B, S, S, S, generator=g)
    flat = x.reshape(B, -1)
    x = ((flat - flat.mean(1, keepdim=True)) / (flat.std(1, keepdim=True) + 1e-8)).reshape(B, S, S, S)   # DatasetADNI.py:213
    y = torch.randint(0, 2, (B,), generator=g)
    return x.to(device), y.to(device)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cgroup_cpu_quota():
    """CPU quota of this process's cgroup in cores (cgroup v2 cpu.max, v1 cfs_quota_us / cfs_period_us), or None when unlimited / unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(float(q) / float(per) + 0.5))
        return None
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return max(1, int(q / per + 0.5)) if q > 0 else None
    except (OSError, ValueError):
        return None


def cpu_leg(a):
    """--cpu-leg THREADS: the oracle's train step of the preset on THREADS host threads, CPU only (child of cpu_baseline)."""
    import torch as _t
    from neurovit_amd import config as nvcfg
    from oracle import ref_cpu, train_step
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import weights as W
    _t.set_num_threads(a.cpu_leg)
    size = nvcfg.preset(a.preset)
    S, p = size["TRAINING_VIT_INPUT_SIZE"], size["TRAINING_VIT_PATCH_SIZE"]
    vcfg = dict(image_size=S, image_patch_size=p, frames=S, frame_patch_size=p, num_classes=2, dim=size["TRAINING_VIT_DIM"],
                depth=size["TRAINING_VIT_DEPTH"], heads=size["TRAINING_VIT_HEADS"], mlp_dim=size["TRAINING_VIT_MLP_DIM"], channels=1, dim_head=64)
    cfg = ref_cpu.ViTCfg(**vcfg)
    sd = W.make_tensors(W.vit_param_spec(**vcfg), 3)                  # timing only: random weights of the right shapes
    opt = train_step.AdamW(sd, lr=1e-4, weight_decay=1e-2)
    g = _t.Generator().manual_seed(4244)
    x = _t.randn(a.batch, S, S, S, generator=g)
    y = _t.randint(0, 2, (a.batch,), generator=g)
    video = ref_cpu.fmri_to_video(x)
    train_step.train_step(sd, cfg, opt, video, y)
    t0 = time.perf_counter()
    for _ in range(a.cpu_steps):
        train_step.train_step(sd, cfg, opt, video, y)
    print(json.dumps({"value": a.batch * a.cpu_steps / (time.perf_counter() - t0), "unit": "volumes/s", "cores": _t.get_num_threads()}), flush=True)


def cpu_baseline(model, vcfg, B, S, steps, warmup=3, preset="base"):
    """The oracle's restatement of the same train step (eager PyTorch CPU fp32 = the reference's own CPU path,
    SURVEY.md 8d), timed on this host's cores on a bounded sample: `warmup` + `steps` steps of the bench workload, plus
    BASELINE.json configs[0] (ViT3D tiny, batch 2: the reference's own CPU-runnable case) as a second line."""
    from oracle import ref_cpu, train_step
    try:
        nthreads = len(os.sched_getaffinity(0))
    except AttributeError:
        nthreads = os.cpu_count() or 1
    visible = nthreads
    usable = min(visible, _cgroup_cpu_quota() or visible)      # cores this process may really use: affinity AND the cgroup's CPU quota
    nthreads = max(1, min(usable, 16))         # the GPU box grants a 16-core share per GPU: more threads than that only oversubscribe it
    torch.set_num_threads(nthreads)

    def run(sd, cfg, batch, n_warm, n_timed, seed):
        opt = train_step.AdamW(sd, lr=1e-4, weight_decay=1e-2)
        x, y = make_batch(batch, cfg.image_size, "cpu", seed)
        video = ref_cpu.fmri_to_video(x)
        for _ in range(n_warm):
            train_step.train_step(sd, cfg, opt, video, y)
        t0 = time.perf_counter()
        for _ in range(n_timed):
            train_step.train_step(sd, cfg, opt, video, y)
        return batch * n_timed / (time.perf_counter() - t0)

    sd = {k[len("volume_encoder.vit3d."):]: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    value = run(sd, ref_cpu.ViTCfg(**vcfg), B, warmup, steps, 4242)
    # the same sample on EVERY core the process can see (north_star: "the node's host cores"): more threads than the box's share of the
    # host may oversubscribe it, so both figures are reported and `value` stays the one measured inside the share
    # tiny: random weights of the right shapes (timing only)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import weights as W
    tcfg = ref_cpu.ViTCfg(**W.TINY)
    tsd = W.make_tensors(W.vit_param_spec(**W.TINY), 3)
    tiny = run(tsd, tcfg, 2, warmup, max(steps, 20), 4243)
    all_cores = {"value": None, "cores": usable, "cores_visible": visible,
                 "note": f"not run: the process may use {usable} cores (affinity {visible}, cgroup CPU quota {_cgroup_cpu_quota()}), which `value` already uses"}
    if usable > nthreads:
        # In a CHILD process with a wall-clock limit: threads beyond the real share of the host - a quota this process cannot see - make
        # eager PyTorch crawl (measured: minutes per step at 256 threads on a 16-core share); the child never touches the GPU
        n_all = max(2, steps // 2)
        limit = 90
        log(f"cpu baseline: all-cores leg on {usable} threads (child process, {limit} s limit)")
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-leg", str(usable), "--batch", str(B), "--preset", preset, "--cpu-steps", str(n_all)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit, env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode == 0 and line:
                all_cores = dict(json.loads(line[-1]), cores_visible=visible, sample=f"{n_all} train steps after 1 warm-up step, same workload, child process")
            else:
                all_cores["note"] = f"child failed (rc {r.returncode}): {r.stderr.strip()[-200:]}"
        except subprocess.TimeoutExpired:
            all_cores["note"] = (f"abandoned after {limit} s: {usable} threads oversubscribe this process's share of the host "
                                 f"({nthreads} threads: {value:.2f} volumes/s)")
    return {"value": value, "unit": "volumes/s", "cores": torch.get_num_threads(), "cores_visible": visible,
            "cores_note": "threads used = min(cores this process may use [affinity and cgroup CPU quota], 16: a one-GPU box's CPU share); all_cores = the same "
                          "sample on every usable core (probed with one step first, abandoned when that step shows the share is oversubscribed)",
            "all_cores": all_cores, "cpu_model": _cpu_model(), "kind": "port",
            "sample": f"{steps} train steps (fwd+bwd+AdamW, fp32 eager PyTorch CPU) of the same workload, batch {B}, after {warmup} warm-up steps",
            "tiny_config": {"value": tiny, "unit": "volumes/s",
                            "workload": "BASELINE.json configs[0]: ViT3D tiny 64^3 patch 16 dim 192 depth 4 heads 3 mlp 384, batch 2, train step",
                            "sample": f"{max(steps, 20)} steps after {warmup} warm-up steps"}}


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def dry_run(a, world, rank):
    """--dry-run: the N-rank plumbing of the timed region (process group, barrier, max over ranks, one JSON line) with an
    empty step - runs without a GPU (gloo)."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(a.backend if a.backend != "nccl" else "gloo", rank=rank, world_size=world)
    n = dist.get_world_size() if world > 1 else 1
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pass
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    if rank == 0:
        print(json.dumps({"metric": "dry run (launcher rehearsal, no GPU work)", "value": None, "unit": "volumes/s", "n_gpus": n,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / max(a.steps, 1) * 1e3, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "none",
                          "config": {"workload": "none", "global_batch": a.batch * n, "parallelism": f"dp{n}"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def forward_only_bench(a, model, x, size, S, p, B, world, rank, device, dist):
    """--forward-only: inference forwards of the preset (bf16, or fp8 with --fp8) at batch B per GPU, same timing contract."""
    from neurovit_amd.engine import flops_forward, make_config
    model.eval()
    vit = model.volume_encoder.vit3d
    if a.fp8 and a.precise:
        raise SystemExit("--fp8 and --precise are different arithmetic modes: choose one")
    if a.fp8:
        with torch.no_grad():
            vit.enable_fp8(x.permute(0, 3, 1, 2).unsqueeze(1))
    if a.precise:
        vit.eval_precision = "fp32"

    def fwd():
        with torch.no_grad():
            return model(x)

    for _ in range(max(a.warmup, 2)):
        fwd()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = fwd()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    vcfg = dict(image_size=S, image_patch_size=p, frames=S, frame_patch_size=p, num_classes=2, dim=size["TRAINING_VIT_DIM"],
                depth=size["TRAINING_VIT_DEPTH"], heads=size["TRAINING_VIT_HEADS"], mlp_dim=size["TRAINING_VIT_MLP_DIM"], channels=1, dim_head=64)
    f_fwd = flops_forward(make_config(**vcfg))
    value = B * world * a.steps / elapsed
    # fp8 runs are priced against the dense fp8 peak for the linears they run in fp8 and the bf16 peak for the rest: reported as
    # the fraction of the bf16 peak (conservative, one number) and, separately, of the fp8 peak
    out_line = {"metric": f"fMRI volumes/sec (forward only) ViT3D {S}^3 p{p} d{size['TRAINING_VIT_DIM']} L{size['TRAINING_VIT_DEPTH']}",
                "value": round(value, 2), "unit": "volumes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "fp8 (e4m3 qkv/FC1/FC2 operands, bf16 elsewhere, fp32 accumulate)" if a.fp8 else ("f32 (fp32 MFMA, every operand fp32)" if a.precise else a.operands),
                "data": "synthetic",
                "config": {"workload": f"ViT3D-{a.preset} {S}^3 patch {p}, inference forward, batch {B}/GPU", "global_batch": B * world, "parallelism": f"dp{world}"},
                "mfma_frac_bf16_peak": round(value / world * f_fwd / (PEAK_BF16_TFLOPS * 1e12), 4),
                "mfma_frac_fp8_peak": round(value / world * f_fwd / (2 * PEAK_BF16_TFLOPS * 1e12), 4) if a.fp8 else None,
                "mfma_frac_fp32_peak": round(value / world * f_fwd / (PEAK_F32_TFLOPS * 1e12), 4) if a.precise else None,
                "logits_finite": bool(torch.isfinite(out).all())}
    if rank == 0:
        print(json.dumps(out_line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    global torch
    a = parse()
    if a.cpu_leg:
        return cpu_leg(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a)                                     # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        log(f"--gpus {a.gpus} does not match WORLD_SIZE={world} of the launch environment")
        sys.exit(2)
    import torch as _torch
    torch = _torch
    import torch.distributed as dist
    if a.dry_run:
        return dry_run(a, world, rank)
    if a.no_probe:
        os.environ["NEUROVIT_DP_PROBE"] = "0"
    rccl_log = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC - without it RCCL's peer-buffer exchange (and any
        # CUDA-tensor sharing across processes) fails with `hipIpcGetMemHandle: invalid argument`.  Set here too (not only by launch_ranks)
        # because the driver starts the ranks through torch.distributed.run
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rank == 0 and a.backend == "nccl" and "NCCL_DEBUG" not in os.environ:
            # rank 0 relays what RCCL chose (algorithm / protocol / channels / transport) to stderr after the run: the scaling record
            # then shows whether the all-reduce ran as a ring over one xGMI link or used the fully connected topology
            import tempfile
            rccl_log = os.path.join(tempfile.gettempdir(), f"rccl_rank0_{os.getpid()}.log")
            os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH,TUNING", NCCL_DEBUG_FILE=rccl_log)
        if a.same_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
        world = dist.get_world_size()                       # n_gpus of the JSON line = what the process group reports
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from neurovit_amd import config as nvcfg
    from neurovit_amd._cabi import lib, require_gpu
    from neurovit_amd.NeuroEncoder import NeuroEncoder
    from neurovit_amd.trainer import TrainStep
    require_gpu()

    size = nvcfg.preset(a.preset)
    S, p = size["TRAINING_VIT_INPUT_SIZE"], size["TRAINING_VIT_PATCH_SIZE"]
    config = dict(DEVICE=str(device), TRAINING_DIM=3, TRAINING_DROPOUT=a.dropout, GRADCAM_CUBE_SIZE=8, DATASET_NAME="adni",
                  TRAINING_LEARNING_RATE=1e-4, TRAINING_WEIGHT_DECAY=1e-2, TRAINING_VIT_OPERANDS=a.operands, **size)
    torch.manual_seed(42)                                   # main.py:86-88
    model = NeuroEncoder(config)
    model.train()
    step = TrainStep(model, process_group=None, n_buckets=a.buckets, overlap_optimizer=a.overlap_optimizer,
                     grad_comm_dtype=torch.bfloat16 if a.grad_comm == "bf16" else torch.float32, grad_comm_algo=a.grad_algo,
                     native_dp=(a.dp_path == "native") if world > 1 else None)
    B = a.batch
    x, y = make_batch(B, S, device, 42 + (0 if a.same_data else rank))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if a.forward_only:
        return forward_only_bench(a, model, x, size, S, p, B, world, rank, device, dist)
    if a.fp8:
        # BASELINE.json configs[4] ("fp8 MFMA", quoted fwd / fwd+bwd): training forwards with qkv / FC1 / FC2 on e4m3 operands, bf16 backward;
        # activation scales calibrated on the bench batch, weights re-quantised (in place) after every optimizer step - inside the timed step
        with torch.no_grad():
            model.volume_encoder.vit3d.enable_fp8(x.permute(0, 3, 1, 2).unsqueeze(1), out_proj=False, training=True)
    log(f"model built on {device}, warm-up {a.warmup} steps")
    for _ in range(a.warmup):
        step(x, y)
    barrier()
    log("timed region")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step(x, y)
    enqueued = time.perf_counter() - t0          # host time to enqueue the K steps: far below `elapsed` = the GPU is the limiter
    barrier()
    elapsed = time.perf_counter() - t0
    # host cost of enqueueing ONE step, measured from an idle queue (over the K timed steps the host runs into the runtime's queue
    # depth and is throttled to the GPU's pace, so `enqueued` only bounds it from above)
    t1 = time.perf_counter()
    step(x, y)
    host_step_ms = (time.perf_counter() - t1) * 1e3
    torch.cuda.synchronize()
    log(f"rank {rank}: step path: {step.last_path} (native = nv_vit_train_step; native-dp = the same call with RCCL all-reduce per bucket from native code"
        f"{' ' + str(getattr(step, 'last_dp', '')) if step.last_path == 'native-dp' else ''}; general = autograd-driven stages), AdamW placement fuse_update = {step.last_fuse_update}, "
        f"loss scale: {'dynamic (device-side GradScaler)' if step.scaler is not None else (step.static_scale or 'none')}")
    log(f"host enqueue {host_step_ms:.3f} ms for one step from an idle queue ({enqueued / a.steps * 1e3:.3f} ms/step inside the timed loop) "
        f"of {elapsed / a.steps * 1e3:.3f} ms/step")
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    ms = elapsed / a.steps * 1e3
    value = B * world * a.steps / elapsed

    # ---- secondary lines SURVEY 8(d) asks for (single rank; not the headline): forward-only (validate, Trainer.py:101-118),
    #      fwd+bwd without the optimizer, and the full step with the config's default dropout 0.1
    also = None
    if world == 1 and not a.no_extras:
        def timed(fn, n):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return B * n / (time.perf_counter() - t)

        crit = step.criterion

        def fwd_bwd():
            step.optimizer.zero_grad(set_to_none=True)
            crit(model(x), y).backward()

        def fwd_only():
            with torch.no_grad():
                model(x)

        also = {"fwd_bwd_no_optimizer_volumes_s": round(timed(fwd_bwd, a.steps), 1)}
        model.eval()
        also["forward_only_eval_volumes_s"] = round(timed(fwd_only, a.steps), 1)
        with model.precision("fp32"):           # the reference's validate arithmetic (Trainer.py:101-118): fp32 MFMA path
            also["forward_only_eval_fp32_volumes_s"] = round(timed(fwd_only, max(3, a.steps // 3)), 1)
        model.train()
        torch.manual_seed(7)
        dcfg = dict(config, TRAINING_DROPOUT=0.1)
        dmodel = NeuroEncoder(dcfg)
        dmodel.train()
        dstep = TrainStep(dmodel, process_group=None, n_buckets=a.buckets)
        also["train_step_dropout_0.1_volumes_s"] = round(timed(lambda: dstep(x, y), a.steps), 1)
        del dstep, dmodel
        torch.cuda.empty_cache()
        if a.operands == "bf16" and not a.fp8:
            # the same model on fp16 MFMA operands (the reference's autocast arithmetic, Trainer.py:68; the path that holds north_star's 1e-3
            # against the reference's fp32 CPU forward: tests/test_fp16_gpu.py): forward, and the train step with the dynamic loss scale
            # (GradScaler semantics on the device: every gradient checked, update gated - so AdamW runs behind the backward pass) and with a static one
            torch.manual_seed(42)
            hmodel = NeuroEncoder(dict(config, TRAINING_VIT_OPERANDS="fp16"))
            hmodel.train()
            hstep = TrainStep(hmodel, process_group=None, n_buckets=a.buckets)
            fp16 = {"train_step_dynamic_loss_scale_volumes_s": round(timed(lambda: hstep(x, y), a.steps), 1),
                    "loss_scale_after": hstep.scaler.get_scale(), "updates_applied": hstep.scaler.steps_applied(), "updates_skipped": hstep.scaler.steps_skipped()}
            hstep2 = TrainStep(hmodel, process_group=None, n_buckets=a.buckets, loss_scale=4096.0)
            fp16["train_step_static_loss_scale_volumes_s"] = round(timed(lambda: hstep2(x, y), a.steps), 1)
            hmodel.eval()

            def hfwd():
                with torch.no_grad():
                    hmodel(x)
            fp16["forward_only_eval_volumes_s"] = round(timed(hfwd, a.steps), 1)
            also["fp16_operands"] = fp16
            del hstep, hstep2, hmodel
            torch.cuda.empty_cache()
        # BASELINE.json configs[3] shape: one 4D sample = T = 20 volumes through the (frozen) encoder, forward only
        x20 = make_batch(20, S, device, 77)[0]
        model.eval()

        def fwd20():
            with torch.no_grad():
                model(x20)

        for _ in range(2):
            fwd20()
        torch.cuda.synchronize()
        t20 = time.perf_counter()
        for _ in range(10):
            fwd20()
        torch.cuda.synchronize()
        also["forward_only_batch20_volumes_s"] = round(20 * 10 / (time.perf_counter() - t20), 1)
        del x20
        model.train()
        # BASELINE.json configs[3] as a train step: the 4D NeuroEncoder (this encoder frozen, loaded from a checkpoint as
        # NeuroEncoder.py:23-36 does; native temporal head trained), one sample x T = 20 per micro-step, accumulation 4 (config4D.yaml)
        if a.preset == "base":
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                torch.save(model.state_dict(), os.path.join(td, "best3d.pth"))
                m4 = NeuroEncoder(dict(config, TRAINING_DIM=4, GLOBAL_BASE_PATH=td, BEST_MODEL_PATH="best3d.pth"))
            m4.train(); m4.volume_encoder.eval()
            step4 = TrainStep(m4, process_group=None, accumulation_steps=4)
            x4 = torch.randn(1, S, S, S, 20, device=device)
            y4 = torch.zeros(1, dtype=torch.long, device=device)
            for _ in range(4):
                step4(x4, y4)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            for _ in range(12):
                step4(x4, y4)
            torch.cuda.synchronize()
            also["neuro4d_train_microstep_T20_volumes_s"] = round(20 * 12 / (time.perf_counter() - t4), 1)
            del step4, m4, x4
            torch.cuda.empty_cache()
        log(f"extras: {also}")

    log(f"{ms:.3f} ms/step, {value:.1f} volumes/s; roofline leg")
    # ---- roofline leg: per-launch hipEvent durations of the dominant kernel family, same steps, same streams
    prof_steps = 3

    def prof_leg():
        lib.nv_prof_enable(1)
        for _ in range(prof_steps):
            step(x, y)
        torch.cuda.synchronize()
        kinds = {}
        for k in KIND_NAMES:
            msk, wk, ck, bk = ctypes.c_double(), ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
            lib.nv_prof_summary(k, ctypes.byref(msk), ctypes.byref(wk), ctypes.byref(ck))
            lib.nv_prof_summary_bytes(k, ctypes.byref(bk))
            if ck.value:
                kinds[k] = dict(ms=msk.value, flops=wk.value, launches=ck.value, bytes=bk.value)
        lib.nv_prof_enable(0)
        gemm = [kinds[k] for k in GEMM_KINDS if k in kinds]
        g_ms, g_fl, g_n = sum(k["ms"] for k in gemm), sum(k["flops"] for k in gemm), sum(k["launches"] for k in gemm)
        return kinds, g_ms, g_fl, g_n

    # (1) as timed: weight-gradient GEMMs run on the auxiliary stream beside the main stream's kernels, so each launch's
    #     event-bracketed duration includes sharing the chip with the other stream (this is what rocprofv3 of this command shows)
    kinds, g_ms, g_fl, g_n = prof_leg()
    achieved = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
    # (2) the same kernels alone on the chip (engine run with a single stream): the kernel-quality figure
    rt = model.volume_encoder.vit3d._rt
    was = rt.use_aux_stream
    rt.use_aux_stream = False
    step(x, y)
    _, s_ms, s_fl, _ = prof_leg()
    rt.use_aux_stream = was
    achieved_serial = s_fl / (s_ms * 1e-3) / 1e12 if s_ms > 0 else 0.0
    # (3) the step as timed updates the layers' Linear weights DURING the backward pass (TrainStep.fuse_update: per-layer AdamW launches on
    #     the auxiliary stream, or the update inside the weight-gradient GEMMs): the GEMMs then share the chip with that HBM-bound
    #     work and their event-bracketed durations grow although the step is faster.  The same family with the whole update as
    #     one launch behind the backward pass (the definition of rounds 1-3), for comparison:
    achieved_unfused = None
    fused_was = getattr(step, "fuse_update", 0)
    fuse_mode = getattr(step, "last_fuse_update", 0)
    if fuse_mode:
        step.fuse_update = 0
        step(x, y)
        _, u_ms, u_fl, _ = prof_leg()
        step.fuse_update = fused_was
        step(x, y)
        achieved_unfused = u_fl / (u_ms * 1e-3) / 1e12 if u_ms > 0 else None
    # HBM-side check.  PMC counters cannot be read in-process: they come from rocprofv3 --pmc passes of THIS command on an MI355X
    # (tools/pmc_traffic_summary.py -> profiles/rNN_pmc_traffic.json, per nv_prof kind); the algorithmic bytes beside them are
    # counted live, per launch, by the launchers (operands read once + outputs written once).  traffic / algorithmic = the waste.
    pmc, traffic_src = {}, None
    for tag in ("r05", "r04", "r03", "r02", "r01"):
        tfile = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")
        if os.path.exists(tfile):
            tj = json.load(open(tfile))
            pmc, traffic_src = tj.get("by_kind", {}), tj["source"]
            if not pmc:                          # round-1/2 file: one family-wide figure only
                pmc = {"family": {"fetch_MB_per_launch": tj["fetch_MB_per_launch"], "write_MB_per_launch": tj["write_MB_per_launch"]}}
            break

    # matrix-pipe utilisation of the same launches, ALONE on the chip: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) from a rocprofv3 --pmc
    # pass of this command (tools/r05_pmc.sh -> tools/pmc_sq_summary.py; the profiler serialises dispatches under --pmc)
    sq, sq_src = {}, None
    for tag in ("r05",):
        sfile = os.path.join(ROOT, "profiles", f"{tag}_pmc_sq.json")
        if os.path.exists(sfile):
            sj = json.load(open(sfile))
            sq, sq_src = sj.get("by_kind", {}), sj["source"]

    def kind_traffic_mb(k):
        d = pmc.get(str(k))
        return None if d is None else d["fetch_MB_per_launch"] + d["write_MB_per_launch"]

    by_kernel = {}
    for k, v in kinds.items():
        algo_mb = v["bytes"] / v["launches"] / 1e6 if v.get("bytes") else None
        t_mb = kind_traffic_mb(k)
        by_kernel[KIND_NAMES[k]] = {"TFLOP/s": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1), "avg_us": round(v["ms"] * 1e3 / v["launches"], 2),
                                    "launches_per_step": v["launches"] // prof_steps,
                                    "algorithmic_MB": None if algo_mb is None else round(algo_mb, 2),
                                    "traffic_MB": None if t_mb is None else round(t_mb, 2),
                                    "mfma_busy": sq.get(str(k), {}).get("mfma_busy"),
                                    "traffic_over_algorithmic": None if (algo_mb is None or t_mb is None) else round(t_mb / algo_mb, 2)}
    # family figure over exactly the launches `achieved` is computed from: per-kind PMC traffic weighted by this run's launch counts
    gk = [k for k in GEMM_KINDS if k in kinds]
    if gk and all(kind_traffic_mb(k) is not None for k in gk):
        traffic = sum(kind_traffic_mb(k) * kinds[k]["launches"] for k in gk) / g_n * 1e6
    elif "family" in pmc:
        traffic = (pmc["family"]["fetch_MB_per_launch"] + pmc["family"]["write_MB_per_launch"]) * 1e6
    else:
        traffic = None
    g_bytes = sum(kinds[k].get("bytes", 0.0) for k in gk)
    roofline = {"bound": "mfma", "kernel": "bf16 MFMA GEMM family: gemm_pp_kernel / gemm_pp_grouped_tn_kernel (256x128 tiles) + gemm_ws_kernel (64x128 tiles), all fused epilogues",
                "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                "mfma_busy_source": sq_src,
                "in_step_stretch": "profiles/r05_step_timeline_*.txt: the N = 768 data-gradient GEMM takes 18.8 us alone, 19.5 us beside the weight-gradient GEMMs, 32.3 us beside "
                                   "them AND the per-layer AdamW (HBM / fabric contention, not CU starvation); ln_bwd 9.8 us alone, 22.4 us beside the weight-gradient GEMMs (CU sharing)",
                "traffic": traffic, "traffic_unit": "bytes/launch of FABRIC traffic (L2 memory-side read + write requests, rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE: Infinity-Cache hits are counted, so this is an upper bound of the HBM bytes, not the HBM bytes; none of the 688 counters `rocprofv3 -L` lists on the MI355X box is a memory-side-cache or memory-controller counter, so the split cannot be measured), mean over the launches of `achieved`", "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(g_bytes / max(g_n, 1), 1),
                "traffic_over_algorithmic": None if not (traffic and g_bytes) else round(traffic / (g_bytes / g_n), 2),
                "avg_launch_us": round(g_ms * 1e3 / max(g_n, 1), 2), "launches_per_step": g_n // prof_steps,
                "achieved_single_stream": round(achieved_serial, 2), "frac_single_stream": round(achieved_serial / PEAK_BF16_TFLOPS, 4),
                "achieved_update_unfused": None if achieved_unfused is None else round(achieved_unfused, 2),
                "frac_update_unfused": None if achieved_unfused is None else round(achieved_unfused / PEAK_BF16_TFLOPS, 4),
                "note": "achieved = per-launch hipEvent durations inside the concurrent two-stream step (agrees with rocprofv3 of this command); in this step AdamW of the "
                        "layers' weights runs during the backward pass (config.adamw), sharing the chip with these launches; "
                        "achieved_update_unfused = the same family when the whole update is one launch behind the backward pass (rounds 1-3's definition; that step is "
                        "2.8-3.7 % slower at batch 4, profiles/r04_adamw_in_wgrad_epilogue.log); "
                        "achieved_single_stream = same kernels, same shapes, engine run on one stream (kernels alone on the chip); traffic = PMC "
                        "counters of a rocprofv3 run of this command (not readable in-process), algorithmic bytes counted live per launch",
                "by_kernel": by_kernel}
    executed_flops_step = sum(v["flops"] for v in kinds.values()) / prof_steps       # what the MFMA kernels of a step really execute

    from neurovit_amd.engine import flops_forward, make_config
    vcfg = dict(image_size=S, image_patch_size=p, frames=S, frame_patch_size=p, num_classes=2, dim=size["TRAINING_VIT_DIM"],
                depth=size["TRAINING_VIT_DEPTH"], heads=size["TRAINING_VIT_HEADS"], mlp_dim=size["TRAINING_VIT_MLP_DIM"], channels=1, dim_head=64)
    f_fwd = flops_forward(make_config(**vcfg))
    f_step = 3.0 * f_fwd                                    # fwd + bwd = 3 x fwd algorithmic FLOPs

    # BASELINE.json's metric is "fMRI volumes/sec (fwd+bwd) ViT3D 128^3 p16 d768 L12"; the timed step is the reference's whole train
    # step (Trainer.py:65-79), i.e. it also contains the AdamW update - said in the string so the number is not read as fwd+bwd only
    metric = (f"fMRI volumes/sec (fwd+bwd) ViT3D {S}^3 p{p} d{size['TRAINING_VIT_DIM']} L{size['TRAINING_VIT_DEPTH']}"
              " [timed step = fwd+bwd+AdamW update]")
    out = {"metric": metric, "value": round(value, 2), "unit": "volumes/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None,
           "dtype": "fp8 forward (e4m3 qkv/FC1/FC2 operands, fp32 accumulate), bf16 backward" if a.fp8 else a.operands, "data": "synthetic",
           "config": {"workload": f"ViT3D-{a.preset} {S}^3 patch {p}, train step (fwd+bwd+AdamW), batch {B}/GPU, dropout {a.dropout:g}",
                      "global_batch": B * world, "parallelism": f"dp{world}", "grad_buckets": a.buckets,
                      "grad_allreduce": ("none (1 GPU)" if world == 1 else f"{a.grad_comm} messages, sum, overlapped with backward"),
                      "step_path": step.last_path, "dp": getattr(step, "last_dp", None),
                      "adamw": ({1: "layers' Linear weights updated in their weight-gradient GEMM epilogues, the rest in one launch",
                                 2: "layers' Linear weights updated in their weight-gradient GEMM epilogues (gradients kept), the rest in one launch",
                                 3: "layers' Linear weights updated per layer on the auxiliary stream behind their weight-gradient GEMMs, the rest in one launch"}.get(
                                     fuse_mode, "one launch over the arena behind the backward pass" if world == 1 else "per arena range behind the gradient all-reduce"))},
           "mfma_frac_step": round(value / world * f_step / (PEAK_BF16_TFLOPS * 1e12), 4),
           # the last block runs on its B cls rows (pool='cls': the other rows never reach the head; tests prove identical logits and
           # gradients), so ~5 % of the ALGORITHMIC FLOPs above are not executed: this is the fraction over the FLOPs the step's MFMA
           # kernels really ran (GEMM + attention launches as counted by the per-launch profiler; skinny cls-row kernels excluded)
           "mfma_frac_step_executed": round(executed_flops_step / (ms * 1e-3) / (PEAK_BF16_TFLOPS * 1e12), 4),
           "executed_over_algorithmic_flops": round(executed_flops_step / (B * f_step), 4),
           "host_enqueue_ms_per_step": round(host_step_ms, 3),
           "loss": round(float(loss), 5), "roofline": roofline}
    if also is not None:
        also["forward_only_mfma_frac"] = round(also["forward_only_eval_volumes_s"] * f_fwd / (PEAK_BF16_TFLOPS * 1e12), 4)
        out["also"] = also
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        log("cpu baseline leg")
        out["cpu_baseline"] = cpu_baseline(model, vcfg, B, S, a.cpu_steps, preset=a.preset)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if rccl_log and os.path.exists(rccl_log):
        relay_rccl_choices(rccl_log)


def relay_rccl_choices(path, limit=40):
    """stderr summary of rank 0's RCCL INFO log: topology / algorithm / protocol / channel lines, deduplicated."""
    import re
    keep = re.compile(r"(Channel \d+/\d+ *:|\d+ coll channels|nChannels|Trees|Ring \d+ *:|Algo|algorithm|protocol|Proto|via P2P|via SHM|via NET|XGMI|xgmi|"
                      r"comm 0x[0-9a-f]+ rank|Connected all (rings|trees)|threadThresholds|Using tuner|NCCL_ALGO|NCCL_PROTO)")
    seen, n = set(), 0
    try:
        for line in open(path, errors="replace"):
            if not keep.search(line):
                continue
            body = re.sub(r"^.*?NCCL INFO ", "", line.strip())
            key = re.sub(r"\d+", "#", body)
            if key in seen:
                continue
            seen.add(key)
            log(f"rccl: {body[:200]}")
            n += 1
            if n >= limit:
                log("rccl: ... (more in the NCCL_DEBUG_FILE)")
                break
        if n == 0:
            log(f"rccl: no topology / algorithm lines in {path}")
    except OSError as e:
        log(f"rccl: could not read {path}: {e}")


if __name__ == "__main__":
    main()
