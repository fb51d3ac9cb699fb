#!/usr/bin/env python3
"""Single-GPU rehearsal of the RCCL path bench.py takes at N > 1: a real "nccl" process group of size 1 (so every collective
goes through RCCL kernels and torch's stream semantics), with the data-parallel machinery forced on as if world_size were 2:
bucketed bf16 / fp32 all-reduce on the side stream during the staged backward, barrier, broadcast.  With one rank the sum
is the identity, so the loss curve must equal the plain single-process run (1/world scaling forced back to 1)."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")

from neurovit_amd import config as nvcfg                   # noqa: E402
from neurovit_amd.NeuroEncoder import NeuroEncoder         # noqa: E402
from neurovit_amd.parallel import GradSync, broadcast_parameters   # noqa: E402
from neurovit_amd.trainer import TrainStep                 # noqa: E402


def build():
    size = nvcfg.preset("base")
    cfg = dict(DEVICE="cuda:0", TRAINING_DIM=3, TRAINING_DROPOUT=0.0, GRADCAM_CUBE_SIZE=8, DATASET_NAME="adni",
               TRAINING_LEARNING_RATE=1e-4, TRAINING_WEIGHT_DECAY=1e-2, **size)
    torch.manual_seed(42)
    m = NeuroEncoder(cfg)
    m.train()
    return m


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    x = torch.randn(4, 128, 128, 128, device="cuda")
    y = torch.tensor([0, 1, 1, 0], device="cuda")
    timed = int(os.environ.get("REHEARSAL_TIMED_STEPS", "20"))
    ref = TrainStep(build())
    losses_ref = [float(ref(x, y)) for _ in range(4)]
    for dtype in (torch.float32, torch.bfloat16):
        model = build()
        step = TrainStep(model)
        step.sync = GradSync(None, n_buckets=int(os.environ.get("REHEARSAL_BUCKETS", "7")), comm_dtype=dtype)
        step.sync.world = 2                       # force the collective path; the group itself has one rank
        step.sync.write_back = dtype != torch.bfloat16     # as TrainStep sets it: bf16 -> the fused AdamW reads the reduced buffer
        step.world = 1                            # keep the 1/world scaling of the optimizer at 1
        arena, _ = model.volume_encoder.vit3d.flat_parameters()
        broadcast_parameters(arena)
        dist.barrier()
        losses = [float(step(x, y)) for _ in range(4)]
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(timed):
            step(x, y)
        host_ms = (time.perf_counter() - t) / timed * 1e3
        dist.barrier()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / timed * 1e3
        print(f"{dtype}: host enqueue {host_ms:.3f} ms/step; losses {['%.5f' % v for v in losses]} (reference {['%.5f' % v for v in losses_ref]}), {ms:.3f} ms/step, "
              f"{step.sync.bytes_reduced / (4 + timed) / 1e6:.0f} MB reduced per step")
        tol = 0 if dtype == torch.float32 else 5e-2
        assert all(abs(a - b) <= tol * max(1.0, abs(b)) for a, b in zip(losses, losses_ref)), "loss curve differs from the single-process run"
    # the NATIVE data-parallel step (nv_vit_train_step + nv_dp_plan): a one-rank communicator of the library's own, collectives issued from native code
    for msgs, per_bucket in ((torch.float32, "1"), (torch.bfloat16, "1"), (torch.float32, "0")):
        os.environ["NEUROVIT_DP_UPDATE_PER_BUCKET"] = per_bucket
        model = build()
        step = TrainStep(model, n_buckets=int(os.environ.get("REHEARSAL_BUCKETS", "7")), native_dp=True, grad_comm_dtype=msgs)
        losses = [float(step(x, y)) for _ in range(4)]
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(timed):
            step(x, y)
        host_ms = (time.perf_counter() - t) / timed * 1e3
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / timed * 1e3
        print(f"native-dp {msgs} update_per_bucket={per_bucket}: path {step.last_path} {step.last_dp}; host enqueue {host_ms:.3f} ms/step; {ms:.3f} ms/step; losses {['%.5f' % v for v in losses]}")
        tol = 0 if msgs == torch.float32 else 5e-2
        assert all(abs(a - b) <= tol * max(1.0, abs(b)) for a, b in zip(losses, losses_ref)), "loss curve differs from the single-process run"
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(timed):
        ref(x, y)
    torch.cuda.synchronize()
    print(f"single-process native step: {(time.perf_counter() - t) / timed * 1e3:.3f} ms/step (fuse_update {ref.last_fuse_update})")
    dist.destroy_process_group()
    print("rccl rehearsal ok")


if __name__ == "__main__":
    main()
