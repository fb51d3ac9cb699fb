#!/usr/bin/env python3
"""One configuration of the native data-parallel step per process (one-rank RCCL communicator): ms/step against the single-process native
step on the same box.  usage: dp_native_probe.py {fp32|bf16} {0|1: update_per_bucket} [buckets]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurovit_amd import config as nvcfg                   # noqa: E402
from neurovit_amd.NeuroEncoder import NeuroEncoder         # noqa: E402
from neurovit_amd.trainer import TrainStep                 # noqa: E402


def build():
    size = nvcfg.preset("base")
    cfg = dict(DEVICE="cuda:0", TRAINING_DIM=3, TRAINING_DROPOUT=0.0, GRADCAM_CUBE_SIZE=8, DATASET_NAME="adni",
               TRAINING_LEARNING_RATE=1e-4, TRAINING_WEIGHT_DECAY=1e-2, **size)
    torch.manual_seed(42)
    m = NeuroEncoder(cfg)
    m.train()
    return m


def timed(step, x, y, n=30):
    for _ in range(6):
        step(x, y)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        step(x, y)
    host = (time.perf_counter() - t) / n * 1e3
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, host


def main():
    msgs = torch.bfloat16 if sys.argv[1] == "bf16" else torch.float32
    os.environ["NEUROVIT_DP_UPDATE_PER_BUCKET"] = sys.argv[2]
    buckets = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    x = torch.randn(4, 128, 128, 128, device="cuda")
    y = torch.tensor([0, 1, 1, 0], device="cuda")
    ref = TrainStep(build(), fuse_update=int(os.environ["PROBE_REF_FUSE"]) if "PROBE_REF_FUSE" in os.environ else None)
    r_ms, r_host = timed(ref, x, y)
    step = TrainStep(build(), n_buckets=buckets, native_dp=True, grad_comm_dtype=msgs)
    d_ms, d_host = timed(step, x, y)
    r2_ms, _ = timed(ref, x, y)
    print(f"{sys.argv[1]} per_bucket={sys.argv[2]} buckets={buckets}: native-dp {d_ms:.3f} ms/step (host {d_host:.3f}) {step.last_dp}; single-process {r_ms:.3f} / {r2_ms:.3f} ms/step "
          f"(host {r_host:.3f}); overhead {100 * (d_ms / min(r_ms, r2_ms) - 1):+.1f} %", flush=True)


if __name__ == "__main__":
    main()
