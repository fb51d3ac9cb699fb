#!/usr/bin/env python3
"""Race check at full size: ViT3D-base, batch 4, 30 train steps in every fuse_update mode - parameters, both moments, the bf16 shadow and the
losses must be bit-identical to mode 0 (the per-layer update rewrites the bf16 weights while the backward pass is still running: a reader ordered
wrongly would show up here, where the kernels take their real time, not in the micro-model unit test)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

bench.torch = torch
from neurovit_amd import config as nvcfg  # noqa: E402
from neurovit_amd.NeuroEncoder import NeuroEncoder  # noqa: E402
from neurovit_amd.trainer import TrainStep  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
drop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
size = nvcfg.preset("base")
S = size["TRAINING_VIT_INPUT_SIZE"]
config = dict(DEVICE="cuda:0", TRAINING_DIM=3, TRAINING_DROPOUT=drop, GRADCAM_CUBE_SIZE=8, DATASET_NAME="adni", TRAINING_LEARNING_RATE=1e-4,
              TRAINING_WEIGHT_DECAY=1e-2, **size)
batches = [bench.make_batch(4, S, torch.device("cuda:0"), 42 + i) for i in range(3)]
ref = None
for mode in (0, 3, 1, 2, 3):
    torch.manual_seed(42)
    model = NeuroEncoder(config)
    model.train()
    step = TrainStep(model, fuse_update=mode)
    torch.manual_seed(7)
    losses = torch.stack([step(*batches[i % 3]).clone() for i in range(steps)])
    vit = model.volume_encoder.vit3d
    m, v = step.optimizer.arena_state(vit)
    got = (losses, vit.flat_parameters()[0].clone(), vit.flat_parameters()[1].clone(), m.clone(), v.clone())
    torch.cuda.synchronize()
    if ref is None:
        ref = got
        print(f"mode 0: {steps} steps, last loss {float(losses[-1]):.6f}", flush=True)
        continue
    same = [bool(torch.equal(a, b)) for a, b in zip(ref, got)]
    print(f"mode {mode}: losses / parameters / bf16 shadow / exp_avg / exp_avg_sq bit-identical to mode 0: {same}", flush=True)
    assert all(same), f"mode {mode} diverges from mode 0"
print("ok")
