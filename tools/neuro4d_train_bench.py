#!/usr/bin/env python3
"""BASELINE.json configs[3]: the 4D train step (frozen ViT3D-base over T = 20 timepoints, temporal head trained, accumulation 4).
Times one sample's micro-step against the bare frozen-encoder forward of the same 20 volumes: the difference is the temporal
head's forward / backward / optimizer share."""
import os, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurovit_amd import config as nvcfg
from neurovit_amd.NeuroEncoder import NeuroEncoder
from neurovit_amd.trainer import TrainStep

size = nvcfg.preset("base")
base = dict(DEVICE="cuda:0", TRAINING_DROPOUT=0.0, GRADCAM_CUBE_SIZE=8, DATASET_NAME="adni", **size)
torch.manual_seed(0)
m3 = NeuroEncoder(dict(base, TRAINING_DIM=3)).eval()
with tempfile.TemporaryDirectory() as td:
    torch.save(m3.state_dict(), os.path.join(td, "c.pth"))
    m4 = NeuroEncoder(dict(base, TRAINING_DIM=4, GLOBAL_BASE_PATH=td, BEST_MODEL_PATH="c.pth"))
del m3
m4.train()
m4.volume_encoder.eval()
B = int(os.environ.get("B4D", "1"))
x = torch.randn(B, 128, 128, 128, 20, device="cuda")
y = torch.randint(0, 2, (B,), device="cuda")
step = TrainStep(m4, accumulation_steps=4)


def timed(fn, n=20, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def enc():
    with torch.no_grad():
        m4.volume_encoder.vit3d(x, time_points=20)


t_enc = timed(enc)
t_step = timed(lambda: step(x, y))
print(f"4D train micro-step, {B} sample(s) x T = 20 (ViT3D-base frozen): {t_step:.3f} ms ({B * 20 / t_step * 1e3:.0f} volumes/s); frozen encoder forward alone {t_enc:.3f} ms; "
      f"temporal head + loss + optimizer share {t_step - t_enc:.3f} ms = {100 * (t_step - t_enc) / t_step:.1f} %")
