#!/usr/bin/env python3
"""BASELINE.json configs[3] shape: one 4D sample = ViT3D-base over T = 20 timepoints + temporal head.  Times the forward with the fused
4D gather (no regroup copy) against the reference's regroup copy + per-volume gather, and the raw-volume path against zscore_crop + forward."""
import os, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurovit_amd import config as nvcfg
from neurovit_amd.NeuroEncoder import NeuroEncoder
from neurovit_amd.preprocess import zscore_crop

size = nvcfg.preset("base")
base = dict(DEVICE="cuda:0", TRAINING_DROPOUT=0.0, GRADCAM_CUBE_SIZE=8, DATASET_NAME="adni", **size)
torch.manual_seed(0)
m3 = NeuroEncoder(dict(base, TRAINING_DIM=3)).eval()
with tempfile.TemporaryDirectory() as td:
    torch.save(m3.state_dict(), os.path.join(td, "c.pth"))
    m4 = NeuroEncoder(dict(base, TRAINING_DIM=4, GLOBAL_BASE_PATH=td, BEST_MODEL_PATH="c.pth")).eval()


def timed(fn, n=10):
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


x = torch.randn(1, 128, 128, 128, 20, device="cuda")
fused = timed(lambda: m4(x))
xs = x[..., :19]          # T = 19: not a multiple of 4 -> the copy path
xs20 = torch.randn(1, 128, 128, 128, 20, device="cuda").transpose(1, 2)   # non-contiguous -> the copy path at T = 20
copy = timed(lambda: m4(xs20))
print(f"4D sample (T = 20, ViT3D-base): fused gather {fused:.3f} ms / sample, regroup copy + per-volume gather {copy:.3f} ms / sample")
raw = (torch.randn(4, 129, 147, 129, device="cuda") * 40 + 300)
a = timed(lambda: m3.forward_raw(raw), 30)
b = timed(lambda: m3(zscore_crop(raw)), 30)
print(f"raw volumes (batch 4): forward_raw {a:.3f} ms, zscore_crop + forward {b:.3f} ms")
