#!/usr/bin/env python3
"""Hypothesis check before any restructuring: would the batch-4 step gain from running as two batch-2 halves on two streams (one half's
LayerNorm / attention beside the other half's GEMMs)?  Two INDEPENDENT ViT3D-base models train concurrently, each on a stream of its
own with batch 2, against one model at batch 4 - forward + backward only (no optimizer update: an accumulation window that never
closes), and full steps.  usage: split_probe.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

bench.torch = torch
from neurovit_amd import config as nvcfg  # noqa: E402
from neurovit_amd.NeuroEncoder import NeuroEncoder  # noqa: E402
from neurovit_amd.trainer import TrainStep  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
size = nvcfg.preset("base")
S = size["TRAINING_VIT_INPUT_SIZE"]
config = dict(DEVICE="cuda:0", TRAINING_DIM=3, TRAINING_DROPOUT=0.0, GRADCAM_CUBE_SIZE=8, DATASET_NAME="adni", TRAINING_LEARNING_RATE=1e-4,
              TRAINING_WEIGHT_DECAY=1e-2, **size)


def make(acc):
    torch.manual_seed(42)
    m = NeuroEncoder(config)
    m.train()
    return TrainStep(m, accumulation_steps=acc)


def run(tag, nmodels, batch, acc):
    models = [make(acc) for _ in range(nmodels)]
    data = [bench.make_batch(batch, S, torch.device("cuda:0"), 42 + i) for i in range(nmodels)]
    streams = [torch.cuda.Stream() for _ in range(nmodels)] if nmodels > 1 else [torch.cuda.current_stream()]

    def one():
        for st, (x, y), sm in zip(models, data, streams):
            with torch.cuda.stream(sm):
                st(x, y)

    for _ in range(8):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{tag:72s} {ms:7.3f} ms per round  {nmodels * batch / ms * 1e3:8.1f} volumes/s", flush=True)
    del models


NEVER = 1 << 30
for _ in range(2):
    run("one model, batch 4, forward + backward (no update)", 1, 4, NEVER)
    run("two models, batch 2 each, two streams, forward + backward (no update)", 2, 2, NEVER)
    run("one model, batch 2, forward + backward (no update)", 1, 2, NEVER)
    run("one model, batch 4, full step", 1, 4, 1)
    run("two models, batch 2 each, two streams, full steps", 2, 2, 1)
