#!/usr/bin/env python3
"""Experiment: the engine's auxiliary (weight-gradient) stream at LOW queue priority, with the grouped weight gradients on long (256x128,
36 us per workgroup) or short (64x128, 9 us) tiles - do the weight gradients then fill the CUs the data-gradient chain leaves idle
(58 of 256 during its 198-workgroup GEMMs) instead of competing with it?   usage: prio_probe.py [steps]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurovit_amd._cabi import lib  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def prio_stream(prio):
    st = ctypes.c_void_p()
    rc = hip.hipStreamCreateWithPriority(ctypes.byref(st), 1, prio)          # hipStreamNonBlocking
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def main():
    lo, hi = ctypes.c_int(), ctypes.c_int()
    hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
    print(f"stream priority range: least {lo.value}, greatest {hi.value}", flush=True)
    import bench
    bench.torch = torch
    from neurovit_amd import config as nvcfg
    from neurovit_amd.NeuroEncoder import NeuroEncoder
    from neurovit_amd.trainer import TrainStep
    size = nvcfg.preset("base")
    S = size["TRAINING_VIT_INPUT_SIZE"]
    config = dict(DEVICE="cuda:0", TRAINING_DIM=3, TRAINING_DROPOUT=0.0, GRADCAM_CUBE_SIZE=8, DATASET_NAME="adni", TRAINING_LEARNING_RATE=1e-4,
                  TRAINING_WEIGHT_DECAY=1e-2, **size)
    x, y = bench.make_batch(4, S, torch.device("cuda:0"), 42)

    def run(tag, aux_prio, main_prio, pp_grouped):
        lib.nv_gemm_set_tile(7, 1 if pp_grouped else 0)
        torch.manual_seed(42)
        model = NeuroEncoder(config)
        model.train()
        step = TrainStep(model)
        vit = model.volume_encoder.vit3d
        if aux_prio is not None:
            vit._rt._aux["cuda:0"] = prio_stream(aux_prio)
        import contextlib
        ctx = torch.cuda.stream(prio_stream(main_prio)) if main_prio is not None else contextlib.nullcontext()
        with ctx:
            for _ in range(10):
                loss = step(x, y)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = step(x, y)
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        print(f"{tag:78s} {ms:7.3f} ms/step  {4 / ms * 1e3:8.1f} volumes/s  loss {float(loss):.5f}", flush=True)
        lib.nv_gemm_set_tile(7, 1)

    for _ in range(2):
        run("default (aux stream from torch's pool, weight gradients on 256x128 tiles)", None, None, True)
        run("aux stream at the LEAST priority, 256x128 tiles", lo.value, None, True)
        run("aux stream at the LEAST priority, 64x128 tiles (short workgroups)", lo.value, None, False)
        run("aux from the pool, 64x128 tiles", None, None, False)
        run("main stream at the GREATEST priority, aux least, 64x128 tiles", lo.value, hi.value, False)
        run("main stream at the GREATEST priority, aux least, 256x128 tiles", lo.value, hi.value, True)


if __name__ == "__main__":
    main()
