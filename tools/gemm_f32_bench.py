#!/usr/bin/env python3
"""fp32-MFMA GEMM (nv_gemm_f32) on the ViT3D-base shapes at batch 4, every wave tile.  Tuning aid."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurovit_amd import ops            # noqa: E402
from neurovit_amd._cabi import lib      # noqa: E402

SHAPES = [("qkv", 0, 2052, 2304, 768), ("out-proj", 4, 2052, 768, 768), ("fc1+gelu", 3, 2052, 3072, 768), ("fc2", 4, 2052, 768, 3072),
          ("patch", 2, 2048, 768, 4096), ("b20 qkv", 0, 10260, 2304, 768), ("b20 fc2", 4, 10260, 768, 3072), ("big", 0, 4096, 4096, 4096)]
for name, epi, M, N, K in SHAPES:
    A, W = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / K ** 0.5
    bias, resid = torch.randn(N, device="cuda"), torch.randn(M, N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    line = f"{name:9s} M={M:5d} N={N:5d} K={K:5d}:"
    for tile in ((0, 0), (2, 2), (2, 4), (4, 2), (4, 4)):
        lib.nv_gemm_f32_set_tile(*tile)
        kw = dict(bias=bias if epi >= 2 else None, resid=resid if epi == 4 else None, out=out)
        for _ in range(3):
            ops.gemm_f32(epi, A, W, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            ops.gemm_f32(epi, A, W, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        line += f"  {tile[0]}x{tile[1]}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF"
    lib.nv_gemm_f32_set_tile(0, 0)
    print(line, flush=True)
