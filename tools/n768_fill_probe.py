#!/usr/bin/env python3
"""Is the N = 768 data-gradient GEMM bound per CU (LDS fill of its one workgroup) or by something chip-wide?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurovit_amd import ops
from neurovit_amd._cabi import lib

def timeit(fn, iters=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

def run(layout, M, N, K, tile):
    lib.nv_gemm_set_tile(*tile)
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g).cuda().bfloat16()
    B = (torch.randn(K, N, generator=g) if layout == ops.NN else torch.randn(N, K, generator=g)).cuda().bfloat16()
    out = torch.empty(M, N, device="cuda")
    t = timeit(lambda: ops.gemm(layout, ops.EPI_STORE_F32, A, B, out=out))
    lib.nv_gemm_set_tile(0, 0)
    return t

for lname, layout in (("NN", ops.NN), ("NT", ops.NT)):
    for tname, tile, bm, bn in (("ws 64x128 ring 3x128", (3, 3), 64, 128), ("ws 64x128 ring 3x64", (3, 1), 64, 128), ("ws 128x128", (1, 1), 128, 128)):
        for (M, N, K) in ((2052, 768, 3072), (2048, 768, 3072), (2048, 1024, 3072), (1024, 768, 3072), (512, 768, 3072), (2048, 768, 1536), (2048, 768, 768), (2048, 768, 6144)):
            t = run(layout, M, N, K, tile)
            wgs = -(-M // bm) * -(-N // bn)
            print(f"{lname} {tname:22s} M {M:5d} N {N:5d} K {K:5d}: {t:7.2f} us  {wgs:4d} workgroups  {2.0 * M * N * K / t / 1e6:7.1f} TFLOP/s   fill per workgroup {(bm + bn) * K * 2 / 1e6:5.2f} MB -> {(bm + bn) * K * 2 / (t * 1e-6) / 1e9:6.1f} GB/s per CU", flush=True)
