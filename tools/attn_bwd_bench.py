#!/usr/bin/env python3
"""Backward attention kernels side by side (nv_attn_set_mode: 1 streaming, 0 heuristic, 2 LDS-resident, 4 wide dQ + wide dK/dV).  Tuning aid."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurovit_amd import ops
from neurovit_amd._cabi import lib
for (B, n, heads) in ((4, 513, 12), (20, 513, 12), (4, 4097, 16), (16, 1001, 8), (2, 1001, 2)):
    qkv = torch.randn(B * n, 3 * heads * 64, device="cuda").bfloat16()
    do = torch.randn(B * n, heads * 64, device="cuda").bfloat16()
    lib.nv_attn_set_mode(1)
    out, lse = ops.attn_fwd(qkv, B, n, heads)
    ref, dref = ops.attn_bwd(qkv, out, do, lse, B, n, heads)
    for mode in (1, 0, 2, 4):
        if mode == 2 and n > 576:
            continue
        lib.nv_attn_set_mode(mode)
        for _ in range(2):
            dqkv, delta = ops.attn_bwd(qkv, out, do, lse, B, n, heads)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(10):
            ops.attn_bwd(qkv, out, do, lse, B, n, heads)
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 100
        d = (dqkv.float() - ref.float()).abs()
        print(f"B={B} n={n} heads={heads} mode {mode}: {us:8.1f} us  {10.0 * B * heads * n * n * 64 / us / 1e6:6.0f} TFLOP/s   mismatching elements vs streaming: {(d > 0).float().mean().item():.2e} max {d.max().item():.3e}  delta equal {torch.equal(delta, dref)}", flush=True)
    lib.nv_attn_set_mode(0)
