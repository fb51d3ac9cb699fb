#!/usr/bin/env python3
"""Forward attention kernels side by side (nv_attn_set_mode: 1 streaming, 2 LDS-resident, 3 wide streaming).  Tuning aid."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurovit_amd import ops
from neurovit_amd._cabi import lib
for (B, n, heads) in ((4, 513, 12), (20, 513, 12), (4, 4097, 16), (16, 1001, 8)):
    qkv = torch.randn(B * n, 3 * heads * 64, device="cuda").bfloat16()
    for mode in (1, 2, 3):
        if mode == 2 and n > 576:
            continue
        lib.nv_attn_set_mode(mode)
        for _ in range(3):
            ops.attn_fwd(qkv, B, n, heads)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(20):
            ops.attn_fwd(qkv, B, n, heads)
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 50
        print(f"B={B} n={n} heads={heads} mode {mode}: {us:8.1f} us  {4.0 * B * heads * n * n * 64 / us / 1e6:6.0f} TFLOP/s", flush=True)
    lib.nv_attn_set_mode(0)
