#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels at the ViT3D-base shape (B=4, n=513, 12 heads, dh=64).  Tuning aid."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurovit_amd import ops  # noqa: E402

B, n, heads = 4, 513, 12
inner = heads * 64
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * n, 3 * inner, generator=g).cuda().bfloat16()
do = torch.randn(B * n, inner, generator=g).cuda().bfloat16()
out, lse = ops.attn_fwd(qkv, B, n, heads)


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


from neurovit_amd._cabi import lib  # noqa: E402
fl = 4.0 * B * heads * n * n * 64
for mode, name in ((1, "streaming      "), (2, "resident x1    "), (22, "resident x2    ")):
    lib.nv_attn_set_mode(mode)
    tf = timeit(lambda: ops.attn_fwd(qkv, B, n, heads))
    tb = timeit(lambda: ops.attn_bwd(qkv, out, do, lse, B, n, heads))
    print(f"{name}: attn fwd {tf:7.2f} us ({fl / tf / 1e6:6.1f} TFLOP/s)   bwd (dq+dkv) {tb:7.2f} us ({2.5 * fl / tb / 1e6:6.1f} TFLOP/s algorithmic)")
lib.nv_attn_set_mode(0)
