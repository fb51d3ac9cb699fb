#!/usr/bin/env python3
"""Attention launches for a GPU-side timing (host loops are launch-bound at these sizes: wrap in rocprofv3 --kernel-trace and
read the per-dispatch durations with tools/trace_durations.py).  usage: attn_trace_bench.py [mode] [B n heads]..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurovit_amd import ops  # noqa: E402
from neurovit_amd._cabi import lib  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
shapes = [tuple(int(v) for v in sys.argv[i:i + 3]) for i in range(2, len(sys.argv) - 2, 3)] or [(4, 513, 12), (20, 513, 12), (4, 4097, 16)]
lib.nv_attn_set_mode(mode)
for (B, n, heads) in shapes:
    qkv = torch.randn(B * n, 3 * heads * 64, device="cuda").bfloat16()
    do = torch.randn(B * n, heads * 64, device="cuda").bfloat16()
    for _ in range(24):
        out, lse = ops.attn_fwd(qkv, B, n, heads)
        ops.attn_bwd(qkv, out, do, lse, B, n, heads)
    torch.cuda.synchronize()
