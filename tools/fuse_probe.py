#!/usr/bin/env python3
"""Solo timing of a ViT3D-base layer's four weight-gradient GEMMs (K = 2052 rows) + AdamW of those 7.08 M weights:
grouped GEMM then nv_adamw_step over the same range, against nv_gemm_bf16_grouped_adamw (update in the epilogue)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurovit_amd import ops  # noqa: E402

K = 2052
shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
g = torch.Generator().manual_seed(1)
A = [torch.randn(K, m, generator=g).bfloat16().cuda() for m, n in shapes]
B = [(torch.randn(K, n, generator=g) * K ** -0.5).bfloat16().cuda() for m, n in shapes]
offs, cur = [], 0
for m, n in shapes:
    offs.append(cur)
    cur += m * n
total = cur
p, gr, mm, vv = (torch.randn(total, generator=g).cuda() * 0.05 for _ in range(4))
mm.zero_(); vv.abs_()
p16 = p.bfloat16()
views = [gr[o:o + m * n].view(m, n) for o, (m, n) in zip(offs, shapes)]


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def separate():
    ops.gemm_tn_grouped([(a, b, c, False) for a, b, c in zip(A, B, views)])
    ops.adamw_step(p, gr, mm, vv, p16, 3, 1e-4)


def gemm_only():
    ops.gemm_tn_grouped([(a, b, c, False) for a, b, c in zip(A, B, views)])


def fused(keep):
    opt = ops.adamw_arena(p, gr, mm, vv, p16, 3, 1e-4, keep_grads=keep)
    ops.gemm_tn_grouped_adamw(list(zip(A, B, views)), opt)


for _ in range(2):
    print(f"grouped GEMM alone {timed(gemm_only):7.2f} us | GEMM + AdamW launch {timed(separate):7.2f} us | fused {timed(lambda: fused(False)):7.2f} us | fused, gradients kept {timed(lambda: fused(True)):7.2f} us"
          f"   ({total / 1e6:.2f} M weights: 34 / 26 / 30 B per weight)", flush=True)
