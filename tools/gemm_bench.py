#!/usr/bin/env python3
"""Micro-benchmark of the GEMM shapes of ViT3D-base (B=4 -> M=2052) through the C-ABI.  Tuning aid, not the bench."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurovit_amd import ops  # noqa: E402

M, d, inner, m = 2052, 768, 768, 3072
SHAPES = [  # name, layout, epi, M, N, K
    ("qkv      NT bf16 ", ops.NT, ops.EPI_STORE_BF16, M, 3 * inner, d),
    ("out-proj NT resid", ops.NT, ops.EPI_BIAS_RESID, M, d, inner),
    ("fc1      NT gelu ", ops.NT, ops.EPI_BIAS_GELU, M, m, d),
    ("fc2      NT resid", ops.NT, ops.EPI_BIAS_RESID, M, d, m),
    ("patch    NT bias ", ops.NT, ops.EPI_BIAS_F32, 2048, d, 4096),
    ("dU       NN dgelu", ops.NN, ops.EPI_DGELU, M, m, d),
    ("dxn(fc1) NN f32  ", ops.NN, ops.EPI_STORE_F32, M, d, m),
    ("dAO      NN bf16 ", ops.NN, ops.EPI_STORE_BF16, M, inner, d),
    ("dxn(qkv) NN f32  ", ops.NN, ops.EPI_STORE_F32, M, d, 3 * inner),
    ("dW2      TN f32  ", ops.TN, ops.EPI_STORE_F32, d, m, M),
    ("dW1      TN f32  ", ops.TN, ops.EPI_STORE_F32, m, d, M),
    ("dWo      TN f32  ", ops.TN, ops.EPI_STORE_F32, d, inner, M),
    ("dWqkv    TN f32  ", ops.TN, ops.EPI_STORE_F32, 3 * inner, d, M),
    ("big      NT bf16 ", ops.NT, ops.EPI_STORE_BF16, 8192, 8192, 4096),
    ("xfc1shp  NT bf16 ", ops.NT, ops.EPI_STORE_BF16, M, m, d),
    ("xfc1shp  NT f32  ", ops.NT, ops.EPI_STORE_F32, M, m, d),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--tile", default="", help="force tile, e.g. 64x128 (default: heuristic)")
    ap.add_argument("--debug", type=int, default=0)
    a = ap.parse_args()
    assert not a.debug, "the ablation hooks were removed from the library (see profiles/r01_gemm_ablation.log)"
    if a.tile:
        from neurovit_amd._cabi import lib
        bm, bn = {"ws128x128": (1, 1), "ws128x64": (2, 1), "ws64x128": (3, 1), "ws128x128d": (1, 2), "ws128x64d": (2, 2), "ws64x128d": (3, 2), "ws128x64k": (2, 3), "ws64x128k": (3, 3)}.get(a.tile) or tuple(int(v) for v in a.tile.split("x"))
        lib.nv_gemm_set_tile(bm, bn)
        print(f"--- tile {bm}x{bn}")
    dev = "cuda"
    tot_t, tot_f = 0.0, 0.0
    for name, layout, epi, Mo, N, K in SHAPES:
        if a.only and a.only not in name:
            continue
        g = torch.Generator(device="cpu").manual_seed(1)
        if layout == ops.NT:
            A, B = torch.randn(Mo, K, generator=g), torch.randn(N, K, generator=g)
        elif layout == ops.NN:
            A, B = torch.randn(Mo, K, generator=g), torch.randn(K, N, generator=g)
        else:
            A, B = torch.randn(K, Mo, generator=g), torch.randn(K, N, generator=g)
        A, B = A.to(dev).bfloat16(), B.to(dev).bfloat16()
        bias = torch.randn(N, device=dev)
        resid = torch.randn(Mo, N, device=dev)
        u = torch.randn(Mo, N, device=dev).bfloat16()
        kw = {}
        if epi in (ops.EPI_BIAS_F32, ops.EPI_BIAS_GELU, ops.EPI_BIAS_RESID):
            kw["bias"] = bias
        if epi == ops.EPI_BIAS_RESID:
            kw["aux_in"] = resid
        if epi == ops.EPI_DGELU:
            kw["aux_in"] = u
        if epi == ops.EPI_BIAS_GELU:
            kw["aux_out"] = torch.empty_like(u)
        out = ops.gemm(layout, epi, A, B, **kw)
        for _ in range(3):
            ops.gemm(layout, epi, A, B, out=out, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            ops.gemm(layout, epi, A, B, out=out, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        fl = 2.0 * Mo * N * K
        if "big" not in name and not name.startswith("x"):
            tot_t += us; tot_f += fl
        print(f"{name}  M={Mo:5d} N={N:5d} K={K:5d}  {us:8.2f} us  {fl / us / 1e6:8.1f} TFLOP/s", flush=True)
    if tot_t:
        print(f"model shapes total: {tot_t:.1f} us, {tot_f / tot_t / 1e6:.1f} TFLOP/s aggregate")


if __name__ == "__main__":
    main()
