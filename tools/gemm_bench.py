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
    ("b20 qkv  NT bf16 ", ops.NT, ops.EPI_STORE_BF16, 10260, 3 * inner, d),
    ("b20 fc1  NT gelu ", ops.NT, ops.EPI_BIAS_GELU, 10260, m, d),
    ("b20 fc2  NT resid", ops.NT, ops.EPI_BIAS_RESID, 10260, d, m),
    ("b20 out  NT resid", ops.NT, ops.EPI_BIAS_RESID, 10260, d, inner),
    ("lrg qkv  NT bf16 ", ops.NT, ops.EPI_STORE_BF16, 16388, 3072, 1024),
    ("lrg fc1  NT gelu ", ops.NT, ops.EPI_BIAS_GELU, 16388, 4096, 1024),
    ("lrg fc2  NT resid", ops.NT, ops.EPI_BIAS_RESID, 16388, 1024, 4096),
    ("lrg dU   NN dgelu", ops.NN, ops.EPI_DGELU, 16388, 4096, 1024),
    ("lrg dxn1 NN f32  ", ops.NN, ops.EPI_STORE_F32, 16388, 1024, 4096),
    ("lrg dxnq NN f32  ", ops.NN, ops.EPI_STORE_F32, 16388, 1024, 3072),
    ("lrg dW1  TN f32  ", ops.TN, ops.EPI_STORE_F32, 4096, 1024, 16388),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--tile", default="", help="force tile, e.g. 64x128 (default: heuristic)")
    ap.add_argument("--w32", action="store_true", help="NT problems of the 256x128 kernel on v_mfma_f32_32x32x16_bf16 (nv_gemm_set_tile(11, 1))")
    ap.add_argument("--dbg", type=int, default=0, help="256x128 kernel timing ablation (results wrong by design): 1 no DMA, 2 DMA from one L2-hot region, 4 fragments read once, 5 MFMA only")
    a = ap.parse_args()
    if a.dbg:
        from neurovit_amd._cabi import lib as _l
        _l.nv_gemm_set_tile(8, a.dbg)
    if a.w32:
        from neurovit_amd._cabi import lib as _l2
        _l2.nv_gemm_set_tile(11, 1)
        print("--- 32x32x16 MFMA form for NT problems on 256x128 tiles")
    if a.tile:
        from neurovit_amd._cabi import lib
        bm, bn = {"ws128x128": (1, 1), "ws64x128": (3, 1), "ws64x128k": (3, 3), "pp": (4, 0), "nopp": (5, 0), "pq": (9, 0)}.get(a.tile) or tuple(int(v) for v in a.tile.split("x"))
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
        if "big" not in name and not name.startswith(("x", "b20", "lrg")):
            tot_t += us; tot_f += fl
        print(f"{name}  M={Mo:5d} N={N:5d} K={K:5d}  {us:8.2f} us  {fl / us / 1e6:8.1f} TFLOP/s", flush=True)
    if not a.only or "f8" in a.only:
        for name, Mo, N, K in (("f8 big ", 8192, 8192, 4096), ("f8 lrg qkv", 16388, 3072, 1024), ("f8 lrg fc1 gelu->f8", 16388, 4096, 1024), ("f8 lrg fc2 resid", 16388, 1024, 4096),
                               ("f8 b20 qkv", 10260, 2304, 768)):
            A8 = torch.randint(0, 120, (Mo, K), dtype=torch.uint8, device=dev)
            B8 = torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev)
            cs = torch.rand(N, device=dev) * 1e-3
            bias = torch.randn(N, device=dev)
            resid = torch.randn(Mo, N, device=dev)
            epi = ops.EPI_BIAS_GELU_F8 if "gelu" in name else (ops.EPI_BIAS_RESID if "resid" in name else ops.EPI_STORE_BF16)
            kw = dict(bias=bias) if epi != ops.EPI_STORE_BF16 else {}
            if epi == ops.EPI_BIAS_RESID:
                kw["aux_in"] = resid
            out = ops.gemm_f8(epi, A8, B8, cs, **kw)
            for _ in range(3):
                ops.gemm_f8(epi, A8, B8, cs, out=out, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(a.iters):
                ops.gemm_f8(epi, A8, B8, cs, out=out, **kw)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            print(f"{name:22s} M={Mo:5d} N={N:5d} K={K:5d}  {us:8.2f} us  {2.0 * Mo * N * K / us / 1e6:8.1f} TFLOP/s", flush=True)
    if not a.only or "grouped" in a.only:
        g = torch.Generator(device="cpu").manual_seed(2)
        probs = []
        fl = 0.0
        for Mo, N in ((d, m), (m, d), (d, inner), (3 * inner, d)):
            At, B2 = torch.randn(M, Mo, generator=g).to(dev).bfloat16(), torch.randn(M, N, generator=g).to(dev).bfloat16()
            probs.append((At, B2, torch.empty(Mo, N, device=dev), False))
            fl += 2.0 * M * Mo * N
        for _ in range(3):
            ops.gemm_tn_grouped(probs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            ops.gemm_tn_grouped(probs)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        print(f"grouped wgrad (4 problems of a layer, K={M})        {us:8.2f} us  {fl / us / 1e6:8.1f} TFLOP/s", flush=True)
    if tot_t:
        print(f"model shapes total: {tot_t:.1f} us, {tot_f / tot_t / 1e6:.1f} TFLOP/s aggregate")


if __name__ == "__main__":
    main()
