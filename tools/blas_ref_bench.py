#!/usr/bin/env python3
"""Reality check for tools/gemm_bench.py: the same ViT3D-base GEMM shapes through torch.matmul (hipBLASLt / rocBLAS, plain
bf16 output, no fused epilogue).  Tuning aid only - the product never calls a BLAS library."""
import torch

M, d, m = 2052, 768, 3072
SHAPES = [("qkv NT", M, 3 * d, d), ("out-proj NT", M, d, d), ("fc1 NT", M, m, d), ("fc2 NT", M, d, m), ("patch NT", 2048, d, 4096),
          ("dW2 TN", d, m, M), ("dWqkv TN", 3 * d, d, M), ("big NT", 8192, 8192, 4096)]


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for name, Mo, N, K in SHAPES:
    if "TN" in name:
        A = torch.randn(K, Mo, device="cuda").bfloat16().t()      # K-strided operands, like the weight-gradient GEMMs
        B = torch.randn(K, N, device="cuda").bfloat16()
        fn = lambda: torch.matmul(A, B)
    else:
        A = torch.randn(Mo, K, device="cuda").bfloat16()
        B = torch.randn(N, K, device="cuda").bfloat16()
        fn = lambda: torch.matmul(A, B.t())
    t = timeit(fn)
    print(f"{name:12s} M={Mo:5d} N={N:5d} K={K:5d}  {t:8.2f} us  {2.0 * Mo * N * K / t / 1e6:7.1f} TFLOP/s")
