"""Per-kernel parity on a real MI355X, through the C-ABI (neurovit_amd.ops -> libneurovit_hip.so).

Checker = the CPU oracle (oracle/ref_cpu.py) or a plain fp32/fp64 restatement of the single op.
Tolerances (north_star: "within 1e-3 rel bf16; bit-exact for patch indexing / cls-token scatter"):
  * fp32 outputs : max|a-b| <= 1e-3 * max|b|                       (REL; most are held to 1e-5)
  * bf16 outputs : |a-b| <= 1e-3 * max|b| + 1 bf16 ulp of |b|      (a value on a rounding boundary may
                   round the other way when the fp32 accumulation order differs)
  * attention    : ||a-b||_2 <= 1e-3 ||b||_2 and max|a-b| <= 2^-7 max|b| - the probabilities P are themselves
                   rounded to bf16 INSIDE the op (MFMA operand), so a one-ulp flip of a large P moves an output
                   element by up to ~2^-8 of |v|; such flips are isolated and do not move the L2 error.
  * integer / index work: bit exact.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import weights as W
from conftest import rel_err, rel_l2, report
from oracle import ref_cpu, train_step

pytestmark = pytest.mark.gpu
REL = 1e-3


@pytest.fixture(scope="module")
def ops():
    from neurovit_amd import ops as _ops
    from neurovit_amd._cabi import require_gpu
    require_gpu()
    return _ops


def dev(t):
    return t.cuda()


def bf(t):
    return t.to(torch.bfloat16)


def assert_close_bf16(a, b, what=""):
    a, b = a.detach().float().cpu().double(), b.detach().float().cpu().double()
    ulp = 2.0 ** (torch.floor(torch.log2(b.abs().clamp_min(1e-30))) - 7)
    tol = REL * b.abs().max() + ulp
    bad = ((a - b).abs() > tol)
    assert not bad.any(), f"{what}: {int(bad.sum())} / {bad.numel()} outside tol, max diff {(a - b).abs().max():.3e}, max ref {b.abs().max():.3e}"


def assert_close_stat(a, b, what=""):
    l2, mx = rel_l2(a.float(), b), rel_err(a.float(), b)
    assert l2 <= REL and mx <= 2.0 ** -7, f"{what}: rel_l2 {l2:.3e}, rel_max {mx:.3e}"


def assert_close_f32(a, b, what="", rel=REL):
    e = rel_err(a, b)
    assert e <= rel, f"{what}: rel err {e:.3e} > {rel}"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ------------------------------------------------------------------------------------------ GEMM
SHAPES = [(130, 136, 72), (128, 128, 64), (257, 264, 200), (65, 192, 4096), (2052, 768, 768), (16, 8, 8)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_nt_epilogues(ops, M, N, K):
    A, B = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=K ** -0.5))
    bias, resid = rnd(N, seed=3), rnd(M, N, seed=4)
    ref = A.double() @ B.double().T
    Ad, Bd = dev(A), dev(B)
    assert_close_bf16(ops.gemm(ops.NT, ops.EPI_STORE_BF16, Ad, Bd), ref, "store_bf16")
    assert_close_f32(ops.gemm(ops.NT, ops.EPI_STORE_F32, Ad, Bd), ref, "store_f32", 1e-5)
    assert_close_f32(ops.gemm(ops.NT, ops.EPI_BIAS_F32, Ad, Bd, bias=dev(bias)), ref + bias.double(), "bias_f32", 1e-5)
    out = ops.gemm(ops.NT, ops.EPI_BIAS_RESID, Ad, Bd, bias=dev(bias), aux_in=dev(resid))
    assert_close_f32(out, ref + bias.double() + resid.double(), "bias_resid", 1e-5)
    u = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    h = ops.gemm(ops.NT, ops.EPI_BIAS_GELU, Ad, Bd, bias=dev(bias), aux_out=u)
    uref = ref + bias.double()
    assert_close_bf16(u, uref, "gelu.u")
    assert_close_bf16(h, F.gelu(uref), "gelu.h")
    # accumulate: C += A B^T
    c0 = rnd(M, N, seed=5)
    c = dev(c0.clone())
    ops.gemm(ops.NT, ops.EPI_STORE_F32, Ad, Bd, out=c, accumulate=True)
    assert_close_f32(c, ref + c0.double(), "accumulate", 1e-5)


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_nn_tn(ops, M, N, K):
    A, Bt = bf(rnd(M, K, seed=6)), bf(rnd(K, N, seed=7, scale=K ** -0.5))
    ref = A.double() @ Bt.double()
    assert_close_f32(ops.gemm(ops.NN, ops.EPI_STORE_F32, dev(A), dev(Bt)), ref, "nn_f32", 1e-5)
    assert_close_bf16(ops.gemm(ops.NN, ops.EPI_STORE_BF16, dev(A), dev(Bt)), ref, "nn_bf16")
    u = bf(rnd(M, N, seed=8))
    dg = ops.gemm(ops.NN, ops.EPI_DGELU, dev(A), dev(Bt), aux_in=dev(u))
    assert_close_bf16(dg, ref * ref_cpu._gelu_grad(u.double()), "dgelu")
    # TN: C[Mo, N] = A[K, Mo]^T B[K, N]   (K plays the token dimension; ragged K is the 2052-row case)
    Mo = (M + 7) // 8 * 8
    At, B2 = bf(rnd(K, Mo, seed=9)), bf(rnd(K, N, seed=10, scale=K ** -0.5))
    assert_close_f32(ops.gemm(ops.TN, ops.EPI_STORE_F32, dev(At), dev(B2)), At.double().T @ B2.double(), "tn_f32", 1e-5)


def test_gemm_tn_ragged_tokens(ops):
    K, Mo, N = 2052, 768, 264          # reduction over 2052 token rows (not a multiple of the 64-deep K tile)
    At, B2 = bf(rnd(K, Mo, seed=11)), bf(rnd(K, N, seed=12, scale=K ** -0.5))
    assert_close_f32(ops.gemm(ops.TN, ops.EPI_STORE_F32, dev(At), dev(B2)), At.double().T @ B2.double(), "tn_ragged", 1e-5)


def test_gemm_tn_grouped_equals_single_launches(ops):
    """The four weight-gradient GEMMs of a layer in ONE grouped launch (csrc/gemm.hip::gemm_ws_grouped_kernel): every problem
    bit-identical to its own nv_gemm_bf16 launch through the same 64x128 kernel (store and accumulate), ragged token count."""
    from neurovit_amd._cabi import lib
    K = 2052
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    probs, singles = [], []
    lib.nv_gemm_set_tile(3, 1)            # the grouped kernel's configuration: warp-specialised 64x128, 3 x 64-deep ring
    try:
        for i, (Mo, N) in enumerate(shapes):
            At, B2 = dev(bf(rnd(K, Mo, seed=20 + i))), dev(bf(rnd(K, N, seed=30 + i, scale=K ** -0.5)))
            acc = i % 2 == 1
            base = dev(rnd(Mo, N, seed=40 + i)) if acc else torch.empty(Mo, N, device="cuda")
            C1 = base.clone()
            singles.append(ops.gemm(ops.TN, ops.EPI_STORE_F32, At, B2, out=base.clone(), accumulate=acc))
            probs.append((At, B2, C1, acc))
    finally:
        lib.nv_gemm_set_tile(0, 0)
    ops.gemm_tn_grouped(probs)
    for (At, B2, C1, acc), ref in zip(probs, singles):
        assert torch.equal(C1, ref)
    ops.gemm_tn_grouped(probs[2:3])                                   # a group of one
    assert_close_f32(probs[2][2], probs[2][0].double().T @ probs[2][1].double(), "grouped_single", 1e-5)


@pytest.mark.parametrize("ws", [1, 3])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (257, 264, 192), (130, 136, 128), (2052, 768, 768), (300, 8, 64)])
def test_gemm_warp_specialised_kernel_forced(ops, M, N, K, ws):
    """The warp-specialised LDS-DMA kernels (128x128, 64x128) on ragged M / N (hardware bounds give
    zeros) and on a ragged token count."""
    from neurovit_amd._cabi import lib
    lib.nv_gemm_set_tile(ws, 0)
    try:
        A, B = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=K ** -0.5))
        bias, resid = rnd(N, seed=3), rnd(M, N, seed=4)
        ref = A.double() @ B.double().T
        assert_close_f32(ops.gemm(ops.NT, ops.EPI_BIAS_RESID, dev(A), dev(B), bias=dev(bias), aux_in=dev(resid)),
                         ref + bias.double() + resid.double(), "pipe.nt", 1e-5)
        Bt = bf(rnd(K, N, seed=7, scale=K ** -0.5))
        assert_close_f32(ops.gemm(ops.NN, ops.EPI_STORE_F32, dev(A), dev(Bt)), A.double() @ Bt.double(), "pipe.nn", 1e-5)
        Kt, Mo = M, (K + 7) // 8 * 8                      # TN: reduction over a ragged token count
        At, B2 = bf(rnd(Kt, Mo, seed=9)), bf(rnd(Kt, N, seed=10, scale=Kt ** -0.5))
        c0 = rnd(Mo, N, seed=11)
        c = dev(c0.clone())
        ops.gemm(ops.TN, ops.EPI_STORE_F32, dev(At), dev(B2), out=c, accumulate=True)
        assert_close_f32(c, At.double().T @ B2.double() + c0.double(), "pipe.tn", 1e-5)
    finally:
        lib.nv_gemm_set_tile(0, 0)


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (257, 264, 192), (2052, 768, 768), (2052, 2304, 768), (520, 392, 1024), (300, 8, 128)])
def test_gemm_ping_pong_kernel_forced(ops, M, N, K):
    """The eight-wave 256 x 128 ping-pong kernel (csrc/gemm_pp.hip): every layout and fused epilogue against fp64, ragged M / N
    (hardware bounds), one to sixteen K tiles (prologue / steady state / drain of the three-stage ring), ragged token count of
    the weight-gradient layout, and run-to-run bit equality (the staggered wave groups must not race on the ring)."""
    from neurovit_amd._cabi import lib
    lib.nv_gemm_set_tile(4, 0)
    try:
        A, B = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=K ** -0.5))
        bias, resid = rnd(N, seed=3), rnd(M, N, seed=4)
        ref = A.double() @ B.double().T
        Ad, Bd = dev(A), dev(B)
        first = ops.gemm(ops.NT, ops.EPI_STORE_BF16, Ad, Bd)
        assert_close_bf16(first, ref, "pp.nt_bf16")
        for _ in range(5):
            assert torch.equal(ops.gemm(ops.NT, ops.EPI_STORE_BF16, Ad, Bd), first)
        assert_close_f32(ops.gemm(ops.NT, ops.EPI_BIAS_F32, Ad, Bd, bias=dev(bias)), ref + bias.double(), "pp.bias_f32", 1e-5)
        assert_close_f32(ops.gemm(ops.NT, ops.EPI_BIAS_RESID, Ad, Bd, bias=dev(bias), aux_in=dev(resid)),
                         ref + bias.double() + resid.double(), "pp.nt_resid", 1e-5)
        u = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        h = ops.gemm(ops.NT, ops.EPI_BIAS_GELU, Ad, Bd, bias=dev(bias), aux_out=u)
        assert_close_bf16(u, ref + bias.double(), "pp.gelu.u")
        assert_close_bf16(h, F.gelu(ref + bias.double()), "pp.gelu.h")
        Bt = bf(rnd(K, N, seed=7, scale=K ** -0.5))
        refn = A.double() @ Bt.double()
        assert_close_f32(ops.gemm(ops.NN, ops.EPI_STORE_F32, Ad, dev(Bt)), refn, "pp.nn_f32", 1e-5)
        assert_close_bf16(ops.gemm(ops.NN, ops.EPI_STORE_BF16, Ad, dev(Bt)), refn, "pp.nn_bf16")
        ug = bf(rnd(M, N, seed=8))
        assert_close_bf16(ops.gemm(ops.NN, ops.EPI_DGELU, Ad, dev(Bt), aux_in=dev(ug)), refn * ref_cpu._gelu_grad(ug.double()), "pp.dgelu")
        Kt, Mo = M, (K + 7) // 8 * 8                      # TN: reduction over a ragged token count
        At, B2 = bf(rnd(Kt, Mo, seed=9)), bf(rnd(Kt, N, seed=10, scale=Kt ** -0.5))
        c0 = rnd(Mo, N, seed=11)
        c = dev(c0.clone())
        ops.gemm(ops.TN, ops.EPI_STORE_F32, dev(At), dev(B2), out=c, accumulate=True)
        assert_close_f32(c, At.double().T @ B2.double() + c0.double(), "pp.tn", 1e-5)
    finally:
        lib.nv_gemm_set_tile(0, 0)


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (257, 264, 192), (2052, 2304, 768), (2052, 768, 3072), (520, 392, 1024), (300, 8, 128)])
def test_gemm_ping_pong_kernel_on_32x32x16_mfma(ops, M, N, K):
    """The 256 x 128 kernel's NT problems on v_mfma_f32_32x32x16_bf16 (nv_gemm_set_tile(11, 1): 32-row fragments from an image with a
    swizzle of its own, accumulators parked from the 32 x 32 register layout): every NT epilogue against fp64 at the gates of the
    16 x 16 x 32 form, ragged M / N, one to forty-eight K tiles, run-to-run bit equality, and agreement with the 16 x 16 x 32 form to
    fp32 rounding (the same products, accumulated 16 deep instead of 32 deep per instruction)."""
    from neurovit_amd._cabi import lib
    A, B = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=K ** -0.5))
    bias, resid = rnd(N, seed=3), rnd(M, N, seed=4)
    ref = A.double() @ B.double().T
    Ad, Bd = dev(A), dev(B)
    lib.nv_gemm_set_tile(4, 0)
    try:
        narrow = ops.gemm(ops.NT, ops.EPI_BIAS_F32, Ad, Bd, bias=dev(bias)).clone()
        lib.nv_gemm_set_tile(11, 1)
        first = ops.gemm(ops.NT, ops.EPI_STORE_BF16, Ad, Bd)
        assert_close_bf16(first, ref, "pp32.nt_bf16")
        for _ in range(5):
            assert torch.equal(ops.gemm(ops.NT, ops.EPI_STORE_BF16, Ad, Bd), first)
        wide = ops.gemm(ops.NT, ops.EPI_BIAS_F32, Ad, Bd, bias=dev(bias))
        assert_close_f32(wide, ref + bias.double(), "pp32.bias_f32", 1e-5)
        assert rel_err(wide, narrow) < 2e-6
        assert_close_f32(ops.gemm(ops.NT, ops.EPI_STORE_F32, Ad, Bd), ref, "pp32.f32", 1e-5)
        assert_close_f32(ops.gemm(ops.NT, ops.EPI_BIAS_RESID, Ad, Bd, bias=dev(bias), aux_in=dev(resid)),
                         ref + bias.double() + resid.double(), "pp32.nt_resid", 1e-5)
        u = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        h = ops.gemm(ops.NT, ops.EPI_BIAS_GELU, Ad, Bd, bias=dev(bias), aux_out=u)
        assert_close_bf16(u, ref + bias.double(), "pp32.gelu.u")
        assert_close_bf16(h, F.gelu(ref + bias.double()), "pp32.gelu.h")
        # the other layouts keep the 16 x 16 x 32 kernels under the switch
        Bt = bf(rnd(K, N, seed=7, scale=K ** -0.5))
        assert_close_f32(ops.gemm(ops.NN, ops.EPI_STORE_F32, Ad, dev(Bt)), A.double() @ Bt.double(), "pp32.nn_f32", 1e-5)
    finally:
        lib.nv_gemm_set_tile(11, 0)
        lib.nv_gemm_set_tile(0, 0)


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (257, 264, 192), (2052, 768, 768), (520, 776, 1024), (300, 8, 128)])
def test_gemm_256x256_kernel_forced(ops, M, N, K):
    """gemm_pq.hip (256 x 256 tiles, DMA issued by the compute waves, two-stage ring, two-pass epilogue): every layout and fused
    epilogue against fp64, ragged M / N, one to sixteen K tiles, the fused column sums, run-to-run bit equality."""
    from neurovit_amd._cabi import lib
    lib.nv_gemm_set_tile(9, 0)
    try:
        A, B = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=K ** -0.5))
        bias, resid = rnd(N, seed=3), rnd(M, N, seed=4)
        ref = A.double() @ B.double().T
        Ad, Bd = dev(A), dev(B)
        first = ops.gemm(ops.NT, ops.EPI_STORE_BF16, Ad, Bd)
        assert_close_bf16(first, ref, "pq.nt_bf16")
        for _ in range(5):
            assert torch.equal(ops.gemm(ops.NT, ops.EPI_STORE_BF16, Ad, Bd), first)
        assert_close_f32(ops.gemm(ops.NT, ops.EPI_BIAS_RESID, Ad, Bd, bias=dev(bias), aux_in=dev(resid)), ref + bias.double() + resid.double(), "pq.nt_resid", 1e-5)
        u = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        h = ops.gemm(ops.NT, ops.EPI_BIAS_GELU, Ad, Bd, bias=dev(bias), aux_out=u)
        assert_close_bf16(u, ref + bias.double(), "pq.gelu.u")
        assert_close_bf16(h, F.gelu(ref + bias.double()), "pq.gelu.h")
        Bt = bf(rnd(K, N, seed=7, scale=K ** -0.5))
        refn = A.double() @ Bt.double()
        assert_close_f32(ops.gemm(ops.NN, ops.EPI_STORE_F32, Ad, dev(Bt)), refn, "pq.nn_f32", 1e-5)
        ug = bf(rnd(M, N, seed=8))
        plain = ops.gemm(ops.NN, ops.EPI_DGELU, Ad, dev(Bt), aux_in=dev(ug))
        assert_close_bf16(plain, refn * ref_cpu._gelu_grad(ug.double()), "pq.dgelu")
        fused, part = ops.gemm_dgelu_colsum(Ad, dev(Bt), dev(ug))
        assert torch.equal(plain, fused) and part.shape[0] == (M + 127) // 128
        assert_close_f32(part.sum(0), fused.double().sum(0), "pq.colsum", 1e-5)
        Kt, Mo = M, (K + 7) // 8 * 8                      # TN: reduction over a ragged token count
        At, B2 = bf(rnd(Kt, Mo, seed=9)), bf(rnd(Kt, N, seed=10, scale=Kt ** -0.5))
        c0 = rnd(Mo, N, seed=11)
        c = dev(c0.clone())
        ops.gemm(ops.TN, ops.EPI_STORE_F32, dev(At), dev(B2), out=c, accumulate=True)
        assert_close_f32(c, At.double().T @ B2.double() + c0.double(), "pq.tn", 1e-5)
        if K % 128 == 0 and N % 8 == 0:                   # fp8 operands through the same tile
            x = rnd(M, K, seed=21)
            sa = 448.0 / float(x.abs().max()) * 0.5
            w8, cs = ops.quant_rows_f8(dev(rnd(N, K, seed=22, scale=K ** -0.5)), sa)
            a8 = dev((x * sa).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8))
            ref8 = (a8.cpu().view(torch.float8_e4m3fn).float().double() @ w8.cpu().view(torch.float8_e4m3fn).float().double().T) * cs.cpu().double()
            assert_close_f32(ops.gemm_f8(ops.EPI_STORE_F32, a8, w8, cs), ref8, "pq.f8", 1e-4)
    finally:
        lib.nv_gemm_set_tile(0, 0)


def test_weight_gradient_bf16_mirror_and_cast_ranges(ops):
    """Data-parallel message path: the fp32-store epilogue also writes a bf16 copy of what it stored (single TN launch, grouped
    launch, with accumulation: the mirror is the rounded SUM), bit-equal to casting the fp32 result; nv_cast_ranges_bf16 converts the
    ranges in between (aligned and unaligned starts, tails, > 48 ranges)."""
    K, specs = 2052, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    probs, outs = [], []
    for i, (Mo, N) in enumerate(specs):
        At, B2 = dev(bf(rnd(K, Mo, seed=30 + i))), dev(bf(rnd(K, N, seed=40 + i, scale=K ** -0.5)))
        C = dev(rnd(Mo, N, seed=50 + i))
        C16 = torch.zeros((Mo, N), dtype=torch.bfloat16, device="cuda")
        probs.append((At, B2, C, True, C16))
        outs.append((At, B2, C.clone(), C, C16))
    ops.gemm_tn_grouped(probs)
    for At, B2, C0, C, C16 in outs:
        assert_close_f32(C, At.double().cpu().T @ B2.double().cpu() + C0.double().cpu(), "mirror.f32", 1e-5)
        assert torch.equal(C16, C.to(torch.bfloat16))
    At, B2 = dev(bf(rnd(520, 264, seed=61))), dev(bf(rnd(520, 136, seed=62)))
    m16 = torch.zeros((264, 136), dtype=torch.bfloat16, device="cuda")
    c = ops.gemm(ops.TN, ops.EPI_STORE_F32, At, B2, aux_out=m16)
    assert torch.equal(m16, c.to(torch.bfloat16))
    src = dev(rnd(200000, seed=63))
    dst = torch.full((200000,), 7.0, dtype=torch.bfloat16, device="cuda")
    ranges = [(0, 8), (16, 16), (24, 1000), (1003, 1010), (4096, 150001)] + [(160000 + 64 * i, 160000 + 64 * i + 1 + (i % 63)) for i in range(60)]
    ops.cast_ranges_bf16(src, dst, ranges)
    want = torch.full((200000,), 7.0, dtype=torch.bfloat16, device="cuda")
    for b, e in ranges:
        want[b:e] = src[b:e].to(torch.bfloat16)
    assert torch.equal(dst, want)


@pytest.mark.parametrize("R,n", [(4, 513), (2, 65), (8, 10), (11, 7)])
def test_skinny_linear_kernels_on_strided_rows(ops, R, n):
    """nv_skinny_nt / nv_skinny_nn (the last block's Linear layers on its cls rows, skinny.hip): rows addressed as every n-th row of
    [R * n, .] buffers, all five epilogues against fp64 and against the tiled kernels on the same rows; the rows in between stay
    untouched; R > 8 walks row slices."""
    d, m = 768, 3072
    x = bf(rnd(R * n, d, seed=1))
    W1, W2 = bf(rnd(m, d, seed=2, scale=d ** -0.5)), bf(rnd(d, m, seed=3, scale=m ** -0.5))
    b1, b2 = rnd(m, seed=4), rnd(d, seed=5)
    res = rnd(R * n, d, seed=6)
    xd, rd = dev(x), dev(res)
    # forward: h = gelu(x W1^T + b1) (bf16, with u), y = res + (h W2^T + b2) (f32)
    u = torch.full((R * n, m), 3.0, dtype=torch.bfloat16, device="cuda")
    h = torch.full((R * n, m), 3.0, dtype=torch.bfloat16, device="cuda")
    ops.skinny_nt(1, xd[::n], dev(W1), dev(b1), h[::n], u_out=u[::n])
    uref = x[::n].double() @ W1.double().T + b1.double()
    assert_close_bf16(u[::n], uref, "sk.u")
    assert_close_bf16(h[::n], F.gelu(uref), "sk.h")
    keep = torch.ones(R * n, dtype=torch.bool); keep[::n] = False
    assert bool((h.cpu()[keep] == 3.0).all()) and bool((u.cpu()[keep] == 3.0).all())
    y = torch.zeros((R * n, d), device="cuda")
    ops.skinny_nt(0, h[::n], dev(W2), dev(b2), y[::n], resid=rd[::n])
    yref = res[::n].double() + (h[::n].cpu().double() @ W2.double().T + b2.double())
    assert_close_f32(y[::n], yref, "sk.resid", 1e-5)
    tiled = ops.gemm(ops.NT, ops.EPI_BIAS_RESID, h[::n].contiguous(), dev(W2), bias=dev(b2), aux_in=rd[::n].contiguous())
    assert rel_l2(y[::n], tiled) < 1e-6
    # backward: dU = (g W2) * gelu'(u) (bf16, + column sums), dxn = dU W1 (f32), dAO-like bf16 store
    g = bf(rnd(R * n, d, seed=7))
    gd = dev(g)
    du = torch.full((R * n, m), 5.0, dtype=torch.bfloat16, device="cuda")
    dcol = torch.ones(m, device="cuda") if R <= 4 else None
    ops.skinny_nn(0, gd[::n], dev(W2), du[::n], u=u[::n], dcol=dcol, accumulate=True)
    duref = (g[::n].double() @ W2.double()) * ref_cpu._gelu_grad(u[::n].cpu().double())
    assert_close_bf16(du[::n], duref, "sk.du")
    assert bool((du.cpu()[keep] == 5.0).all())
    if dcol is not None:
        assert_close_f32(dcol, 1.0 + du[::n].cpu().double().sum(0), "sk.dcol", 1e-5)
    dxn = torch.zeros((R * n, d), device="cuda")
    ops.skinny_nn(1, du[::n], dev(W1), dxn[::n])
    assert_close_f32(dxn[::n], du[::n].cpu().double() @ W1.double(), "sk.dxn", 1e-5)
    o16 = torch.zeros((R * n, m), dtype=torch.bfloat16, device="cuda")
    ops.skinny_nn(2, gd[::n], dev(W2), o16[::n])
    assert_close_bf16(o16[::n], g[::n].double() @ W2.double(), "sk.bf16")
    again = torch.zeros_like(o16)
    ops.skinny_nn(2, gd[::n], dev(W2), again[::n])
    assert torch.equal(again, o16)


def test_skinny_linear_kernels_carry_the_dropout_masks_of_the_dense_tensors(ops):
    """The cls rows of the last block under dropout: nv_skinny_nt / nv_skinny_nn on every n-th row of a dense [B * n, .] tensor must apply the SAME
    nn.Dropout mask the tiled GEMM epilogues apply to those rows of the dense tensor (the element offset of the strided view is the dense element
    index): all three dropout sites of a block - out-projection / FC2 (bias + residual), FC1 (GELU), and dU in the backward pass - against the
    tiled kernels run on the whole dense problem with the same seed: same zeros, same values."""
    R, n, d, m, p = 4, 513, 768, 3072, 0.25
    M = R * n
    x = dev(bf(rnd(M, d, seed=1)))
    W1, W2 = dev(bf(rnd(m, d, seed=2, scale=d ** -0.5))), dev(bf(rnd(d, m, seed=3, scale=m ** -0.5)))
    b1, b2 = dev(rnd(m, seed=4)), dev(rnd(d, seed=5))
    res = dev(rnd(M, d, seed=6))
    # FC1 + GELU + dropout
    u_t = torch.empty((M, m), dtype=torch.bfloat16, device="cuda")
    h_t = ops.gemm(ops.NT, ops.EPI_BIAS_GELU, x, W1, bias=b1, aux_out=u_t, drop_seed=11, drop_p=p)
    h_s = torch.zeros((M, m), dtype=torch.bfloat16, device="cuda"); u_s = torch.zeros_like(h_s)
    ops.skinny_nt(1, x[::n], W1, b1, h_s[::n], u_out=u_s[::n], drop_seed=11, drop_p=p)
    zeros_t = h_t[::n] == 0
    assert 0.2 < float(zeros_t.float().mean()) < 0.3 and torch.equal(zeros_t, h_s[::n] == 0)
    assert rel_l2(h_s[::n].float(), h_t[::n].float()) < 4e-3 and rel_l2(u_s[::n].float(), u_t[::n].float()) < 4e-3
    # FC2 + bias + dropout + residual
    y_t = ops.gemm(ops.NT, ops.EPI_BIAS_RESID, h_t, W2, bias=b2, aux_in=res, drop_seed=12, drop_p=p)
    y_s = torch.zeros((M, d), device="cuda")
    ops.skinny_nt(0, h_t[::n], W2, b2, y_s[::n], resid=res[::n], drop_seed=12, drop_p=p)
    dropped_t = y_t[::n] == res[::n]
    assert 0.2 < float(dropped_t.float().mean()) < 0.3 and torch.equal(dropped_t, y_s[::n] == res[::n])
    assert rel_l2(y_s[::n], y_t[::n]) < 1e-5
    # dU = (g W2 * mask) * gelu'(u)
    g = dev(bf(rnd(M, d, seed=7)))
    du_t = ops.gemm(ops.NN, ops.EPI_DGELU, g, W2, aux_in=u_t, drop_seed=11, drop_p=p)
    du_s = torch.zeros((M, m), dtype=torch.bfloat16, device="cuda")
    ops.skinny_nn(0, g[::n], W2, du_s[::n], u=u_t[::n], drop_seed=11, drop_p=p)
    assert torch.equal(du_t[::n] == 0, du_s[::n] == 0) and torch.equal(du_t[::n] == 0, zeros_t | (du_t[::n] == 0))
    assert rel_l2(du_s[::n].float(), du_t[::n].float()) < 4e-3
    report("skinny cls-row kernels under dropout 0.25: masks identical to the tiled epilogues on the dense tensors (FC1, FC2 + residual, dU)")


def test_gemm_ping_pong_grouped_equals_single_launches(ops):
    """Grouped weight gradients on the ping-pong tiles: bit-identical to single launches of the same kernel."""
    from neurovit_amd._cabi import lib
    K = 2052
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    probs, singles = [], []
    lib.nv_gemm_set_tile(4, 0)
    try:
        for i, (Mo, N) in enumerate(shapes):
            At, B2 = dev(bf(rnd(K, Mo, seed=20 + i))), dev(bf(rnd(K, N, seed=30 + i, scale=K ** -0.5)))
            acc = i % 2 == 1
            base = dev(rnd(Mo, N, seed=40 + i)) if acc else torch.empty(Mo, N, device="cuda")
            singles.append(ops.gemm(ops.TN, ops.EPI_STORE_F32, At, B2, out=base.clone(), accumulate=acc))
            probs.append((At, B2, base.clone(), acc))
        ops.gemm_tn_grouped(probs)
        for (At, B2, C1, acc), ref in zip(probs, singles):
            assert torch.equal(C1, ref)
        assert_close_f32(probs[0][2], probs[0][0].double().T @ probs[0][1].double(), "pp.grouped", 1e-5)
    finally:
        lib.nv_gemm_set_tile(0, 0)


@pytest.mark.parametrize("tile", [0, 1, 3, 4])
def test_gemm_dgelu_fused_colsum_and_reduce_multi(ops, tile):
    """Epilogue 6: dU as epilogue 5 (bit-identical) plus per-tile column sums of the stored bf16 values; nv_reduce_multi sums the
    partial rows (and other jobs in the same launch) deterministically.  Every large-tile kernel family."""
    from neurovit_amd._cabi import lib
    M, N, K = 2052, 3072 if tile in (0, 4) else 392, 768
    A, Bt = bf(rnd(M, K, seed=6)), bf(rnd(K, N, seed=7, scale=K ** -0.5))
    u = bf(rnd(M, N, seed=8))
    lib.nv_gemm_set_tile(tile, 0)
    try:
        plain = ops.gemm(ops.NN, ops.EPI_DGELU, dev(A), dev(Bt), aux_in=dev(u))
        fused, part = ops.gemm_dgelu_colsum(dev(A), dev(Bt), dev(u))
        fused2, part2 = ops.gemm_dgelu_colsum(dev(A), dev(Bt), dev(u))
    finally:
        lib.nv_gemm_set_tile(0, 0)
    assert torch.equal(plain, fused) and torch.equal(fused, fused2) and torch.equal(part, part2)
    want = fused.double().sum(0)
    db = torch.full((N,), 7.0, device="cuda")
    other = dev(rnd(40, 3 * 64, seed=9))
    o0, o2 = torch.zeros(64, device="cuda"), torch.ones(64, device="cuda")
    ops.reduce_multi([(part, N, [db], False), (other, 64, [o0, None, o2], True)])
    assert_close_f32(db, want, "colsum", 1e-5)
    assert_close_f32(o0, other[:, :64].double().sum(0), "seg0", 1e-5)
    assert_close_f32(o2, other[:, 128:].double().sum(0) + 1.0, "seg2 accumulate", 1e-5)
    ops.reduce_multi([(part, N, [db], True)])
    assert_close_f32(db, 2 * want, "colsum accumulate", 1e-5)


def _e4m3(t):
    """fp32 -> OCP e4m3 (round to nearest even, saturating) on the CPU: (bytes, dequantised fp32)."""
    q = t.clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), q.float()


@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (2052, 2304, 768), (520, 392, 1024), (300, 8, 256)])
def test_gemm_fp8_kernel(ops, M, N, K):
    """fp8 (OCP e4m3) GEMM on v_mfma_scale_f32_16x16x128_f8f6f4: exact integer-free check - the operands are exactly representable,
    products are accumulated in fp32, so the result must match the fp64 product of the DEQUANTISED operands to fp32 rounding;
    row quantisation and the LayerNorm -> fp8 kernel are checked byte for byte against torch's e4m3 conversion."""
    W = rnd(N, K, seed=2, scale=K ** -0.5)
    x = rnd(M, K, seed=1)
    sa = 448.0 / float(x.abs().max()) * 0.5
    w8, cs = ops.quant_rows_f8(dev(W), sa)
    sw = 448.0 / W.abs().amax(dim=1, keepdim=True)
    ref8, wdq = _e4m3(W * sw)
    assert (w8.cpu().to(torch.int16) - ref8.to(torch.int16)).abs().max().item() <= 1          # reciprocal rounding may move a tie
    assert (w8.cpu() != ref8).float().mean().item() < 1e-3
    assert_close_f32(cs, (1.0 / (sa * sw)).reshape(-1).double(), "colscale", 1e-6)
    a8, adq = _e4m3(x * sa)
    wdq_dev = w8.cpu().view(torch.float8_e4m3fn).float()                                     # what the device actually multiplies
    ref = (adq.double() @ wdq_dev.double().T) * cs.cpu().double()
    bias, resid = rnd(N, seed=3), rnd(M, N, seed=4)
    A8 = dev(a8)
    assert_close_f32(ops.gemm_f8(ops.EPI_STORE_F32, A8, w8, cs), ref, "f8.store_f32", 1e-4)   # the K = 128 MFMA sums its 128 products with fewer guard bits than an fp32 fma chain: 2e-5 measured
    assert_close_bf16(ops.gemm_f8(ops.EPI_STORE_BF16, A8, w8, cs), ref, "f8.store_bf16")
    assert_close_f32(ops.gemm_f8(ops.EPI_BIAS_RESID, A8, w8, cs, bias=dev(bias), aux_in=dev(resid)), ref + bias.double() + resid.double(), "f8.resid", 1e-4)
    h8 = ops.gemm_f8(ops.EPI_BIAS_GELU_F8, A8, w8, cs, bias=dev(bias), out_scale=16.0)
    _, want = _e4m3((F.gelu(ref + bias.double()) * 16.0).float())
    got = h8.cpu().view(torch.float8_e4m3fn).float()
    step = torch.maximum(want.abs(), torch.tensor(2.0 ** -6)) * 0.126 + 2.0 ** -9            # one e4m3 code step (3 mantissa bits)
    assert ((got - want).abs() <= step).all()                                                # -0 == +0; at most one step at rounding ties
    assert (got != want).float().mean().item() < 2e-2
    first = ops.gemm_f8(ops.EPI_STORE_F32, A8, w8, cs)
    for _ in range(3):
        assert torch.equal(ops.gemm_f8(ops.EPI_STORE_F32, A8, w8, cs), first)


def test_ln_fwd_fp8(ops):
    M, d = 2052, 1024
    x, gamma, beta = rnd(M, d, seed=1) * 2 + 0.3, 1 + 0.1 * rnd(d, seed=2), 0.1 * rnd(d, seed=3)
    y8 = ops.ln_fwd_f8(dev(x), dev(gamma), dev(beta), 32.0)
    ref = F.layer_norm(x.double(), (d,), gamma.double(), beta.double(), 1e-5) * 32.0
    _, want = _e4m3(ref.float())
    got = y8.cpu().view(torch.float8_e4m3fn).float()
    step = torch.maximum(want.abs(), torch.tensor(2.0 ** -6)) * 0.126 + 2.0 ** -9
    assert ((got - want).abs() <= step).all() and (got != want).float().mean().item() < 1e-2


def test_gemm_rejects_bad_args(ops):
    A, B = dev(bf(rnd(16, 12))), dev(bf(rnd(8, 12)))
    with pytest.raises((RuntimeError, AssertionError)):
        ops.gemm(ops.NT, ops.EPI_STORE_BF16, A, B)          # K = 12 not a multiple of 8
    with pytest.raises(RuntimeError):
        ops.gemm(ops.NT, ops.EPI_STORE_BF16, A.cpu(), B.cpu())


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("M,d", [(65, 192), (130, 128), (2052, 768), (9, 1024), (33, 2048)])
def test_ln_fwd_bwd(ops, M, d):
    x, gamma, beta = rnd(M, d, seed=1) * 2 + 0.3, 1 + 0.1 * rnd(d, seed=2), 0.1 * rnd(d, seed=3)
    y, st = ops.ln_fwd(dev(x), dev(gamma), dev(beta))
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = F.layer_norm(xd, (d,), gd, bd, 1e-5)
    assert_close_bf16(y, ref, "ln_fwd")
    assert_close_f32(st[0], x.double().mean(1), "mean", 1e-5)
    assert_close_f32(st[1], 1 / torch.sqrt(x.double().var(1, unbiased=False) + 1e-5), "rstd", 1e-5)
    dy, g_in = rnd(M, d, seed=4), rnd(M, d, seed=5)
    ref.backward(dy.double())
    g_out, g16, dg, db, dc = ops.ln_bwd(dev(dy), dev(x), st, dev(gamma), g_in=dev(g_in.clone()))
    assert_close_f32(g_out, xd.grad + g_in.double(), "ln_bwd.dx", 1e-5)
    assert_close_bf16(g16, xd.grad + g_in.double(), "ln_bwd.g16")
    assert_close_f32(dg, gd.grad, "ln_bwd.dgamma", 1e-5)
    assert_close_f32(db, bd.grad, "ln_bwd.dbeta", 1e-5)
    assert_close_f32(dc, (xd.grad + g_in.double()).sum(0), "ln_bwd.colsum", 1e-5)


# ------------------------------------------------------------------------------------------ patch gather (G1 bit exact) + LN(P)
def _balanced_pm1(B, S, p, seed):
    """Volume of +-1 whose every p^3 patch has exactly as many +1 as -1: LN(eps=0) maps it onto itself
    exactly, so the kernel's bf16 output IS the gathered voxel -> bit-exact test of the index map."""
    idx = torch.from_numpy(ref_cpu.patch_index_map(S, p))            # [N, P]
    g = torch.Generator().manual_seed(seed)
    vol = torch.empty(B, S ** 3)
    half = torch.cat([torch.ones(p ** 3 // 2), -torch.ones(p ** 3 - p ** 3 // 2)])
    for b in range(B):
        for n in range(idx.shape[0]):
            vol[b, idx[n]] = half[torch.randperm(p ** 3, generator=g)]
    return vol.reshape(B, S, S, S)


@pytest.mark.parametrize("S,p", [(32, 8), (64, 16), (16, 4)])
def test_patch_gather_bit_exact(ops, S, p):
    B = 2
    vol = _balanced_pm1(B, S, p, seed=S + p)
    P = p ** 3
    out, st = ops.patch_ln_fwd(ref_cpu.fmri_to_video(dev(vol)), p, p, p, dev(torch.ones(P)), dev(torch.zeros(P)), eps=0.0)
    tok = ref_cpu.patchify(ref_cpu.fmri_to_video(vol), p, p, p).reshape(-1, P)
    assert torch.equal(out.float().cpu(), tok), "patch index map is not bit exact"
    assert torch.equal(st[0].cpu(), torch.zeros_like(st[0].cpu())) and torch.equal(st[1].cpu(), torch.ones_like(st[1].cpu()))


@pytest.mark.parametrize("S,p,B", [(128, 16, 2), (128, 8, 1), (64, 16, 3), (32, 8, 2), (96, 12, 1)])
def test_patch_slabs_equal_the_per_token_gather_bitwise(ops, S, p, B):
    """K1's two forms (nv_patch_set_mode): a wave per token gathering its 64-byte runs (the default: faster) against a workgroup per patch column staging
    (F, H, W) slabs through LDS with coalesced reads (mode 2): the same rows, statistics included, bit for bit, for 16-bit and fp32 tokens and for raw volumes
    (vol_sigma)."""
    from neurovit_amd._cabi import lib
    g = torch.Generator().manual_seed(S + p)
    vol = torch.randn(B, S, S, S, generator=g) * 3.0 + 1.5
    P = p ** 3
    gamma, beta = dev(torch.randn(P, generator=g)), dev(torch.randn(P, generator=g))
    sig = dev(torch.rand(B, generator=g) + 0.5)
    video = ref_cpu.fmri_to_video(dev(vol))
    res = {}
    for mode in (0, 2):
        lib.nv_patch_set_mode(mode)
        try:
            res[mode] = (ops.patch_ln_fwd(video, p, p, p, gamma, beta), ops.patch_ln_fwd_f32(video, p, p, p, gamma, beta),
                         ops.patch_ln_fwd(video, p, p, p, gamma, beta, vol_sigma=sig))
        finally:
            lib.nv_patch_set_mode(0)
    for (o0, s0), (o1, s1) in zip(res[0], res[2]):
        assert torch.equal(o0, o1) and torch.equal(s0, s1)
    # ... and against the oracle's patchify + LayerNorm (fp32 tokens)
    tok = ref_cpu.patchify(ref_cpu.fmri_to_video(vol), p, p, p).reshape(-1, P)
    want = torch.nn.functional.layer_norm(tok, (P,), gamma.cpu(), beta.cpu(), 1e-5)
    assert rel_err(res[0][1][0].cpu(), want) < 1e-5


def test_patch_gather_matches_golden_index_map(ops, golden):
    """Same check against the fixture produced by the reference's own Rearrange (S=32, p=8)."""
    S, p = 32, 8
    idx = torch.from_numpy(golden("patchify.npz")[f"tok_S{S}_p{p}"].astype(np.int64))
    vol = _balanced_pm1(1, S, p, seed=5)
    out, _ = ops.patch_ln_fwd(ref_cpu.fmri_to_video(dev(vol)), p, p, p, dev(torch.ones(p ** 3)), dev(torch.zeros(p ** 3)), eps=0.0)
    assert torch.equal(out.float().cpu(), vol.reshape(-1)[idx])


@pytest.mark.parametrize("S,p,C", [(32, 8, 1), (18, 9, 1), (16, 8, 3)])
def test_patch_ln_fwd_bwd_general(ops, S, p, C):
    """Scalar path: odd patch size (p=9, P=729 -> padded to 736) and multi-channel video."""
    B = 2
    g = torch.Generator().manual_seed(3)
    video = torch.randn(B, C, S, S, S, generator=g)
    P = C * p ** 3
    gamma, beta = 1 + 0.1 * rnd(P, seed=1), 0.1 * rnd(P, seed=2)
    out, st = ops.patch_ln_fwd(dev(video), p, p, p, dev(gamma), dev(beta))
    tok = ref_cpu.patchify(video, p, p, p).reshape(-1, P).double()
    ref = F.layer_norm(tok, (P,), gamma.double(), beta.double(), 1e-5)
    assert out.shape[1] == (P + 7) // 8 * 8
    assert_close_bf16(out[:, :P], ref, "patch_ln")
    assert torch.count_nonzero(out[:, P:]) == 0
    dxp = rnd(tok.shape[0], out.shape[1], seed=4)
    dg, db = ops.patch_ln_bwd(dev(video), p, p, p, dev(dxp), st)
    xh = (tok - tok.mean(1, keepdim=True)) / torch.sqrt(tok.var(1, unbiased=False, keepdim=True) + 1e-5)
    assert_close_f32(dg, (dxp[:, :P].double() * xh).sum(0), "patch_ln.dgamma", 1e-4)
    assert_close_f32(db, dxp[:, :P].double().sum(0), "patch_ln.dbeta", 1e-4)


# ------------------------------------------------------------------------------------------ embed finish (A4+A5)
def test_embed_finish_scatter_bit_exact(ops):
    """cls/pos scatter (G1): with eps=0 and balanced +-1 rows LN is the identity, so every output row must
    equal (+-1 + pos[i]) / (cls + pos[0]) bit for bit."""
    B, N, d = 3, 27, 64
    g = torch.Generator().manual_seed(0)
    t = torch.stack([torch.cat([torch.ones(d // 2), -torch.ones(d // 2)])[torch.randperm(d, generator=g)] for _ in range(B * N)])
    pos, cls = rnd(N + 1, d, seed=1), rnd(d, seed=2)
    x, st = ops.embed_finish_fwd(dev(t), B, N, dev(torch.ones(d)), dev(torch.zeros(d)), dev(pos), dev(cls), eps=0.0)
    x = x.cpu()
    assert torch.equal(x[:, 0], (cls + pos[0]).expand(B, -1))
    assert torch.equal(x[:, 1:], t.reshape(B, N, d) + pos[1:])


def test_embed_finish_fwd_bwd(ops):
    B, N, d = 2, 64, 192
    t, gamma, beta = rnd(B * N, d, seed=1), 1 + 0.1 * rnd(d, seed=2), 0.1 * rnd(d, seed=3)
    pos, cls = rnd(N + 1, d, seed=4), rnd(d, seed=5)
    x, st = ops.embed_finish_fwd(dev(t), B, N, dev(gamma), dev(beta), dev(pos), dev(cls))
    td = t.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    pd, cd = pos.double().requires_grad_(True), cls.double().requires_grad_(True)
    a4 = F.layer_norm(td, (d,), gd, bd, 1e-5).reshape(B, N, d)
    ref = torch.cat((cd.expand(B, 1, d), a4), dim=1) + pd
    assert_close_f32(x, ref, "embed_finish", 1e-5)
    g = rnd(B, N + 1, d, seed=6)
    ref.backward(g.double())
    dt, dt16, dg, db, dbias, dpos, dcls = ops.embed_finish_bwd(dev(g), dev(t), st, dev(gamma), B, N)
    assert_close_f32(dt, td.grad, "dt", 1e-5)
    assert_close_bf16(dt16, td.grad, "dt16")
    assert_close_f32(dg, gd.grad, "dgamma", 1e-5)
    assert_close_f32(db, bd.grad, "dbeta", 1e-5)
    assert_close_f32(dbias, td.grad.sum(0), "dbias", 1e-5)
    assert_close_f32(dpos, pd.grad, "dpos", 1e-5)
    assert_close_f32(dcls, cd.grad, "dcls", 1e-5)


# ------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,n,heads", [(2, 65, 3), (1, 513, 2), (2, 9, 1), (1, 130, 2), (1, 64, 1)])
def test_attention_fwd_bwd(ops, B, n, heads):
    dh, inner = 64, heads * 64
    qkv = bf(rnd(B * n, 3 * inner, seed=n)).float()
    q, k, v = (t.reshape(B, n, heads, dh).permute(0, 2, 1, 3).clone().requires_grad_(True) for t in qkv.chunk(3, dim=-1))
    ref = ref_cpu._AttnEmu.apply(q, k, v, dh ** -0.5)
    out, lse = ops.attn_fwd(dev(bf(qkv)), B, n, heads)
    ref2 = ref.permute(0, 2, 1, 3).reshape(B * n, inner)
    assert_close_stat(out, ref2, "attn.out")
    s = torch.matmul(q.double(), k.double().transpose(-1, -2)) * dh ** -0.5
    assert_close_f32(lse, torch.logsumexp(s, dim=-1), "attn.lse", 1e-4)
    do = bf(rnd(B * n, inner, seed=7)).float()
    ref.backward(do.reshape(B, n, heads, dh).permute(0, 2, 1, 3))
    # feed the kernel the oracle's own bf16 forward output so both sides use the same delta = rowsum(dO * O)
    dqkv, delta = ops.attn_bwd(dev(bf(qkv)), dev(bf(ref2.detach())), dev(bf(do)), lse, B, n, heads)
    dref = torch.cat([t.grad.permute(0, 2, 1, 3).reshape(B * n, inner) for t in (q, k, v)], dim=-1)
    assert_close_stat(dqkv, dref, "attn.dqkv")


@pytest.mark.parametrize("B,n,heads,dh,drop", [(2, 65, 3, 32, 0.0), (1, 200, 2, 128, 0.0), (2, 37, 2, 48, 0.0), (1, 130, 2, 32, 0.2), (1, 70, 1, 128, 0.1),
                                                 (1, 513, 2, 8, 0.0)])
def test_attention_other_head_dims(ops, B, n, heads, dh, drop):
    """dim_head != 64 (vit_3d.py:29 takes any; attention_generic.hip: scalar kernels, exact row maximum): forward, LSE and the three
    gradients against an fp64 attention on the same bf16 inputs, with the product's dropout mask regenerated by the oracle.
    Tolerance 4e-3 rel l2: outputs and the P / dS operands are rounded to bf16 (2^-9 each) as in the MFMA kernels."""
    inner, seed = heads * dh, 4242
    qkv = bf(rnd(B * n, 3 * inner, seed=n + dh)).float()
    q, k, v = (t.reshape(B, n, heads, dh).permute(0, 2, 1, 3).double().clone().requires_grad_(True) for t in qkv.chunk(3, dim=-1))
    mask = ref_cpu.attn_drop_mask(seed, drop, B, heads, n).double() if drop else None
    s = torch.matmul(q, k.transpose(-1, -2)) * dh ** -0.5
    p = torch.softmax(s, dim=-1)
    ref = torch.matmul(p if mask is None else p * mask, v)
    out, lse = ops.attn_fwd(dev(bf(qkv)), B, n, heads, dh, drop_seed=seed if drop else 0, drop_p=drop)
    ref2 = ref.permute(0, 2, 1, 3).reshape(B * n, inner)
    assert rel_l2(out.float(), ref2.detach()) < 4e-3
    assert_close_f32(lse, torch.logsumexp(s, dim=-1).detach(), "gen.lse", 1e-4)
    do = bf(rnd(B * n, inner, seed=7)).float()
    ref.backward(do.double().reshape(B, n, heads, dh).permute(0, 2, 1, 3))
    dqkv, delta = ops.attn_bwd(dev(bf(qkv)), out, dev(bf(do)), lse, B, n, heads, dh, drop_seed=seed if drop else 0, drop_p=drop)
    for name, t, lo in (("dq", q, 0), ("dk", k, inner), ("dv", v, 2 * inner)):
        g = t.grad.permute(0, 2, 1, 3).reshape(B * n, inner)
        assert rel_l2(dqkv[:, lo:lo + inner].float(), g) < 6e-3, name       # + the bf16 forward output inside delta
    assert_close_f32(delta.reshape(B, heads, n), (do.reshape(B, n, heads, dh).permute(0, 2, 1, 3).double() * out.float().cpu().reshape(B, n, heads, dh).permute(0, 2, 1, 3).double()).sum(-1),
                     "gen.delta", 1e-4)
    # run-to-run bit equality (fixed reduction order)
    out2, _ = ops.attn_fwd(dev(bf(qkv)), B, n, heads, dh, drop_seed=seed if drop else 0, drop_p=drop)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("B,n,heads,drop", [(2, 513, 3, 0.0), (1, 576, 2, 0.0), (3, 65, 2, 0.1), (2, 9, 1, 0.0), (1, 500, 1, 0.2)])
def test_attention_resident_equals_streaming(ops, B, n, heads, drop):
    """n <= 576: the LDS-resident kernels (K/V - or Q/dO - of one head loaded once) must reproduce the streaming kernels
    bit for bit: same tile order, same arithmetic, same dropout counters."""
    from neurovit_amd._cabi import lib
    inner = heads * 64
    qkv = dev(bf(rnd(B * n, 3 * inner, seed=n + heads)))
    do = dev(bf(rnd(B * n, inner, seed=3)))
    res = {}
    try:
        # streaming; resident (default: one wave per row group); resident with two partner waves; resident backward as ONE launch
        # whose dK/dV workgroups compute delta themselves (+100)
        for mode in (1, 2, 22, 102):
            lib.nv_attn_set_mode(mode)
            out, lse = ops.attn_fwd(qkv, B, n, heads, drop_seed=99, drop_p=drop)
            dqkv, delta = ops.attn_bwd(qkv, out, do, lse, B, n, heads, drop_seed=99, drop_p=drop)
            res[mode] = (out, lse, dqkv, delta)
    finally:
        lib.nv_attn_set_mode(0)
    for other in (2, 22, 102):
        for a, b, name in zip(res[1], res[other], ("out", "lse", "dqkv", "delta")):
            assert torch.equal(a, b), (other, name)
    assert torch.isfinite(res[2][2].float()).all()


@pytest.mark.parametrize("B,n,heads,mode", [(2, 513, 2, 2), (1, 700, 2, 1), (2, 1001, 2, 3), (20, 513, 12, 0)])
def test_attention_forward_fp8_output(ops, B, n, heads, mode):
    """nv_attn_fwd_o8 (the fp8 path's out-projection operand written by the attention kernel itself): every forward kernel
    (2 resident, 1 streaming, 3 wide, 0 the dispatcher's choice) - the e4m3 bytes decode to the bf16 output within e4m3's rounding
    (2^-4 relative, half an ulp of the 3-bit mantissa) plus the bf16 rounding of the comparison value; saturating scale handled."""
    from neurovit_amd._cabi import lib
    qkv = dev(bf(rnd(B * n, 3 * heads * 64, seed=n + heads)))
    lib.nv_attn_set_mode(mode)
    try:
        ref, _ = ops.attn_fwd(qkv, B, n, heads)
        scale = 448.0 / (2.0 * float(ref.float().abs().max()))
        o8 = ops.attn_fwd_o8(qkv, B, n, heads, scale)
    finally:
        lib.nv_attn_set_mode(0)
    deq = o8.view(torch.float8_e4m3fn).float() / scale
    r = ref.float()
    tol = r.abs() * (2.0 ** -4 + 2.0 ** -8) + 2.0 ** -9 / scale * 2      # relative rounding + the subnormal floor of e4m3 at this scale
    assert ((deq - r).abs() <= tol).all(), float(((deq - r).abs() - tol).max())
    assert float((deq - r).norm() / r.norm()) < 0.04                        # mean relative error of a 3-bit mantissa: ~2.5 %


@pytest.mark.parametrize("B,n,heads,drop", [(2, 1001, 2, 0.0), (1, 4097, 1, 0.0), (20, 513, 12, 0.0), (3, 700, 2, 0.1), (1, 129, 1, 0.0), (2, 65, 3, 0.2)])
def test_attention_wide_forward_equals_streaming(ops, B, n, heads, drop):
    """The wide streaming forward (32 query rows per wave, K / V tiles by LDS-DMA into a two-stage ring, one barrier per tile)
    must reproduce the reference streaming kernel (see the assertions: equal up to isolated one-ulp contraction differences) on long sequences (n = 1001 of the
    reference default, n = 4097 of ViT3D-large), the 4D batch (20 x 12 heads at n = 513), ragged last tiles and dropout."""
    from neurovit_amd._cabi import lib
    inner = heads * 64
    qkv = dev(bf(rnd(B * n, 3 * inner, seed=n + heads)))
    try:
        lib.nv_attn_set_mode(1)
        out1, lse1 = ops.attn_fwd(qkv, B, n, heads, drop_seed=7, drop_p=drop)
        lib.nv_attn_set_mode(3)                   # force the wide kernel
        out3, lse3 = ops.attn_fwd(qkv, B, n, heads, drop_seed=7, drop_p=drop)
        out3b, _ = ops.attn_fwd(qkv, B, n, heads, drop_seed=7, drop_p=drop)
    finally:
        lib.nv_attn_set_mode(0)
    assert torch.equal(out3, out3b)                          # deterministic
    # same tile order and arithmetic; what may differ is how the compiler contracts a few fp32 mul + add pairs in the two kernels:
    # the log-sum-exp to one ulp, and an isolated bf16 output element by one ulp (measured: 1 element in 262 144)
    assert (lse1 - lse3).abs().max().item() <= 2e-6
    diff = (out1.float() - out3.float()).abs()
    assert (diff > 0).float().mean().item() <= 2e-5 and (diff <= out1.float().abs() * 2 ** -7 + 1e-30).all()


@pytest.mark.parametrize("B,n,heads,drop", [(2, 1001, 2, 0.0), (1, 4097, 1, 0.0), (3, 700, 2, 0.1), (1, 129, 1, 0.0), (2, 65, 3, 0.2)])
def test_attention_wide_backward_equals_streaming(ops, B, n, heads, drop):
    """Wide streaming backward kernels (32 query rows / keys per wave, LDS-DMA ring; lse and delta tiles by dword LDS-DMA, padded
    query rows as zeros instead of +inf): dQ, dK, dV and delta bit for bit equal to the reference streaming kernels."""
    from neurovit_amd._cabi import lib
    inner = heads * 64
    qkv = dev(bf(rnd(B * n, 3 * inner, seed=n + heads)))
    do = dev(bf(rnd(B * n, inner, seed=3)))
    try:
        lib.nv_attn_set_mode(1)
        out, lse = ops.attn_fwd(qkv, B, n, heads, drop_seed=11, drop_p=drop)
        d1, delta1 = ops.attn_bwd(qkv, out, do, lse, B, n, heads, drop_seed=11, drop_p=drop)
        lib.nv_attn_set_mode(3)
        d3, delta3 = ops.attn_bwd(qkv, out, do, lse, B, n, heads, drop_seed=11, drop_p=drop)
    finally:
        lib.nv_attn_set_mode(0)
    assert torch.equal(d1, d3) and torch.equal(delta1, delta3)
    assert torch.isfinite(d3.float()).all()


def test_attention_rescale_branch(ops):
    """Force the online-softmax rescale: one key in the LAST tile dominates one query row."""
    B, n, heads, dh = 1, 200, 1, 64
    qkv = bf(rnd(B * n, 3 * dh, seed=1) * 0.5).float()
    qkv[5, :dh] = 4.0
    qkv[190, dh:2 * dh] = 4.0                      # q5 . k190 = 64*16 -> scaled 128, far above everything else
    q, k, v = (t.reshape(B, n, heads, dh).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1))
    ref = ref_cpu._AttnEmu.apply(q, k, v, dh ** -0.5).permute(0, 2, 1, 3).reshape(B * n, dh)
    out, _ = ops.attn_fwd(dev(bf(qkv)), B, n, heads)
    assert_close_stat(out, ref, "attn.rescale")
    assert torch.isfinite(out.float()).all()


# ------------------------------------------------------------------------------------------ head, colsum, CE, AdamW, cast
def test_head_fwd_bwd(ops):
    B, n, d, C = 3, 9, 192, 5
    x, gamma, beta = rnd(B, n, d, seed=1), 1 + 0.1 * rnd(d, seed=2), 0.1 * rnd(d, seed=3)
    Wt, bias = rnd(C, d, seed=4, scale=d ** -0.5), 0.1 * rnd(C, seed=5)
    logits, xh, st = ops.head_fwd(dev(x), dev(gamma), dev(beta), dev(Wt), dev(bias))
    xd = x.double().requires_grad_(True)
    gd, bd, wd, cd = (t.double().requires_grad_(True) for t in (gamma, beta, Wt, bias))
    ref = F.linear(F.layer_norm(xd[:, 0], (d,), gd, bd, 1e-5), wd, cd)
    assert_close_f32(logits, ref, "head.logits", 1e-5)
    dl = rnd(B, C, seed=6)
    ref.backward(dl.double())
    g, g16, dgm, dbt, dW, db, dcol = ops.head_bwd(dev(dl), dev(Wt), dev(x), st, xh, dev(gamma))
    assert_close_f32(g, xd.grad, "head.g", 1e-5)
    assert torch.count_nonzero(g[:, 1:]) == 0
    assert_close_f32(dgm, gd.grad, "head.dgamma", 1e-5)
    assert_close_f32(dbt, bd.grad, "head.dbeta", 1e-5)
    assert_close_f32(dW, wd.grad, "head.dW", 1e-5)
    assert_close_f32(db, cd.grad, "head.db", 1e-5)
    assert_close_f32(dcol, xd.grad.sum((0, 1)), "head.colsum", 1e-5)


def test_colsum_and_cast(ops):
    X = bf(rnd(2052, 384, seed=1))
    assert_close_f32(ops.colsum_bf16(dev(X)), X.double().sum(0), "colsum", 1e-5)
    src = rnd(37, 729, seed=2)
    dst = ops.cast_bf16(dev(src), 736).cpu()
    assert torch.equal(dst[:, :729], bf(src)) and torch.count_nonzero(dst[:, 729:]) == 0


def test_ce_loss(ops):
    logits, target = rnd(4, 7, seed=1) * 3, torch.tensor([0, 6, 3, 3])
    ld = logits.double().requires_grad_(True)
    ref = F.cross_entropy(ld, target)
    ref.backward()
    loss, dl = ops.ce_loss(dev(logits), dev(target))
    assert abs(loss.item() - ref.item()) < 1e-6 * max(1, abs(ref.item()))
    assert_close_f32(dl, ld.grad, "ce.grad", 1e-5)
    assert abs(loss.item() - train_step.cross_entropy(logits, target).item()) < 1e-5


def test_adamw_matches_oracle(ops):
    n = 4096 + 8
    p0, g1, g2 = rnd(n, seed=1), rnd(n, seed=2) * 0.1, rnd(n, seed=3) * 0.1
    params = {"w": p0.clone()}
    opt = train_step.AdamW(params, lr=1e-3, weight_decay=1e-2)
    p, m, v = dev(p0.clone()), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    p16 = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    for step, g in enumerate((g1, g2), start=1):
        opt.step({"w": g})
        ops.adamw_step(p, dev(g), m, v, p16, step, 1e-3, weight_decay=1e-2)
        assert_close_f32(p, params["w"], f"adamw.p step{step}", 1e-6)
        assert_close_f32(m, opt.m["w"], "adamw.m", 1e-6)
        assert_close_f32(v, opt.v["w"], "adamw.v", 1e-6)
        assert torch.equal(p16.cpu(), bf(p.cpu()))


def test_adamw_reads_bf16_gradients(ops):
    """Data-parallel runs with bf16 gradient messages hand the reduced bf16 buffer straight to the optimizer: the update
    must equal the fp32-gradient update on the same (bf16-representable) values bit for bit, capped grid included."""
    n = 8192 + 16
    p0 = rnd(n, seed=4)
    g = bf(rnd(n, seed=5) * 0.1)                       # fp32 storage, bf16-representable values
    outs = []
    for grad, blocks in ((dev(g), 0), (dev(g).bfloat16(), 0), (dev(g).bfloat16(), 3)):
        p, m, v = dev(p0.clone()), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        p16 = torch.empty(n, dtype=torch.bfloat16, device="cuda")
        ops.adamw_step(p, grad, m, v, p16, 1, 1e-3, weight_decay=1e-2, grad_scale=0.5, max_blocks=blocks)
        outs.append((p, m, v, p16))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


@pytest.mark.parametrize("B,drop", [(4, 0.0), (1, 0.0), (8, 0.1), (37, 0.0)])
def test_head_step_equals_head_forward_loss_and_head_backward_bitwise(ops, B, drop):
    """nv_head_step (the train step's head forward + nn.CrossEntropyLoss + head backward as two launches instead of five) against nv_head_fwd, nv_ce_loss and
    nv_head_bwd: every output - logits, normalised row, statistics, loss, dlogits, the residual gradient (fp32 and bf16, zeros outside the
    cls rows included), dgamma, dbeta, dW, dbias, the column sum - bit for bit; a label outside [0, C) poisons the loss with NaN in both."""
    n, d, C = 65, 768, 2
    x = dev(rnd(B, n, d, seed=1))
    gamma, beta = dev(1.0 + 0.1 * rnd(d, seed=2)), dev(0.1 * rnd(d, seed=3))
    Wh, bh = dev(rnd(C, d, seed=4) * d ** -0.5), dev(0.1 * rnd(C, seed=5))
    y = torch.tensor([i % C for i in range(B)], device="cuda")
    logits, xh, st = ops.head_fwd(x, gamma, beta, Wh, bh)
    loss, dl = ops.ce_loss(logits, y)
    sep = (logits, xh, st, loss, dl) + tuple(ops.head_bwd(dl, Wh, x, st, xh, gamma, drop_seed=77, drop_p=drop))
    one = ops.head_step(x, gamma, beta, Wh, bh, y, drop_seed=77, drop_p=drop)
    names = ("logits", "xh", "stats", "loss", "dlogits", "g", "g16", "dgamma", "dbeta", "dW", "dbias", "dcolsum")
    for name, a, b in zip(names, sep, one):
        assert torch.equal(a, b), f"{name} differ between the fused head step and the three launches"
    assert float(one[5].abs().sum()) > 0 and not one[5][:, 1:].any() and not one[6][:, 1:].any()
    bad = y.clone(); bad[0] = C
    assert torch.isnan(ops.head_step(x, gamma, beta, Wh, bh, bad)[3]).all() and torch.isnan(ops.ce_loss(logits, bad)[0]).all()
    report(f"head step, two launches == head forward + CE + head backward (B={B}, dropout {drop}): 12 outputs bitwise")


@pytest.mark.parametrize("keep", [False, True])
def test_weight_gradient_gemm_with_the_adamw_update_in_its_epilogue(ops, keep):
    """nv_gemm_bf16_grouped_adamw: the four weight gradients of a ViT3D-base layer (K = 2052 rows) whose epilogue applies AdamW to the
    weights at the same arena offsets, against nv_gemm_bf16_grouped + nv_adamw_step on the same arenas: parameters, both moments
    and the bf16 shadow bit for bit, over two steps (bias corrections differ); the gradient is only stored when asked; what lies
    between the weights in the arena is not touched.  nv_adamw_ranges finishes the rest: whole arena == one nv_adamw_step."""
    K = 2052
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    gap = 776                                                  # a bias + padding between two weights (multiple of 8)
    offs, cur = [], 8
    for Mo, N in shapes:
        offs.append(cur)
        cur += Mo * N + gap
    total = cur
    p0 = rnd(total, seed=1) * 0.05
    A = [dev(bf(rnd(K, Mo, seed=20 + i))) for i, (Mo, N) in enumerate(shapes)]
    Bm = [dev(bf(rnd(K, N, seed=30 + i, scale=K ** -0.5))) for i, (Mo, N) in enumerate(shapes)]
    small = dev(rnd(total, seed=9) * 0.01)                     # gradients of everything that is not one of the four weights

    def arenas():
        return dev(p0.clone()), small.clone(), torch.zeros(total, device="cuda"), torch.zeros(total, device="cuda"), dev(p0.clone()).bfloat16()

    def views(g):
        return [g[o:o + Mo * N].view(Mo, N) for o, (Mo, N) in zip(offs, shapes)]

    hyper = dict(lr=1e-3, weight_decay=1e-2, grad_scale=0.5)
    ref, fus = arenas(), arenas()
    sentinel = 123.0
    for step in (1, 2):
        # reference: store the gradients, then one AdamW over the arena
        ops.gemm_tn_grouped([(a, b, c, False) for a, b, c in zip(A, Bm, views(ref[1]))])
        ops.adamw_step(ref[0], ref[1], ref[2], ref[3], ref[4], step, **hyper)
        # fused: the GEMM updates the weights, nv_adamw_ranges the rest
        opt = ops.adamw_arena(*fus, step, keep_grads=keep, **hyper)
        for c in views(fus[1]):
            c.fill_(sentinel)
        ops.gemm_tn_grouped_adamw(list(zip(A, Bm, views(fus[1]))), opt)
        rest, cur = [], 0
        for o, (Mo, N) in zip(offs, shapes):
            rest.append((cur, o - cur))
            cur = o + Mo * N
        rest.append((cur, total - cur))
        if step == 1:                                          # before the ranges ran: nothing outside the four weights has moved
            for b, n in rest:
                assert torch.equal(fus[0][b:b + n].cpu(), p0[b:b + n]) and not fus[2][b:b + n].any()
        from neurovit_amd._cabi import lib
        lib.nv_gemm_set_tile(13, 5 if step == 2 else 0)        # second step: five workgroups walk every chunk of the launch
        try:
            ops.adamw_ranges(opt, rest + [(0, 0)])             # (an empty range is skipped)
        finally:
            lib.nv_gemm_set_tile(13, 0)
        for name, x, y in zip(("parameters", "exp_avg", "exp_avg_sq", "bf16 shadow"), (ref[0], ref[2], ref[3], ref[4]), (fus[0], fus[2], fus[3], fus[4])):
            assert torch.equal(x, y), f"step {step}: {name} differ between the fused epilogue and GEMM + AdamW"
        for c_ref, c_fus in zip(views(ref[1]), views(fus[1])):
            if keep:
                assert torch.equal(c_ref, c_fus)
            else:
                assert bool((c_fus == sentinel).all()), "keep_grads = 0 must not write the gradient"
    assert float((ref[0].cpu() - p0).abs().max()) > 1e-4       # the update did something
    report(f"weight-gradient GEMM + AdamW epilogue (keep_grads={int(keep)}) == GEMM then AdamW: p, m, v, bf16 shadow bitwise over 2 steps")


# ------------------------------------------------------------------------------------------ input contract (A0 / F3)
@pytest.mark.parametrize("dtype,four_d", [("float32", False), ("int16", False), ("float32", True), ("int16", True)])
def test_zscore_crop_matches_dataset_preprocessing(dtype, four_d):
    """DatasetADNI.py:212-213 / DatasetADNI_4D.py:86-87: raw 91 x 109 x 91 grid -> [1:, 10:-9, 1:] -> 90^3, z-score with
    the population std over the whole cropped sample (all timepoints for 4D).  Parity with the numpy statement: 2e-6."""
    from neurovit_amd.preprocess import zscore_crop
    g = np.random.default_rng(5)
    shape = (2, 91, 109, 91, 3) if four_d else (2, 91, 109, 91)
    raw = (g.normal(800.0, 300.0, size=shape)).astype(np.float32)
    if dtype == "int16":
        raw = np.clip(raw, 0, 32000).astype(np.int16)
    ref = ref_cpu.zscore_crop(raw)
    out, stats = zscore_crop(torch.from_numpy(raw).cuda(), return_stats=True)
    assert out.shape == ref.shape and out.dtype == torch.float32 and out.is_contiguous()
    err = float((out.cpu() - torch.from_numpy(ref)).abs().max())
    assert err < 2e-6, err
    flat = ref.reshape(ref.shape[0], -1)
    assert np.allclose(flat.mean(1), 0, atol=1e-5) and np.allclose(flat.std(1), 1, atol=1e-5)
    # strided input (a permuted view) goes through the same kernel
    if not four_d:
        view = torch.from_numpy(raw).cuda().permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1)
        assert torch.equal(zscore_crop(view), out)
    # constant volume: std = 0 -> (x - mean) / 1e-8 = 0, no NaN (DatasetADNI.py:213 adds 1e-8 for exactly this)
    const = torch.full((1, 91, 109, 91), 7.0, device="cuda")
    assert torch.count_nonzero(zscore_crop(const)) == 0


# ------------------------------------------------------------------------------------------ LayerNorm folded into the GEMMs around it
def _fold_reference(x, gamma, beta, W, b, gelu):
    y = F.layer_norm(x.double(), (x.shape[1],), gamma.double(), beta.double(), 1e-5) @ W.double().T
    if b is not None:
        y = y + b.double()
    return F.gelu(y) if gelu else y


@pytest.mark.parametrize("M,d,inner,N", [(2052, 768, 768, 2304), (2052, 768, 3072, 3072), (2052, 192, 192, 384), (2052, 1024, 512, 2048)])
def test_layernorm_folded_into_the_gemms_around_it(ops, M, d, inner, N):
    """the producer of the residual stream (nv_gemm_resid_ln) also writes the rows in bf16 and per-(row, 128-column tile) statistics; the consumer
    (nv_gemm_lnfold) contracts the UN-normalised rows with W diag(gamma) and applies mu / rstd in its epilogue.  Against float64: the producer's three outputs,
    the merged statistics, and the consumer's output at the tolerance of the unfolded pair (LayerNorm -> bf16 -> GEMM); d = 192 has a ragged last tile."""
    from neurovit_amd._cabi import lib
    if not (lib.nv_gemm_lnfold_supported(M, N, d) and lib.nv_gemm_lnfold_supported(M, d, inner)):
        pytest.skip("shape outside the LDS-epilogue kernels")
    A, Wp, bp = bf(rnd(M, inner, seed=1)), bf(rnd(d, inner, seed=2, scale=inner ** -0.5)), rnd(d, seed=3)
    resid = rnd(M, d, seed=4) * 2 + 0.5 * rnd(M, 1, seed=5)              # a residual stream whose rows have their own offsets
    out, out16, stats = ops.gemm_resid_ln(dev(A), dev(Wp), dev(bp), dev(resid))
    x = resid.double() + A.double() @ Wp.double().T + bp.double()
    assert_close_f32(out, x, "resid_ln.out", 1e-5)
    assert torch.equal(out16.cpu(), out.cpu().to(torch.bfloat16))
    # the tile partials (sum, sum of squares), merged as the consumer's prologue does it (per tile -> (mean, M2), then Chan's update), against float64
    tiles = (d + 127) // 128
    st = stats.cpu().double().reshape(tiles, M, 2)
    mean, m2, n = torch.zeros(M, dtype=torch.float64), torch.zeros(M, dtype=torch.float64), 0.0
    for t in range(tiles):
        nt = float(min(128, d - 128 * t))
        mt = st[t, :, 0] / nt
        delta = mt - mean
        mean = mean + delta * nt / (n + nt)
        m2 = m2 + (st[t, :, 1] - st[t, :, 0] * mt).clamp_min(0) + delta * delta * n * nt / (n + nt)
        n += nt
    assert_close_f32(mean, x.mean(1), "merged mean", 2e-5)
    assert_close_f32(m2 / d, x.var(1, unbiased=False), "merged variance", 1e-4)
    gamma, beta = 1 + 0.1 * rnd(d, seed=6), 0.1 * rnd(d, seed=7)
    Wc, bc = rnd(N, d, seed=8, scale=d ** -0.5), 0.1 * rnd(N, seed=9)
    for gelu, bias in ((False, None), (True, bc)):
        Wg, cs, fb = ops.ln_fold_weight(dev(Wc), dev(gamma), dev(beta), None if bias is None else dev(bias))
        assert torch.equal(Wg.cpu(), (Wc * gamma).to(torch.bfloat16))
        assert_close_f32(cs, (Wc * gamma).to(torch.bfloat16).double().sum(1), "colsum", 1e-5)
        assert_close_f32(fb, Wc.double() @ beta.double() + (0 if bias is None else bias.double()), "folded bias", 1e-5)
        y = ops.gemm_lnfold(out16, Wg, stats, cs, fb, gelu=gelu)
        ref = _fold_reference(out.cpu(), gamma, beta, Wc, bias, gelu)
        assert rel_l2(y.float(), ref) <= 4e-3 and rel_err(y.float(), ref) <= 2.0 ** -6      # three bf16 roundings (rows, weight, output) against exact float64 inputs
        # ... and it is as close to float64 as the launches it replaces
        xn, _ = ops.ln_fwd(out, dev(gamma), dev(beta))
        plain = ops.gemm(ops.NT, ops.EPI_BIAS_GELU if gelu else ops.EPI_STORE_BF16, xn, dev(bf(Wc)), bias=None if bias is None else dev(bias)) if (gelu or bias is None) else None
        if plain is not None:
            e_fold, e_plain = rel_l2(y.float(), ref), rel_l2(plain.float(), ref)
            assert e_fold <= 1.5 * e_plain + 1e-4, (e_fold, e_plain)


def test_layernorm_fold_error_grows_with_the_row_offset_and_stays_inside_its_bound(ops):
    """The fold is exact algebra; in 16-bit operands its rounding error relative to the unfolded pair grows with |mean| / std of a residual row (x is rounded BEFORE
    the mean is taken out: SURVEY 7.3 item 3's cancellation).  Rows of the transformer's residual stream sit at |mean| / std < 1; the bound
    2^-8 sqrt(1 + (mean / std)^2) is asserted up to 16, and a constant row (std = 0: rstd = 1 / sqrt(eps)) stays finite and matches the reference's beta-only output."""
    M, d, N = 2052, 768, 768
    gamma, beta = 1 + 0.1 * rnd(d, seed=6), 0.1 * rnd(d, seed=7)
    Wc = rnd(N, d, seed=8, scale=d ** -0.5)
    Wg, cs, fb = ops.ln_fold_weight(dev(Wc), dev(gamma), dev(beta))
    zero_a, eye = torch.zeros(M, 64, dtype=torch.bfloat16), torch.zeros(d, 64, dtype=torch.bfloat16)
    for ratio in (0.0, 1.0, 4.0, 16.0):
        x = rnd(M, d, seed=11) + ratio
        out, out16, stats = ops.gemm_resid_ln(dev(zero_a), dev(eye), dev(torch.zeros(d)), dev(x))      # x = resid + 0
        y = ops.gemm_lnfold(out16, Wg, stats, cs, fb)
        ref = _fold_reference(x, gamma, beta, Wc, None, False)
        e = rel_l2(y.float(), ref)
        report(f"LayerNorm fold, row offset / std = {ratio:g}: rel-L2 {e:.2e} (bound {2.0 ** -8 * (1 + ratio * ratio) ** 0.5:.2e})")
        assert e <= 2.0 ** -8 * (1 + ratio * ratio) ** 0.5
    const = torch.full((M, d), 3.0)
    out, out16, stats = ops.gemm_resid_ln(dev(zero_a), dev(eye), dev(torch.zeros(d)), dev(const))
    y = ops.gemm_lnfold(out16, Wg, stats, cs, fb).float().cpu()
    ref = _fold_reference(const, gamma, beta, Wc, None, False)
    assert torch.isfinite(y).all() and rel_l2(y, ref) < 2e-2
