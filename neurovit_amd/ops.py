"""Thin functional wrappers over the C-ABI (one Python function per kernel entry point).

PyTorch is plumbing only: it owns the device buffers and the stream; every FLOP below runs in
libneurovit_hip.so.  No fallbacks: CPU tensors raise.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _cabi
from ._cabi import check, lib


def op16() -> torch.dtype:
    """torch dtype of the 16-bit operand buffers under the current operand format (_cabi.set_operand_format): bfloat16 or float16"""
    return torch.float16 if _cabi.operand_format() == "fp16" else torch.bfloat16


NT, NN, TN = 0, 1, 2
EPI_STORE_BF16, EPI_STORE_F32, EPI_BIAS_F32, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_DGELU, EPI_DGELU_COLSUM = 0, 1, 2, 3, 4, 5, 6


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("neurovit_amd ops run on MI355X only (got a CPU tensor); there is no CPU fallback")


def shape5(video: torch.Tensor):
    assert video.dim() == 5
    return (ctypes.c_long * 5)(*video.shape)


def cast_ranges_bf16(src: torch.Tensor, dst: torch.Tensor, ranges) -> None:
    """dst[b:e] = bf16(src[b:e]) for the element ranges [(b, e), ...] of two flat arenas (fp32 -> bf16), one launch per 48 ranges."""
    _need_cuda(src, dst)
    assert src.dtype == torch.float32 and dst.dtype == op16() and src.dim() == 1 and dst.numel() >= src.numel()
    ranges = [(int(b), int(e)) for b, e in ranges if e > b]
    if not ranges:
        return
    begins = (ctypes.c_long * len(ranges))(*[b for b, _ in ranges])
    lens = (ctypes.c_long * len(ranges))(*[e - b for b, e in ranges])
    check(lib.nv_cast_ranges_bf16(_p(src), _p(dst), begins, lens, len(ranges), _stream()), "nv_cast_ranges_bf16")


def skinny_nt(epi: int, A: torch.Tensor, W: torch.Tensor, bias: torch.Tensor, out: torch.Tensor, resid: Optional[torch.Tensor] = None,
              u_out: Optional[torch.Tensor] = None, drop_seed: int = 0, drop_p: float = 0.0) -> torch.Tensor:
    """out[r] = epilogue(A[r] @ W.T) on a few rows; A / resid / out / u_out may be row-strided 2-D views (e.g. x[::n]).
    epi 0: out f32 = resid + (bias + A W^T) * mask; epi 1: u = bias + A W^T (bf16, optional), out bf16 = gelu(u) * mask.
    mask: nn.Dropout of the dense tensor `out` is a view of (element offsets are hashed), drop_p = 0: none."""
    _need_cuda(A, W)
    R, K = A.shape
    N = W.shape[0]
    check(lib.nv_skinny_nt(epi, R, N, K, _p(A), A.stride(0), _p(W), W.stride(0), _p(bias), _p(resid), 0 if resid is None else resid.stride(0),
                           _p(out), out.stride(0), _p(u_out), 0 if u_out is None else u_out.stride(0), int(drop_seed), float(drop_p), _stream()), "nv_skinny_nt")
    return out


def skinny_nn(epi: int, A: torch.Tensor, W: torch.Tensor, out: torch.Tensor, u: Optional[torch.Tensor] = None,
              dcol: Optional[torch.Tensor] = None, accumulate: bool = False, drop_seed: int = 0, drop_p: float = 0.0) -> torch.Tensor:
    """out[r] = epilogue(A[r] @ W) on a few rows (W [K, N]); epi 0: bf16 (A W * mask) * gelu'(u) (+ column sums into dcol), 1: f32, 2: bf16."""
    _need_cuda(A, W)
    R, K = A.shape
    N = W.shape[1]
    check(lib.nv_skinny_nn(epi, R, N, K, _p(A), A.stride(0), _p(W), W.stride(0), _p(u), 0 if u is None else u.stride(0), _p(out), out.stride(0),
                           _p(dcol), int(accumulate), int(drop_seed), float(drop_p), _stream()), "nv_skinny_nn")
    return out


def strides5(video: torch.Tensor):
    assert video.dim() == 5
    return (ctypes.c_long * 5)(*video.stride())


def gemm(layout: int, epi: int, A: torch.Tensor, B: torch.Tensor, *, out: Optional[torch.Tensor] = None, bias=None,
         aux_in=None, aux_out=None, accumulate: bool = False, alpha: float = 1.0, drop_seed: int = 0, drop_p: float = 0.0) -> torch.Tensor:
    """C = op(A) op(B) with a fused epilogue; A, B bf16 2-D row-major (last stride 1)."""
    _need_cuda(A, B)
    assert A.dtype == op16() and B.dtype == op16() and A.stride(1) == 1 and B.stride(1) == 1
    if layout == NT:
        M, K = A.shape; N = B.shape[0]; assert B.shape[1] == K
    elif layout == NN:
        M, K = A.shape; N = B.shape[1]; assert B.shape[0] == K
    else:
        K, M = A.shape; N = B.shape[1]; assert B.shape[0] == K
    odt = op16() if epi in (EPI_STORE_BF16, EPI_BIAS_GELU, EPI_DGELU, EPI_DGELU_COLSUM) else torch.float32
    if out is None:
        out = torch.empty((M, N), dtype=odt, device=A.device)
    assert out.dtype == odt and out.shape == (M, N) and out.stride(1) == 1
    check(lib.nv_gemm_bf16(layout, epi, M, N, K, _p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0), _p(bias),
                           _p(aux_in), 0 if aux_in is None else aux_in.stride(0), _p(aux_out),
                           0 if aux_out is None else aux_out.stride(0), int(accumulate), float(alpha), drop_seed, drop_p, _stream()), "nv_gemm_bf16")
    return out


EPI_F32_STORE, EPI_F32_BIAS, EPI_F32_BIAS_GELU, EPI_F32_BIAS_RESID = 0, 2, 3, 4


def gemm_f32(epi: int, A: torch.Tensor, W: torch.Tensor, *, bias=None, resid=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 inference path: out[M, N] = epi(A[M, K] @ W[N, K].T), every operand fp32, contraction on the fp32 MFMA.
    A / resid / out may be row-strided 2-D views (last stride 1)."""
    _need_cuda(A, W)
    assert A.dtype == torch.float32 and W.dtype == torch.float32 and A.stride(1) == 1 and W.stride(1) == 1
    M, K = A.shape
    N = W.shape[0]
    assert W.shape[1] == K
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    assert out.dtype == torch.float32 and out.shape == (M, N) and out.stride(1) == 1
    check(lib.nv_gemm_f32(epi, M, N, K, _p(A), A.stride(0), _p(W), W.stride(0), _p(out), out.stride(0), _p(bias), _p(resid),
                          0 if resid is None else resid.stride(0), _stream()), "nv_gemm_f32")
    return out


def attn_fwd_f32(qkv: torch.Tensor, B: int, n: int, heads: int, dim_head: int = 64) -> torch.Tensor:
    """qkv f32 [B*n, 3*inner] -> out f32 [B*n, inner] (softmax(q k^T / sqrt(dh)) v per head, fp32 MFMA)."""
    _need_cuda(qkv)
    assert qkv.dtype == torch.float32 and qkv.stride(1) == 1
    inner = heads * dim_head
    out = torch.empty((B * n, inner), dtype=torch.float32, device=qkv.device)
    check(lib.nv_attn_fwd_f32(_p(qkv), qkv.stride(0), B, n, heads, dim_head, dim_head ** -0.5, _p(out), inner, _stream()), "nv_attn_fwd_f32")
    return out


def ln_fwd_f32(x: torch.Tensor, gamma, beta, eps: float = 1e-5) -> torch.Tensor:
    _need_cuda(x)
    M, d = x.shape
    y = torch.empty((M, d), dtype=torch.float32, device=x.device)
    check(lib.nv_ln_fwd_f32(_p(x), x.stride(0), M, d, _p(gamma), _p(beta), eps, _p(y), d, None, None, _stream()), "nv_ln_fwd_f32")
    return y


def patch_ln_fwd_f32(video: torch.Tensor, p1: int, p2: int, pf: int, gamma, beta, eps: float = 1e-5, vol_sigma=None):
    """As patch_ln_fwd with fp32 tokens [B*N, P] (fp32 inference path)."""
    _need_cuda(video)
    B, C, F, H, W = video.shape
    P = C * p1 * p2 * pf
    N = (F // pf) * (H // p1) * (W // p2)
    out = torch.empty((B * N, P), dtype=torch.float32, device=video.device)
    st = torch.empty((2, B * N), dtype=torch.float32, device=video.device)
    check(lib.nv_patch_ln_fwd_f32(_p(video), strides5(video), B, C, F, H, W, p1, p2, pf, _p(gamma), _p(beta), eps, _p(out), P, _p(st[0]),
                                  _p(st[1]), _p(vol_sigma), _stream()), "nv_patch_ln_fwd_f32")
    return out, st


class GemmProblem(ctypes.Structure):          # include/neurovit_hip.h::nv_gemm_problem
    _fields_ = [("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int), ("A", ctypes.c_void_p), ("lda", ctypes.c_long),
                ("B", ctypes.c_void_p), ("ldb", ctypes.c_long), ("C", ctypes.c_void_p), ("ldc", ctypes.c_long), ("accumulate", ctypes.c_int),
                ("C16", ctypes.c_void_p), ("ldc16", ctypes.c_long)]


def gemm_tn_grouped(problems) -> None:
    """problems: up to four (A[K, M] bf16, B[K, N] bf16, C[M, N] f32, accumulate[, C16[M, N] bf16 mirror]) - C (+)= A^T B, one launch."""
    arr = (GemmProblem * len(problems))()
    for i, pr in enumerate(problems):
        A, B, C, acc = pr[:4]
        C16 = pr[4] if len(pr) > 4 else None
        _need_cuda(A)
        K, M = A.shape
        N = B.shape[1]
        assert B.shape[0] == K and C.shape == (M, N) and C.dtype == torch.float32
        assert C16 is None or (C16.shape == (M, N) and C16.dtype == op16())
        arr[i] = GemmProblem(M, N, K, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), C.data_ptr(), C.stride(0), int(acc),
                             _p(C16), 0 if C16 is None else C16.stride(0))
    check(lib.nv_gemm_bf16_grouped(TN, EPI_STORE_F32, len(problems), ctypes.cast(arr, ctypes.c_void_p), _stream()), "nv_gemm_bf16_grouped")


def adamw_arena(params, grads, adam_m, adam_v, params16, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0, keep_grads=False):
    """struct nv_adamw_arena over five flat tensors that share element offsets (fp32 x 4, bf16 shadow)."""
    from ._cabi import AdamwArena
    n = params.numel()
    assert all(t.is_cuda and t.is_contiguous() and t.numel() == n for t in (params, grads, adam_m, adam_v, params16))
    assert params16.dtype == op16() and all(t.dtype == torch.float32 for t in (params, grads, adam_m, adam_v))
    return AdamwArena(ctypes.sizeof(AdamwArena), int(step), float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), float(grad_scale),
                      int(bool(keep_grads)), params.data_ptr(), grads.data_ptr(), adam_m.data_ptr(), adam_v.data_ptr(), params16.data_ptr())


def gemm_tn_grouped_adamw(problems, opt) -> None:
    """problems: up to four (A[K, M] bf16, B[K, N] bf16, C[M, N] f32 VIEW INTO opt's gradient arena): the gradient C = A^T B is consumed by
    the AdamW update of the parameters at the same arena offsets inside the GEMM's epilogue (nv_gemm_bf16_grouped_adamw)."""
    arr = (GemmProblem * len(problems))()
    for i, pr in enumerate(problems):
        A, B, C = pr[:3]
        _need_cuda(A)
        K, M = A.shape
        N = B.shape[1]
        assert B.shape[0] == K and C.shape == (M, N) and C.dtype == torch.float32
        arr[i] = GemmProblem(M, N, K, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), C.data_ptr(), C.stride(0), 0, None, 0)
    check(lib.nv_gemm_bf16_grouped_adamw(len(problems), ctypes.cast(arr, ctypes.c_void_p), ctypes.byref(opt), _stream()), "nv_gemm_bf16_grouped_adamw")


def adamw_ranges(opt, ranges) -> None:
    """AdamW (nv_adamw_step arithmetic) over [(begin, len)] element ranges of opt's arenas, one launch."""
    n = len(ranges)
    b = (ctypes.c_long * max(n, 1))(*[int(r[0]) for r in ranges])
    l = (ctypes.c_long * max(n, 1))(*[int(r[1]) for r in ranges])
    check(lib.nv_adamw_ranges(ctypes.byref(opt), b, l, n, _stream()), "nv_adamw_ranges")


def ln_fwd(x: torch.Tensor, gamma, beta, eps: float = 1e-5):
    _need_cuda(x)
    M, d = x.shape
    y = torch.empty((M, d), dtype=op16(), device=x.device)
    st = torch.empty((2, M), dtype=torch.float32, device=x.device)
    check(lib.nv_ln_fwd(_p(x), x.stride(0), M, d, _p(gamma), _p(beta), eps, _p(y), d, _p(st[0]), _p(st[1]), _stream()), "nv_ln_fwd")
    return y, st


def ln_bwd(dy, x, st, gamma, g_in=None, want_g16=True, accumulate=False, dgamma=None, dbeta=None, dcolsum=None, drop_seed=0, drop_p=0.0):
    _need_cuda(dy, x)
    M, d = x.shape
    g_out = torch.empty((M, d), dtype=torch.float32, device=x.device) if g_in is None else g_in
    g16 = torch.empty((M, d), dtype=op16(), device=x.device) if want_g16 else None
    dgamma = torch.empty(d, device=x.device) if dgamma is None else dgamma
    dbeta = torch.empty(d, device=x.device) if dbeta is None else dbeta
    dcolsum = torch.empty(d, device=x.device) if dcolsum is None else dcolsum
    nb = lib.nv_ln_bwd_workspace_bytes(M, d)
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    check(lib.nv_ln_bwd(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(st[0]), _p(st[1]), _p(gamma), M, d, _p(g_in), _p(g_out), d, _p(g16), d,
                        _p(dgamma), _p(dbeta), _p(dcolsum), int(accumulate), _p(ws), nb, drop_seed, drop_p, _stream(), None), "nv_ln_bwd")
    return g_out, g16, dgamma, dbeta, dcolsum


def patch_ln_fwd(video: torch.Tensor, p1: int, p2: int, pf: int, gamma, beta, eps: float = 1e-5, ldo: Optional[int] = None, vol_sigma=None):
    """video [B,C,F,H,W] (any strides, e.g. the permuted view of a [B,H,W,D] volume)."""
    _need_cuda(video)
    B, C, F, H, W = video.shape
    P = C * p1 * p2 * pf
    N = (F // pf) * (H // p1) * (W // p2)
    ldo = (P + 7) // 8 * 8 if ldo is None else ldo
    out = torch.empty((B * N, ldo), dtype=op16(), device=video.device)
    st = torch.empty((2, B * N), dtype=torch.float32, device=video.device)
    check(lib.nv_patch_ln_fwd(_p(video), strides5(video), B, C, F, H, W, p1, p2, pf, _p(gamma), _p(beta), eps, _p(out), ldo, _p(st[0]),
                              _p(st[1]), _p(vol_sigma), _stream()), "nv_patch_ln_fwd")
    return out, st


def patch_ln_bwd(video, p1, p2, pf, dxp, st, accumulate=False):
    B, C, F, H, W = video.shape
    P = C * p1 * p2 * pf
    T = dxp.shape[0]
    dg = torch.empty(P, device=video.device); db = torch.empty(P, device=video.device)
    nb = lib.nv_patch_ln_bwd_workspace_bytes(T, P)
    ws = torch.empty(nb, dtype=torch.uint8, device=video.device)
    check(lib.nv_patch_ln_bwd(_p(video), strides5(video), B, C, F, H, W, p1, p2, pf, _p(dxp), dxp.stride(0), _p(st[0]), _p(st[1]), _p(dg),
                              _p(db), int(accumulate), _p(ws), nb, _stream()), "nv_patch_ln_bwd")
    return dg, db


def embed_finish_fwd(t, B, N, gamma, beta, pos, cls, eps=1e-5, drop_seed=0, drop_p=0.0):
    d = t.shape[1]
    x = torch.empty((B, N + 1, d), dtype=torch.float32, device=t.device)
    st = torch.empty((2, B * N), dtype=torch.float32, device=t.device)
    check(lib.nv_embed_finish_fwd(_p(t), t.stride(0), B, N, d, _p(gamma), _p(beta), eps, _p(pos), _p(cls), _p(x), d, _p(st[0]), _p(st[1]),
                                  drop_seed, drop_p, _stream()), "nv_embed_finish_fwd")
    return x, st


def embed_finish_bwd(g, t, st, gamma, B, N, drop_seed=0, drop_p=0.0):
    d = t.shape[1]
    dev = t.device
    dt = torch.empty((B * N, d), device=dev); dt16 = torch.empty((B * N, d), dtype=op16(), device=dev)
    dgamma, dbeta, dbias = (torch.empty(d, device=dev) for _ in range(3))
    dpos = torch.empty((N + 1, d), device=dev); dcls = torch.empty(d, device=dev)
    nb = lib.nv_embed_finish_bwd_workspace_bytes(B, N, d)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    g2 = g.reshape(B * (N + 1), d)
    check(lib.nv_embed_finish_bwd(_p(g2), d, _p(t), t.stride(0), _p(st[0]), _p(st[1]), _p(gamma), B, N, d, _p(dt), d, _p(dt16), d, _p(dgamma),
                                  _p(dbeta), _p(dbias), _p(dpos), _p(dcls), 0, _p(ws), nb, drop_seed, drop_p, _stream()), "nv_embed_finish_bwd")
    return dt, dt16, dgamma, dbeta, dbias, dpos, dcls


def attn_fwd(qkv: torch.Tensor, B: int, n: int, heads: int, dim_head: int = 64, drop_seed=0, drop_p=0.0):
    """qkv bf16 [B*n, 3*inner] -> (out bf16 [B*n, inner], lse f32 [B, heads, n])."""
    _need_cuda(qkv)
    inner = heads * dim_head
    out = torch.empty((B * n, inner), dtype=op16(), device=qkv.device)
    lse = torch.empty((B, heads, n), dtype=torch.float32, device=qkv.device)
    check(lib.nv_attn_fwd(_p(qkv), qkv.stride(0), B, n, heads, dim_head, dim_head ** -0.5, _p(out), inner, _p(lse), drop_seed, drop_p, _stream()), "nv_attn_fwd")
    return out, lse


def attn_fwd_o8(qkv: torch.Tensor, B: int, n: int, heads: int, out_scale: float, dim_head: int = 64) -> torch.Tensor:
    """Attention forward with the output as OCP e4m3 bytes of (value * out_scale): uint8 [B*n, heads*dim_head] (fp8 inference path)."""
    _need_cuda(qkv)
    inner = heads * dim_head
    out = torch.empty((B * n, inner), dtype=torch.uint8, device=qkv.device)
    check(lib.nv_attn_fwd_o8(_p(qkv), qkv.stride(0), B, n, heads, dim_head, dim_head ** -0.5, _p(out), inner, float(out_scale), _stream()), "nv_attn_fwd_o8")
    return out


def attn_bwd(qkv, out, dout, lse, B, n, heads, dim_head=64, drop_seed=0, drop_p=0.0):
    inner = heads * dim_head
    dqkv = torch.empty((B * n, 3 * inner), dtype=op16(), device=qkv.device)
    delta = torch.empty((B, heads, n), dtype=torch.float32, device=qkv.device)
    check(lib.nv_attn_bwd(_p(qkv), qkv.stride(0), _p(out), _p(dout), inner, _p(lse), B, n, heads, dim_head, dim_head ** -0.5, _p(delta),
                          _p(dqkv), 3 * inner, drop_seed, drop_p, _stream()), "nv_attn_bwd")
    return dqkv, delta


def head_fwd(x: torch.Tensor, gamma, beta, W, bias, eps=1e-5):
    """x f32 [B, n, d] -> logits [B, C] from the cls row (pool='cls')."""
    B, n, d = x.shape
    C = W.shape[0]
    xh = torch.empty((B, d), device=x.device); st = torch.empty((B, 2), device=x.device)
    logits = torch.empty((B, C), device=x.device)
    check(lib.nv_head_fwd(_p(x), n * d, B, d, _p(gamma), _p(beta), eps, _p(W), _p(bias), C, _p(xh), _p(st), _p(logits), _stream()), "nv_head_fwd")
    return logits, xh, st


def head_bwd(dlogits, W, x, st, xh, gamma, drop_seed=0, drop_p=0.0):
    B, n, d = x.shape
    C = W.shape[0]
    dev = x.device
    g = torch.empty((B, n, d), device=dev); g16 = torch.empty((B, n, d), dtype=op16(), device=dev)
    dgamma, dbeta, dcol = (torch.empty(d, device=dev) for _ in range(3))
    dW = torch.empty((C, d), device=dev); db = torch.empty(C, device=dev)
    nb = lib.nv_head_bwd_workspace_bytes(B, d)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    check(lib.nv_head_bwd(_p(dlogits), B, C, _p(W), _p(x), n * d, _p(st), _p(xh), _p(gamma), d, n, _p(g), d, _p(g16), d, _p(dgamma), _p(dbeta),
                          _p(dW), _p(db), _p(dcol), 0, _p(ws), nb, drop_seed, drop_p, 0, _stream()), "nv_head_bwd")
    return g, g16, dgamma, dbeta, dW, db, dcol


def head_step(x, gamma, beta, W, bias, labels, eps=1e-5, drop_seed=0, drop_p=0.0):
    """nv_head_step: head forward + CrossEntropyLoss + head backward in two launches.  Returns what head_fwd, ce_loss and head_bwd return together:
    (logits, xh, st, loss, dlogits, g, g16, dgamma, dbeta, dW, db, dcol)."""
    B, n, d = x.shape
    C = W.shape[0]
    dev = x.device
    xh = torch.empty((B, d), device=dev); st = torch.empty((B, 2), device=dev)
    logits = torch.empty((B, C), device=dev); dl = torch.empty((B, C), device=dev); loss = torch.empty(1, device=dev)
    g = torch.full((B, n, d), float("nan"), device=dev); g16 = torch.full((B, n, d), float("nan"), dtype=op16(), device=dev)
    dgamma, dbeta, dcol = (torch.empty(d, device=dev) for _ in range(3))
    dW = torch.empty((C, d), device=dev); db = torch.empty(C, device=dev)
    nb = lib.nv_head_step_workspace_bytes(B, d)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    check(lib.nv_head_step(_p(x), n * d, B, d, _p(gamma), _p(beta), eps, _p(W), _p(bias), C, _p(labels), 1.0, _p(xh), _p(st), _p(logits), _p(loss), _p(dl),
                           n, _p(g), d, _p(g16), d, _p(dgamma), _p(dbeta), _p(dW), _p(db), _p(dcol), 0, _p(ws), nb, drop_seed, drop_p, _stream()), "nv_head_step")
    return logits, xh, st, loss, dl, g, g16, dgamma, dbeta, dW, db, dcol


def colsum_bf16(X: torch.Tensor, accumulate=False, out=None):
    M, N = X.shape
    out = torch.empty(N, device=X.device) if out is None else out
    nb = lib.nv_colsum_workspace_bytes(M, N)
    ws = torch.empty(nb, dtype=torch.uint8, device=X.device)
    check(lib.nv_colsum_bf16(_p(X), X.stride(0), M, N, _p(out), int(accumulate), _p(ws), nb, _stream()), "nv_colsum_bf16")
    return out


def ce_loss(logits: torch.Tensor, target: torch.Tensor, grad_scale: float = 1.0, want_grad=True, scale_state=None):
    B, C = logits.shape
    loss = torch.empty(1, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    check(lib.nv_ce_loss_scaled(_p(logits), _p(target), B, C, grad_scale, _p(scale_state), _p(loss), _p(dl), _stream()), "nv_ce_loss")
    return loss, dl


def adamw_step(p, grad, m, v, p16, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0, max_blocks=0, scale_state=None):
    """grad may be fp32 or the 16-bit operand format (same numel as p).  scale_state: device block of a dynamic loss scale
    (optim.LossScaler.state) - the launch then does nothing when that step is skipped, un-scales the gradients and takes the step
    count / bias corrections from the block (`step` is ignored)."""
    assert grad.dtype in (torch.float32, op16()) and grad.numel() == p.numel()
    check(lib.nv_adamw_step_scaled(_p(p), _p(grad), int(grad.dtype == op16()), _p(m), _p(v), _p(p16), p.numel(), max(int(step), 1), lr, betas[0], betas[1],
                                   eps, weight_decay, grad_scale, int(max_blocks), _p(scale_state), _stream()), "nv_adamw_step")


def loss_scale_init(state: torch.Tensor, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, start_step=0) -> None:
    assert state.is_cuda and state.dtype == torch.float32 and state.numel() >= 16 and state.is_contiguous()
    check(lib.nv_loss_scale_init(_p(state), float(init_scale), float(growth_factor), float(backoff_factor), int(growth_interval), int(start_step), _stream()),
          "nv_loss_scale_init")


def loss_scale_check(grads: torch.Tensor, state: torch.Tensor) -> None:
    """state[found_inf] |= any inf / NaN in the fp32 tensor `grads` (contiguous)."""
    assert grads.is_cuda and grads.dtype == torch.float32 and grads.is_contiguous()
    if grads.numel():
        check(lib.nv_loss_scale_check(_p(grads), grads.numel(), _p(state), _stream()), "nv_loss_scale_check")


def loss_scale_update(state: torch.Tensor, lr: float, betas) -> None:
    check(lib.nv_loss_scale_update(_p(state), float(lr), float(betas[0]), float(betas[1]), _stream()), "nv_loss_scale_update")


def cast_bf16(src: torch.Tensor, ld_dst: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 [rows, cols] -> bf16 [rows, ld_dst] (zero padded columns)."""
    _need_cuda(src)
    rows, cols = src.shape
    if out is not None:
        ld_dst = out.stride(0) if rows > 1 else out.shape[1]
    ld_dst = (cols + 3) // 4 * 4 if ld_dst is None else ld_dst
    dst = torch.empty((rows, ld_dst), dtype=op16(), device=src.device) if out is None else out
    check(lib.nv_cast_bf16_2d(_p(src), src.stride(0), rows, cols, _p(dst), ld_dst, _stream()), "nv_cast_bf16_2d")
    return dst


def gradcam_reduce(act: torch.Tensor, grad: torch.Tensor):
    """act bf16 [B, n, d], grad f32 [B, n, d] (device) -> (cam f32 [B, n-1] min-max normalised, minmax f32 [2]); one launch."""
    _need_cuda(act, grad)
    assert act.dtype == op16() and grad.dtype == torch.float32 and act.shape == grad.shape and act.is_contiguous() and grad.is_contiguous()
    B, n, d = act.shape
    cam = torch.empty((B, n - 1), dtype=torch.float32, device=act.device)
    mm = torch.empty(2, dtype=torch.float32, device=act.device)
    nb = lib.nv_gradcam_workspace_bytes(B, n)
    ws = torch.empty(nb, dtype=torch.uint8, device=act.device)
    check(lib.nv_gradcam_reduce(_p(act), _p(grad), B, n, d, _p(cam), _p(mm), _p(ws), nb, _stream()), "nv_gradcam_reduce")
    return cam, mm


def dropout_apply(x: torch.Tensor, drop_seed: int = 0, drop_p: float = 0.0, want16: bool = True, want32: bool = False):
    """x f32 [M, N] times the dropout mask of one site -> (bf16 copy or None, f32 copy or None)."""
    _need_cuda(x)
    M, N = x.shape
    o16 = torch.empty((M, N), dtype=op16(), device=x.device) if want16 else None
    o32 = torch.empty((M, N), dtype=torch.float32, device=x.device) if want32 else None
    check(lib.nv_dropout_apply(_p(x), x.stride(0), M, N, drop_seed, drop_p, _p(o16), N, _p(o32), N, _stream()), "nv_dropout_apply")
    return o16, o32


class ReduceJob(ctypes.Structure):            # include/neurovit_hip.h::nv_reduce_job
    _fields_ = [("partials", ctypes.c_void_p), ("rows", ctypes.c_int), ("width", ctypes.c_int), ("nseg", ctypes.c_int),
                ("out", ctypes.c_void_p * 3), ("accumulate", ctypes.c_int)]


def reduce_multi(jobs) -> None:
    """jobs: [(partials f32 [rows, nseg*width], width, [out tensors or None] (nseg of them), accumulate)] - one launch."""
    arr = (ReduceJob * len(jobs))()
    for i, (part, width, outs, acc) in enumerate(jobs):
        _need_cuda(part)
        rows, tot = part.shape
        nseg = tot // width
        assert nseg * width == tot and len(outs) == nseg and part.is_contiguous()
        o = (ctypes.c_void_p * 3)(*[(_p(t) if t is not None else None) for t in list(outs) + [None] * (3 - nseg)])
        arr[i] = ReduceJob(part.data_ptr(), rows, width, nseg, o, int(acc))
    check(lib.nv_reduce_multi(ctypes.cast(arr, ctypes.c_void_p), len(jobs), _stream()), "nv_reduce_multi")


def gemm_dgelu_colsum(A: torch.Tensor, B: torch.Tensor, u: torch.Tensor, drop_seed: int = 0, drop_p: float = 0.0):
    """dU = (A B) * gelu'(u) (bf16) with the fused column-sum epilogue -> (dU, partial column sums f32 [tile rows, N]);
    raises when the shape has no large-tile kernel (use EPI_DGELU + colsum_bf16 then)."""
    M, K = A.shape
    N = B.shape[1]
    rows = lib.nv_gemm_tile_rows(NN, M, N, K, A.stride(0), B.stride(0))
    if rows == 0:
        raise RuntimeError("neurovit_amd: fused column-sum epilogue not available for this shape")
    part = torch.empty(((M + rows - 1) // rows, N), dtype=torch.float32, device=A.device)
    out = gemm(NN, EPI_DGELU_COLSUM, A, B, aux_in=u, aux_out=part, drop_seed=drop_seed, drop_p=drop_p)
    return out, part


EPI_BIAS_GELU_F8 = 7


def quant_rows_f8(W: torch.Tensor, act_scale: float):
    """fp32 [rows, cols] -> (e4m3 bytes uint8 [rows, cols], colscale f32 [rows] = 1 / (act_scale * row scale))."""
    _need_cuda(W)
    rows, cols = W.shape
    out = torch.empty((rows, cols), dtype=torch.uint8, device=W.device)
    cs = torch.empty(rows, dtype=torch.float32, device=W.device)
    check(lib.nv_quant_rows_f8(_p(W), W.stride(0), rows, cols, _p(out), cols, float(act_scale), _p(cs), _stream()), "nv_quant_rows_f8")
    return out, cs


def ln_fwd_f8(x: torch.Tensor, gamma, beta, out_scale: float, eps: float = 1e-5) -> torch.Tensor:
    _need_cuda(x)
    M, d = x.shape
    y = torch.empty((M, d), dtype=torch.uint8, device=x.device)
    check(lib.nv_ln_fwd_f8(_p(x), x.stride(0), M, d, _p(gamma), _p(beta), eps, float(out_scale), _p(y), d, _stream()), "nv_ln_fwd_f8")
    return y


def gemm_f8(epi: int, A8: torch.Tensor, B8: torch.Tensor, colscale: torch.Tensor, *, bias=None, aux_in=None, out_scale: float = 1.0, out=None):
    """C = epi((A8 B8^T) * colscale[n]); A8 [M, K], B8 [N, K] e4m3 bytes (uint8 tensors)."""
    _need_cuda(A8, B8)
    assert A8.dtype == torch.uint8 and B8.dtype == torch.uint8
    M, K = A8.shape
    N = B8.shape[0]
    odt = {EPI_STORE_BF16: op16(), EPI_STORE_F32: torch.float32, EPI_BIAS_RESID: torch.float32, EPI_BIAS_GELU_F8: torch.uint8}[epi]
    if out is None:
        out = torch.empty((M, N), dtype=odt, device=A8.device)
    check(lib.nv_gemm_f8(epi, M, N, K, _p(A8), A8.stride(0), _p(B8), B8.stride(0), _p(out), out.stride(0), _p(colscale), _p(bias), _p(aux_in),
                         0 if aux_in is None else aux_in.stride(0), float(out_scale), _stream()), "nv_gemm_f8")
    return out


def patch_ln_fwd_4d(x: torch.Tensor, p1: int, p2: int, pf: int, gamma, beta, eps: float = 1e-5, vol_sigma=None):
    """x f32 contiguous [Bo, H, W, D, T] -> tokens bf16 [Bo*T*N, P] of the T volumes of every sample (row (bo*T + t)*N + n)."""
    _need_cuda(x)
    assert x.dim() == 5 and x.is_contiguous() and x.dtype == torch.float32
    Bo, H, W, D, T = x.shape
    P = p1 * p2 * pf
    N = (D // pf) * (H // p1) * (W // p2)
    out = torch.empty((Bo * T * N, P), dtype=op16(), device=x.device)
    st = torch.empty((2, Bo * T * N), dtype=torch.float32, device=x.device)
    check(lib.nv_patch_ln_fwd_4d(_p(x), Bo, H, W, D, T, p1, p2, pf, _p(gamma), _p(beta), eps, _p(out), P, _p(st[0]), _p(st[1]), _p(vol_sigma),
                                 _stream()), "nv_patch_ln_fwd_4d")
    return out, st


def temporal_head_param_count(ff: int) -> int:
    return int(lib.nv_temporal_head_param_count(int(ff)))


def temporal_head_fwd(x: torch.Tensor, params: torch.Tensor, ff: int, eps: float = 1e-5, drop_seed: int = 0, drop_p: float = 0.0) -> torch.Tensor:
    """The 4D model's temporal head (NeuroEncoder.py:60-66): x f32 [B, T, 2] -> encoder layer -> mean over T -> Linear(2, 2) -> [B, 2]."""
    _need_cuda(x, params)
    assert x.dtype == torch.float32 and x.dim() == 3 and x.shape[2] == 2 and x.is_contiguous() and params.dtype == torch.float32
    assert params.is_contiguous() and params.numel() == temporal_head_param_count(ff)
    B, T, _ = x.shape
    out = torch.empty((B, 2), dtype=torch.float32, device=x.device)
    check(lib.nv_temporal_head_fwd(_p(x), B, T, int(ff), _p(params), eps, int(drop_seed), float(drop_p), _p(out), _stream()), "nv_temporal_head_fwd")
    return out


def temporal_head_bwd(x: torch.Tensor, params: torch.Tensor, ff: int, dout: torch.Tensor, grads: torch.Tensor, accumulate: bool = False,
                      want_dx: bool = False, eps: float = 1e-5, drop_seed: int = 0, drop_p: float = 0.0):
    """Gradients of temporal_head_fwd (forward recomputed from x, same seed / p): parameter gradients into the flat arena `grads`
    (added when accumulate), returns dx [B, T, 2] when want_dx."""
    _need_cuda(x, params, dout, grads)
    assert x.dtype == torch.float32 and x.dim() == 3 and x.shape[2] == 2 and x.is_contiguous()
    assert dout.dtype == torch.float32 and dout.shape == (x.shape[0], 2) and dout.is_contiguous()
    assert grads.dtype == torch.float32 and grads.is_contiguous() and grads.numel() == params.numel() == temporal_head_param_count(ff)
    B, T, _ = x.shape
    dx = torch.empty_like(x) if want_dx else None
    check(lib.nv_temporal_head_bwd(_p(x), B, T, int(ff), _p(params), eps, int(drop_seed), float(drop_p), _p(dout), _p(grads), int(bool(accumulate)),
                                   _p(dx), _stream()), "nv_temporal_head_bwd")
    return dx


def ln_fold_weight(W: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, bias: Optional[torch.Tensor] = None):
    """(Wg16 = op16(W diag(gamma)), colsum of the rounded Wg16 rows, folded bias W beta (+ bias)): the Linear behind a LayerNorm, folded (nv_ln_fold_weight)."""
    _need_cuda(W)
    N, K = W.shape
    Wg = torch.empty((N, K), dtype=op16(), device=W.device)
    cs = torch.empty(N, dtype=torch.float32, device=W.device)
    fb = torch.empty(N, dtype=torch.float32, device=W.device)
    check(lib.nv_ln_fold_weight(_p(W), W.stride(0), N, K, _p(gamma), _p(beta), _p(bias), _p(Wg), Wg.stride(0), _p(cs), _p(fb), _stream()), "nv_ln_fold_weight")
    return Wg, cs, fb


def gemm_resid_ln(A: torch.Tensor, W: torch.Tensor, bias: torch.Tensor, resid: torch.Tensor):
    """x = resid + (A W^T + bias) as f32, the same rows in the operand format, and the per-tile row statistics (nv_gemm_resid_ln)."""
    _need_cuda(A, W)
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    out16 = torch.empty((M, N), dtype=op16(), device=A.device)
    stats = torch.empty(lib.nv_ln_fold_stats_floats(M, N), dtype=torch.float32, device=A.device)
    check(lib.nv_gemm_resid_ln(M, N, K, _p(A), A.stride(0), _p(W), W.stride(0), _p(bias), _p(resid), resid.stride(0), _p(out), out.stride(0), _p(out16), out16.stride(0),
                               _p(stats), _stream()), "nv_gemm_resid_ln")
    return out, out16, stats


def gemm_lnfold(X16: torch.Tensor, Wg16: torch.Tensor, stats: torch.Tensor, colsum: torch.Tensor, fbias: torch.Tensor, gelu: bool = False, eps: float = 1e-5):
    """(gelu)(LayerNorm(x) W^T + b) from the un-normalised rows X16, their statistics and the folded weight (nv_gemm_lnfold)."""
    _need_cuda(X16, Wg16)
    M, K = X16.shape
    N = Wg16.shape[0]
    out = torch.empty((M, N), dtype=op16(), device=X16.device)
    check(lib.nv_gemm_lnfold(int(gelu), M, N, K, _p(X16), X16.stride(0), _p(Wg16), Wg16.stride(0), _p(stats), _p(colsum), _p(fbias), float(eps), _p(out), out.stride(0),
                             _stream()), "nv_gemm_lnfold")
    return out
