"""CPU restatement (eager PyTorch, fp32) of the NeuroViT hot path.  TEST INFRASTRUCTURE ONLY.

Restates, op for op, what the reference executes on this path:
  * ViT / Transformer / Attention / FeedForward      -> reference src/models/vit_3d.py:14-126
  * ViT3DEncoder permute, NeuroEncoder 3D/4D forward  -> reference src/models/NeuroEncoder.py:49-68,197-205
  * TemporalTransformer / ProjectionHead              -> reference src/models/NeuroEncoder.py:207-230
    (torch ``nn.TransformerEncoderLayer`` defaults: post-norm, ReLU, ff=2048)

Two modes
---------
``emulate_bf16=False``  exact fp32 restatement.  Pinned (<=1e-5 rel) against golden
    vectors produced by importing the reference (tests/golden/make_golden.py).
``emulate_bf16=True``   the same math with the HIP path's bf16 cast points
    (bf16 MFMA operands, fp32 accumulate, fp32 residual stream / LN / softmax),
    forward AND backward.  This is the checker the gfx950 kernels are gated
    against (<=1e-3 rel, see tests/).

Parameters are passed as a plain dict keyed exactly like the reference's
``ViT.state_dict()`` (e.g. ``transformer.layers.0.0.to_qkv.weight``).

Parity pin: the reference has no tests or golden vectors of its own
 - the pin is the set of fixtures under tests/golden/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-5  # nn.LayerNorm default (vit_3d.py:18,37,93,95,108)


# --------------------------------------------------------------------------- config

@dataclass
class ViTCfg:
    """Constructor arguments of the reference ViT (vit_3d.py:78)."""
    image_size: int
    image_patch_size: int
    frames: int
    frame_patch_size: int
    num_classes: int
    dim: int
    depth: int
    heads: int
    mlp_dim: int
    pool: str = "cls"
    channels: int = 3
    dim_head: int = 64
    image_width: int = 0        # vit_3d.py:80-81: image_size / image_patch_size may be (height, width) pairs; 0 = square
    patch_width: int = 0

    @property
    def hw(self) -> Tuple[int, int, int, int]:
        """(image height, image width, patch height, patch width)"""
        return self.image_size, self.image_width or self.image_size, self.image_patch_size, self.patch_width or self.image_patch_size

    @property
    def grid(self) -> Tuple[int, int, int]:
        H, Wd, p1, p2 = self.hw
        return (self.frames // self.frame_patch_size, H // p1, Wd // p2)

    @property
    def num_patches(self) -> int:
        f, h, w = self.grid
        return f * h * w

    @property
    def patch_dim(self) -> int:
        return self.channels * self.hw[2] * self.hw[3] * self.frame_patch_size

    @property
    def inner(self) -> int:
        return self.heads * self.dim_head


def strip_prefix(sd: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    """NeuroEncoder.py:27-31 - keep only keys under `prefix`, drop the prefix."""
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


# --------------------------------------------------------------------------- bf16 emulation helpers

# Named forward cast points of the HIP path (tools/cast_point_ablation.py switches them off one at a time to attribute the
# logits error to its sources): "xp" patch-LN output, "w" weights, "xn1"/"xn2" block LayerNorm outputs, "qkv", "p" softmax
# probabilities, "ao" attention output, "h" GELU output.  A point listed in CAST_OFF keeps fp32 there.
CAST_OFF = set()


class _Q8(torch.autograd.Function):
    """OCP e4m3 quantise-dequantise with a straight-through gradient (identity inside the representable range, zero where the value
    saturated).  A plain `.to(torch.float8_e4m3fn)` is NOT that under autograd: its backward casts the GRADIENT to e4m3 as well, which
    flushes everything below 2^-9 to zero - a training step through it learns nothing the HIP path (bf16 backward over the unquantised
    activations) computes."""

    @staticmethod
    def forward(ctx, x, scale):
        y = x * scale
        ctx.save_for_backward(y.abs() <= 448.0)
        return y.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32) / scale

    @staticmethod
    def backward(ctx, g):
        (inside,) = ctx.saved_tensors
        return g * inside, None


def _q8(x: torch.Tensor, scale) -> torch.Tensor:
    """OCP e4m3 quantise-dequantise (round to nearest even, saturating at +-448), fp32 storage: the fp8 path's cast points."""
    if not torch.is_tensor(scale):
        scale = torch.tensor(float(scale))
    return _Q8.apply(x, scale.detach())


class _LinearF8(torch.autograd.Function):
    """A Linear of the fp8 TRAINING forward as the HIP path computes it (csrc/engine.hip::nv_vit_forward_fp8_train + the unchanged bf16 backward):
    forward  y = q8(x * sx) / sx . q8_rows(w)^T   (e4m3 operands, fp32 accumulate);
    backward on the 16-bit copies the same forward kernels wrote, exactly as the bf16 path: dx = r(dy) r(w), dw = r(dy)^T r(x) - no straight-through
    estimator at all (the quantiser never sits in the backward graph), hence no saturation mask either."""

    @staticmethod
    def forward(ctx, x, w, sx):
        ctx.save_for_backward(x, w)
        with torch.no_grad():
            return F.linear(_q8(x, sx), _q8_rows(w))

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _r(dy)
        dx = torch.matmul(dy, _r(w))
        dw = torch.matmul(dy.reshape(-1, dy.shape[-1]).t(), _r(x).reshape(-1, x.shape[-1]))
        return dx, dw, None


class _GeluF8(torch.autograd.Function):
    """gelu(u) * mask of the fp8 training forward (the FC1 epilogue writes it as e4m3 for FC2 and as bf16 for the backward pass); backward as the bf16 path:
    dU = r(dH * mask * gelu'(r(u)))."""

    @staticmethod
    def forward(ctx, u, mask):
        ctx.save_for_backward(_r(u), mask)
        return F.gelu(u) * mask

    @staticmethod
    def backward(ctx, dh):
        u16, mask = ctx.saved_tensors
        return _r(dh * mask * _gelu_grad(u16)), None


def _q8_rows(w: torch.Tensor) -> torch.Tensor:
    """Per-output-row weight quantisation of csrc/quant.hip::quant_rows_f8_kernel (the row scales are constants for autograd)."""
    amax = w.detach().abs().amax(dim=1, keepdim=True)
    sw = torch.where(amax > 0, 448.0 / amax, torch.ones_like(amax))
    return _q8(w, sw)


# 16-bit operand format the emulation rounds to: bfloat16 (the HIP path's default) or float16 (NV_OPERAND_FP16 - the reference's own
# autocast(float16) arithmetic, src/Trainer.py:68).  Every cast point goes through _r, so ONE switch restates either product path.
_ROUND_DTYPE = torch.bfloat16


class operand_format:
    """Context manager: `with operand_format("fp16"): ...` - emulate_bf16=True forwards AND their .backward() inside it round to
    float16 instead of bfloat16 (overflow gives inf, as the hardware conversion does)."""

    def __init__(self, fmt: str):
        self.dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[fmt]

    def __enter__(self):
        global _ROUND_DTYPE
        self.before, _ROUND_DTYPE = _ROUND_DTYPE, self.dtype
        return self

    def __exit__(self, *exc):
        global _ROUND_DTYPE
        _ROUND_DTYPE = self.before
        return False


def _r(x: torch.Tensor, point: Optional[str] = None) -> torch.Tensor:
    """Round-to-nearest-even to the 16-bit operand format (bf16 unless inside operand_format("fp16")), keep fp32 storage.  Autograd: identity."""
    if point is not None and point in CAST_OFF:
        return x
    return x.to(_ROUND_DTYPE).to(torch.float32)


class _RoundGrad(torch.autograd.Function):
    """Identity forward; backward rounds the incoming gradient to bf16 (the HIP
    path feeds gradients to MFMA as bf16 operands)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g)


def _gelu_grad(u: torch.Tensor) -> torch.Tensor:
    # d/du [ u * Phi(u) ] = Phi(u) + u * phi(u)   (exact-erf GELU, vit_3d.py:20)
    cdf = 0.5 * (1.0 + torch.erf(u * (1.0 / math.sqrt(2.0))))
    pdf = torch.exp(-0.5 * u * u) * (1.0 / math.sqrt(2.0 * math.pi))
    return cdf + u * pdf


class _GeluEmu(torch.autograd.Function):
    """h = bf16(gelu(u_fp32) * mask); backward dU = bf16(dH * mask * gelu'(bf16(u))).  mask = dropout mask (or 1)."""

    @staticmethod
    def forward(ctx, u, mask):
        ctx.save_for_backward(_r(u), mask)
        return _r(F.gelu(u) * mask, "h")

    @staticmethod
    def backward(ctx, dh):
        u16, mask = ctx.saved_tensors
        return _r(dh * mask * _gelu_grad(u16)), None


class _AttnEmu(torch.autograd.Function):
    """Flash-style attention with the HIP kernel's cast points.

    forward : online softmax over 64-key tiles (two half-range states merged at the end): S = q k^T (fp32 acc), p = exp(scale*S - m_running),
              l += sum p (fp32), O += bf16(p) @ v (both rescaled when the running max moves),
              returns bf16(O / l) and LSE.
    backward: delta = rowsum(dO*O); P = exp(scale*S - LSE); dV = bf16(P)^T dO;
              dP = dO v^T; dS = bf16(P*(dP-delta)); dQ = scale*dS k; dK = scale*dS^T q;
              all outputs rounded to bf16.
    q, k, v : [B, h, n, dh] (bf16-representable values in fp32 storage).
    """

    TK = 64   # key tile of the kernel's online softmax (csrc/attention.hip)

    @staticmethod
    def forward(ctx, q, k, v, scale, mask=None):
        # Online softmax over 64-key tiles, exactly the kernel's schedule: P is rounded to bf16 relative to
        # the RUNNING row max of its tile (r(c*x) != c*r(x), so the rounding point matters at the 1e-3 level).
        # The key tiles are processed as TWO independent online-softmax states (tiles [0, nh) and [nh, nkt), nh = ceil(nkt / 2):
        # in the LDS-resident kernel two partner waves take one half each) that are merged at the end.
        n = k.shape[-2]
        nkt = (n + _AttnEmu.TK - 1) // _AttnEmu.TK
        nh = (nkt + 1) // 2
        states = []
        for t0, t1 in ((0, nh), (nh, nkt)):
            m = torch.full(q.shape[:-1] + (1,), float("-inf"), dtype=q.dtype)
            l = torch.zeros_like(m)
            o = torch.zeros_like(q)
            for kt in range(t0, t1):
                k0 = kt * _AttnEmu.TK
                s = torch.matmul(q, k[..., k0:k0 + _AttnEmu.TK, :].transpose(-1, -2)) * scale
                mnew = torch.maximum(m, s.amax(dim=-1, keepdim=True))
                alpha = torch.exp(m - mnew)
                p = torch.exp(s - mnew)
                l = l * alpha + p.sum(dim=-1, keepdim=True)
                pm = p if mask is None else p * mask[..., k0:k0 + _AttnEmu.TK]      # dropout hits P.V, not the normaliser
                o = o * alpha + torch.matmul(_r(pm, "p"), v[..., k0:k0 + _AttnEmu.TK, :])
                m = mnew
            states.append((m, l, o))
        (m0, l0, o0), (m1, l1, o1) = states
        m = torch.maximum(m0, m1)
        a0, a1 = torch.exp(m0 - m), torch.exp(m1 - m)          # exp(-inf) = 0 when the second half is empty (n <= 64)
        l = l0 * a0 + l1 * a1
        o = o0 * a0 + o1 * a1
        o = _r(o / l, "ao")
        lse = m + torch.log(l)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.scale = scale
        ctx.mask = mask
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        scale = ctx.scale
        do = _r(do)
        delta = (do * o).sum(dim=-1, keepdim=True)
        p = torch.exp(torch.matmul(q, k.transpose(-1, -2)) * scale - lse)
        mk = 1.0 if ctx.mask is None else ctx.mask
        dv = torch.matmul(_r(p * mk).transpose(-1, -2), do)
        dp = torch.matmul(do, v.transpose(-1, -2)) * mk
        ds = _r(p * (dp - delta))
        dq = torch.matmul(ds, k) * scale
        dk = torch.matmul(ds.transpose(-1, -2), q) * scale
        return _r(dq), _r(dk), _r(dv), None, None


# --------------------------------------------------------------------------- dropout masks of the HIP path
# The product uses a counter-based mask (csrc/common.h nv_hash64 / DropCfg): element idx of a site is kept iff
# a 16-bit field of hash(site_seed, idx >> 2) >= p * 2^16 and scaled by 1/(1-p).  Restated here bit for bit so that dropout runs can be
# checked against the oracle with IDENTICAL masks (torch's Philox stream cannot be matched - SURVEY.md 5 "RNG").
_M64 = (1 << 64) - 1


def site_seed(seed: int, site: int) -> int:
    """csrc/engine.hip::site_seed - site = 4*layer + {0 attn probs, 1 to_out, 2 FF hidden, 3 FF out}; 4*depth = embedding."""
    return (seed ^ ((0x9E3779B97F4A7C15 * (site + 1)) & _M64)) & _M64


def drop_mask(seed: int, p: float, shape) -> torch.Tensor:
    """fp32 tensor of `shape` holding 0 or 1/(1-p); element index = row-major position (csrc/common.h::drop_factor4):
    one 64-bit hash per group of four consecutive elements, one 16-bit field per element, keep iff field >= p * 2^16."""
    if p <= 0:
        return torch.ones(shape)
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        grp, lane = idx >> np.uint64(2), idx & np.uint64(3)
        x = ((grp + np.uint64(0x9E3779B97F4A7C15)) * np.uint64(0xBF58476D1CE4E5B9)) ^ np.uint64(seed)
        x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
        h = (x >> (np.uint64(16) * lane)) & np.uint64(0xFFFF)
    thresh = np.uint64(0x10000) if p >= 1 else np.uint64(int(np.float32(p).astype(np.float64) * 65536.0))
    scale = 0.0 if p >= 1 else float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
    return torch.from_numpy(np.where(h >= thresh, np.float32(scale), np.float32(0)).astype(np.float32).reshape(shape))


def attn_drop_mask(seed: int, p: float, B: int, heads: int, n: int) -> torch.Tensor:
    """Mask of the attention probabilities [B, heads, n, n]: the kernels index element (bh, q, key) as
    (bh * n + q) * npad + key with npad = n rounded up to a multiple of 4 (csrc/attention.hip::fwd_softmax)."""
    npad = (n + 3) // 4 * 4
    return drop_mask(seed, p, (B, heads, n, npad))[..., :n].contiguous()


# --------------------------------------------------------------------------- A1: patchify

def patchify(video: torch.Tensor, p1: int, p2: int, pf: int) -> torch.Tensor:
    """Rearrange('b c (f pf) (h p1) (w p2) -> b (f h w) (p1 p2 pf c)')  (vit_3d.py:92).

    Pure index map, bit exact.  video: [B, C, F, H, W] (any strides)."""
    B, C, Fr, H, W = video.shape
    f, h, w = Fr // pf, H // p1, W // p2
    v = video.reshape(B, C, f, pf, h, p1, w, p2)
    #            b  f  h  w  p1 p2 pf c
    v = v.permute(0, 2, 4, 6, 5, 7, 3, 1)
    return v.reshape(B, f * h * w, p1 * p2 * pf * C)


def zscore_crop(raw: np.ndarray, crop=((1, None), (10, -9), (1, None)), eps: float = 1e-8) -> np.ndarray:
    """Row A0: the dataset's crop + z-score, one sample at a time (src/data/DatasetADNI.py:212-213 for [X,Y,Z],
    src/data/DatasetADNI_4D.py:86-87 for [X,Y,Z,T]): numpy mean / population std over the cropped sample, float32 result."""
    out = []
    for sample in raw:
        c = sample[tuple(slice(lo, hi) for lo, hi in crop)]
        out.append(((c - c.mean()) / (c.std() + eps)).astype(np.float32))
    return np.stack(out)


def fmri_to_video(fmri: torch.Tensor) -> torch.Tensor:
    """ViT3DEncoder.forward (NeuroEncoder.py:200-202): [B,H,W,D] -> [B,1,D,H,W] (a view)."""
    return fmri.permute(0, 3, 1, 2).unsqueeze(1)


def patch_index_map(S: int, p: int) -> np.ndarray:
    """Integer restatement of A1 for the NeuroEncoder layout.

    For a cubic volume V[x, y, z] of side S (dataset layout [H, W, D], z contiguous)
    returns idx[N, P] (int64) with tokens[n, k] == V.flat[idx[n, k]]:
        n = (z//p) G^2 + (x//p) G + (y//p),   k = (x%p) p^2 + (y%p) p + (z%p).
    """
    G = S // p
    x, y, z = np.meshgrid(np.arange(S), np.arange(S), np.arange(S), indexing="ij")
    n = (z // p) * G * G + (x // p) * G + (y // p)
    k = (x % p) * p * p + (y % p) * p + (z % p)
    flat = (x * S + y) * S + z
    idx = np.empty((G ** 3, p ** 3), dtype=np.int64)
    idx[n.ravel(), k.ravel()] = flat.ravel()
    return idx


# --------------------------------------------------------------------------- blocks

def _linear(x, w, b, emulate, xpoint=None):
    if emulate:
        y = _RoundGrad.apply(F.linear(_r(x, xpoint), _r(w, "w")))
        return y if b is None else y + b
    return F.linear(x, w, b)


def attention(sd, pre, x, heads, dim_head, emulate=False, taps=None, drop=None, f8=None):
    """Attention.forward (vit_3d.py:48-60).  `pre` = 'transformer.layers.{i}.0.'.
    drop = (p, seed_attn, seed_out) applies the HIP path's dropout masks (train mode, p > 0)."""
    B, n, d = x.shape
    inner = heads * dim_head
    xn = F.layer_norm(x, (d,), sd[pre + "norm.weight"], sd[pre + "norm.bias"], LN_EPS)
    if taps is not None:
        taps[pre + "norm.out"] = xn
    if f8 is not None:          # fp8 path: LN output and to_qkv weight in e4m3, fp32 accumulate, bf16 qkv (f8 = scale of the LN output, or
                                # (that, scale of the attention output or None): the out-projection in e4m3 too)
        qkv = _r(_LinearF8.apply(xn, sd[pre + "to_qkv.weight"], f8[0] if isinstance(f8, tuple) else f8), "qkv")
    elif emulate:
        xn = _r(xn, "xn1")
        qkv = _r(_linear(xn, sd[pre + "to_qkv.weight"], None, True, "xn1"), "qkv")
    else:
        qkv = F.linear(xn, sd[pre + "to_qkv.weight"])
    q, k, v = qkv.chunk(3, dim=-1)
    # 'b n (h d) -> b h n d'
    q, k, v = (t.reshape(B, n, heads, dim_head).permute(0, 2, 1, 3) for t in (q, k, v))
    scale = dim_head ** -0.5
    amask = attn_drop_mask(drop[1], drop[0], B, heads, n) if drop else None
    if emulate:
        out = _AttnEmu.apply(q, k, v, scale, amask)
    else:
        dots = torch.matmul(q, k.transpose(-1, -2)) * scale
        attn = torch.softmax(dots, dim=-1)
        out = torch.matmul(attn if amask is None else attn * amask, v)
        if taps is not None:
            taps[pre + "attn.rowsum"] = attn.sum(-1)
    # 'b h n d -> b n (h d)'
    out = out.permute(0, 2, 1, 3).reshape(B, n, inner)
    if taps is not None:
        taps[pre + "q"], taps[pre + "k"], taps[pre + "v"] = q, k, v
        taps[pre + "attn.out"] = out
    if (pre + "to_out.0.weight") in sd:            # project_out (vit_3d.py:32,43-46)
        if f8 is not None and isinstance(f8, tuple) and f8[1]:    # fp8 path: attention output and to_out weight in e4m3 (f8[1] = its scale)
            return _LinearF8.apply(out, sd[pre + "to_out.0.weight"], f8[1]) + sd[pre + "to_out.0.bias"]
        out = _linear(out, sd[pre + "to_out.0.weight"], sd[pre + "to_out.0.bias"], emulate, "ao")
        if drop:
            out = out * drop_mask(drop[2], drop[0], (B * n, out.shape[-1])).reshape(out.shape)
    return out


def feed_forward(sd, pre, x, emulate=False, drop=None, f8=None):
    """FeedForward.forward (vit_3d.py:16-26).  `pre` = 'transformer.layers.{i}.1.'.  drop = (p, seed_hidden, seed_out)."""
    d = x.shape[-1]
    rows = x.shape[0] * x.shape[1]
    hmask = drop_mask(drop[1], drop[0], (rows, sd[pre + "net.1.weight"].shape[0])).reshape(x.shape[0], x.shape[1], -1) if drop else None
    omask = drop_mask(drop[2], drop[0], (rows, d)).reshape(x.shape) if drop else None
    xn = F.layer_norm(x, (d,), sd[pre + "net.0.weight"], sd[pre + "net.0.bias"], LN_EPS)
    if f8 is not None:          # fp8 path: f8 = (scale of the LN output, scale of the GELU output); train mode: the two dropout masks as below
        u = _LinearF8.apply(xn, sd[pre + "net.1.weight"], f8[0]) + sd[pre + "net.1.bias"]
        h = _GeluF8.apply(u, torch.ones(()) if hmask is None else hmask)
        y = _RoundGrad.apply(_LinearF8.apply(h, sd[pre + "net.4.weight"], f8[1])) + sd[pre + "net.4.bias"]
        return y if omask is None else y * omask
    if emulate:
        u = _linear(_r(xn, "xn2"), sd[pre + "net.1.weight"], sd[pre + "net.1.bias"], True, "xn2")
        h = _GeluEmu.apply(u, torch.ones(()) if hmask is None else hmask)
        y = _linear(h, sd[pre + "net.4.weight"], sd[pre + "net.4.bias"], True, "h")
        return y if omask is None else y * omask
    h = F.gelu(F.linear(xn, sd[pre + "net.1.weight"], sd[pre + "net.1.bias"]))
    if hmask is not None:
        h = h * hmask
    y = F.linear(h, sd[pre + "net.4.weight"], sd[pre + "net.4.bias"])
    return y if omask is None else y * omask


def patch_embed(sd, cfg: ViTCfg, video, emulate=False, taps=None, drop=None):
    """to_patch_embedding + cls/pos + emb dropout (vit_3d.py:91-96,113-119).  drop = (p_emb, seed) or None."""
    tok = patchify(video, cfg.hw[2], cfg.hw[3], cfg.frame_patch_size)
    P, d = cfg.patch_dim, cfg.dim
    a2 = F.layer_norm(tok, (P,), sd["to_patch_embedding.1.weight"], sd["to_patch_embedding.1.bias"], LN_EPS)
    a3 = _linear(a2, sd["to_patch_embedding.2.weight"], sd["to_patch_embedding.2.bias"], emulate, "xp")
    a4 = F.layer_norm(a3, (d,), sd["to_patch_embedding.3.weight"], sd["to_patch_embedding.3.bias"], LN_EPS)
    B, n, _ = a4.shape
    cls = sd["cls_token"].expand(B, 1, d)
    x = torch.cat((cls, a4), dim=1)
    x = x + sd["pos_embedding"][:, : n + 1]
    if drop:
        x = x * drop_mask(drop[1], drop[0], (B * (n + 1), d)).reshape(x.shape)
    if taps is not None:
        taps["A1"], taps["A2"], taps["A3"], taps["A4"], taps["A5"] = tok, a2, a3, a4, x
    return x


def vit_forward(sd: Dict[str, torch.Tensor], cfg: ViTCfg, video: torch.Tensor,
                emulate_bf16: bool = False, taps: Optional[dict] = None,
                dropout: Optional[Tuple[float, float, int]] = None, fp8_scales=None) -> torch.Tensor:
    """ViT.forward (vit_3d.py:112-126).  video: [B, C, F, H, W].
    dropout = None (eval / p = 0) or (p_blocks, p_embedding, seed): train-mode dropout with the HIP path's masks."""
    dp = dropout if dropout and (dropout[0] > 0 or dropout[1] > 0) else None
    x = patch_embed(sd, cfg, video, emulate_bf16, taps, (dp[1], site_seed(dp[2], 4 * cfg.depth)) if dp and dp[1] > 0 else None)
    for i in range(cfg.depth):
        pa, pf = f"transformer.layers.{i}.0.", f"transformer.layers.{i}.1."
        da = (dp[0], site_seed(dp[2], 4 * i + 0), site_seed(dp[2], 4 * i + 1)) if dp and dp[0] > 0 else None
        df = (dp[0], site_seed(dp[2], 4 * i + 2), site_seed(dp[2], 4 * i + 3)) if dp and dp[0] > 0 else None
        # fp8 inference path (csrc/engine.hip::nv_vit_forward_fp8); rows of 4 scales carry the attention-output scale (out-projection in e4m3)
        f8a = None if fp8_scales is None else ((fp8_scales[i][0], fp8_scales[i][3] or None) if len(fp8_scales[i]) > 3 else fp8_scales[i][0])
        f8f = (fp8_scales[i][1], fp8_scales[i][2]) if fp8_scales is not None else None
        x = attention(sd, pa, x, cfg.heads, cfg.dim_head, emulate_bf16, taps, da, f8a) + x
        x = feed_forward(sd, pf, x, emulate_bf16, df, f8f) + x
        if taps is not None:
            taps[f"block{i}"] = x
    x = x.mean(dim=1) if cfg.pool == "mean" else x[:, 0]
    x = F.layer_norm(x, (cfg.dim,), sd["mlp_head.0.weight"], sd["mlp_head.0.bias"], LN_EPS)
    return F.linear(x, sd["mlp_head.1.weight"], sd["mlp_head.1.bias"])


# --------------------------------------------------------------------------- NeuroEncoder level

def neuro_cfg(config: dict) -> ViTCfg:
    """ViT3DEncoder.__init__ (NeuroEncoder.py:171-195): config dict -> ViT ctor args.
    Optional TRAINING_VIT_* size keys default to the reference's hard-coded constants."""
    S, p = config["TRAINING_VIT_INPUT_SIZE"], config["TRAINING_VIT_PATCH_SIZE"]
    ncls = (S // config["GRADCAM_CUBE_SIZE"]) ** 3 if config["DATASET_NAME"] == "gradcam" else 2
    return ViTCfg(image_size=S, image_patch_size=p, frames=S, frame_patch_size=p, num_classes=ncls,
                  dim=config.get("TRAINING_VIT_DIM", 1024), depth=config.get("TRAINING_VIT_DEPTH", 6),
                  heads=config.get("TRAINING_VIT_HEADS", 8), mlp_dim=config.get("TRAINING_VIT_MLP_DIM", 2048),
                  pool="cls", channels=1, dim_head=config.get("TRAINING_VIT_DIM_HEAD", 64))


TEMPORAL_SITE = 0x9E3779B97F4A7C15      # csrc/temporal.hip::th_drop: site k of the temporal head uses seed ^ (k * this), k = 1..4


def temporal_transformer(sd, pre, x, drop=None):
    """nn.TransformerEncoder(TransformerEncoderLayer(d_model=2, nhead=2, batch_first=True), 1)
    (NeuroEncoder.py:211-212) restated: post-norm, ReLU.  `pre` = 'temporal_transformer.transformer.layers.0.'.
    drop = None: eval mode.  drop = (p, seed): train mode with the native kernel's counter-based masks at the layer's four
    nn.Dropout sites (attention probabilities, dropout1, the FeedForward's inner dropout, dropout2) - torch's own Philox
    masks cannot be reproduced, the sites and the 1/(1-p) scaling are nn.TransformerEncoderLayer's."""
    B, T, E = x.shape
    H = 2
    dh = E // H
    ff = sd[pre + "linear1.weight"].shape[0]
    one = lambda k, shape: 1.0 if drop is None else drop_mask((drop[1] ^ (TEMPORAL_SITE * k)) & 0xFFFFFFFFFFFFFFFF, drop[0], shape).to(x.dtype)
    qkv = F.linear(x, sd[pre + "self_attn.in_proj_weight"], sd[pre + "self_attn.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    q, k, v = (t.reshape(B, T, H, dh).permute(0, 2, 1, 3) for t in (q, k, v))
    a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh), dim=-1) * one(1, (B, H, T, T))
    o = torch.matmul(a, v).permute(0, 2, 1, 3).reshape(B, T, E)
    o = F.linear(o, sd[pre + "self_attn.out_proj.weight"], sd[pre + "self_attn.out_proj.bias"]) * one(2, (B, T, E))
    x = F.layer_norm(x + o, (E,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], LN_EPS)
    f = F.linear(F.relu(F.linear(x, sd[pre + "linear1.weight"], sd[pre + "linear1.bias"])) * one(3, (B, T, ff)),
                 sd[pre + "linear2.weight"], sd[pre + "linear2.bias"]) * one(4, (B, T, E))
    return F.layer_norm(x + f, (E,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], LN_EPS)


def temporal_head(sd, per_volume, drop=None):
    """NeuroEncoder.py:60-66: temporal transformer -> mean over time -> projection head, on the per-timepoint logits [B, T, 2]."""
    enc = temporal_transformer(sd, "temporal_transformer.transformer.layers.0.", per_volume, drop).mean(dim=1)
    return F.linear(enc, sd["projection_head.projection_head.weight"], sd["projection_head.projection_head.bias"])


def neuro_forward(sd: Dict[str, torch.Tensor], config: dict, fmri: torch.Tensor,
                  emulate_bf16: bool = False, taps: Optional[dict] = None) -> torch.Tensor:
    """NeuroEncoder.forward (NeuroEncoder.py:49-68).  `sd` keyed like NeuroEncoder.state_dict()."""
    cfg = neuro_cfg(config)
    vsd = strip_prefix(sd, "volume_encoder.vit3d.")
    if config["TRAINING_DIM"] == 3:
        return vit_forward(vsd, cfg, fmri_to_video(fmri), emulate_bf16, taps)
    f = fmri.permute(0, 4, 1, 2, 3)
    B, T, H, W, D = f.shape
    vols = f.reshape(B * T, H, W, D)
    enc = vit_forward(vsd, cfg, fmri_to_video(vols), emulate_bf16, taps).reshape(B, T, -1)
    return temporal_head(sd, enc)


# --------------------------------------------------------------------------- Grad-CAM (§8f F1)

def grad_cam(activations: torch.Tensor, gradients: torch.Tensor, S: int, p: int, threshold: float):
    """NeuroEncoder.get_attention_map steps 1-6 (NeuroEncoder.py:101-131) given the hooked
    activation / gradient of the last block's attention LayerNorm output ([1, n, d])."""
    weights = gradients.mean(dim=2, keepdim=True)
    cam = (weights * activations).sum(dim=2)[:, 1:]
    G = S // p
    cam = F.relu(cam.reshape(1, G, G, G))
    cam = (cam - cam.min()) / (cam.max() - cam.min() + 1e-8)
    thr = np.percentile(cam.numpy(), 100 - threshold)
    m = torch.from_numpy(np.where(cam.numpy() >= thr, cam.numpy(), 0)).unsqueeze(0)
    return F.interpolate(m, size=(S, S, S), mode="trilinear", align_corners=False).squeeze()


# --------------------------------------------------------------------------- algorithmic work

def flops_forward(cfg: ViTCfg) -> float:
    """Algorithmic FLOPs per volume, forward (2*MAC; LN/softmax/GELU/bias excluded)."""
    N, n, P, d, inner, m, L, C = (cfg.num_patches, cfg.num_patches + 1, cfg.patch_dim, cfg.dim,
                                  cfg.inner, cfg.mlp_dim, cfg.depth, cfg.num_classes)
    per_layer = 2 * n * d * 3 * inner + 2 * n * n * inner + 2 * n * n * inner + 2 * n * inner * d + 4 * n * d * m
    return 2.0 * N * P * d + L * per_layer + 2 * d * C
