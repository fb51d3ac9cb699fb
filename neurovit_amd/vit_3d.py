"""Drop-in for the reference's src/models/vit_3d.py on MI355X.

Same classes, constructor signatures, attribute paths and state_dict keys as the reference
(vit_3d.py:14-126): FeedForward, Attention, Transformer, ViT.  The module tree is built from the same
torch building blocks in the same order, so `torch.manual_seed(s)` yields bit-identical initial
weights to the reference; the compute, however, never goes through those blocks:

  ViT.forward runs the whole encoder through the native gfx950 engine (csrc/engine.hip) in ONE
  C-ABI call, and its backward in one more.  Parameters are views into a flat fp32 arena with a
  bf16 shadow (see engine.py), gradients are written straight into a flat gradient arena that
  `param.grad` views.

There is no CPU / eager fallback: a CPU input raises.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from . import _cabi, engine, ops


def pair(t):
    return t if isinstance(t, tuple) else (t, t)


def _w16(w: torch.Tensor) -> torch.Tensor:
    """16-bit MFMA operand copy (the process's current operand format, _cabi.set_operand_format) of an fp32 [out, in] weight
    (standalone modules; inside a ViT the arena's shadow serves)."""
    return ops.cast_bf16(w.detach().reshape(w.shape[0], -1).float())


def _new_seed(p: float, training: bool) -> int:
    return int(torch.randint(0, 2 ** 62, (1,)).item()) if (training and p > 0) else 0


def _as_rows(x: torch.Tensor):
    if not x.is_cuda:
        raise RuntimeError("neurovit_amd: module inputs must live on the MI355X (cuda) device - there is no CPU fallback")
    d = x.shape[-1]
    return x.reshape(-1, d).float().contiguous(), d


class _FeedForwardFn(torch.autograd.Function):
    """vit_3d.py:16-26 standalone: LayerNorm -> Linear -> exact-erf GELU -> Dropout -> Linear -> Dropout, the same gfx950
    kernels (and cast points) the fused engine runs, one C-ABI call per stage."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w1, b1, w2, b2, p, seeds):
        x2, d = _as_rows(x)
        xn, st = ops.ln_fwd(x2, gamma.detach(), beta.detach())
        w1_16, w2_16 = _w16(w1), _w16(w2)
        u = torch.empty((x2.shape[0], w1.shape[0]), dtype=ops.op16(), device=x.device)
        h = ops.gemm(ops.NT, ops.EPI_BIAS_GELU, xn, w1_16, bias=b1.detach(), aux_out=u, drop_seed=seeds[0], drop_p=p if seeds[0] else 0.0)
        y = ops.gemm(ops.NT, ops.EPI_BIAS_F32, h, w2_16, bias=b2.detach())
        if seeds[1]:
            y = ops.dropout_apply(y, seeds[1], p, want16=False, want32=True)[1]
        ctx.save_for_backward(x2, xn, st, u, h, w1_16, w2_16, gamma.detach())
        ctx.p, ctx.seeds, ctx.shape = p, seeds, x.shape
        return y.view(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, xn, st, u, h, w1_16, w2_16, gamma = ctx.saved_tensors
        p, seeds = ctx.p, ctx.seeds
        dy2 = dy.reshape(-1, dy.shape[-1]).float().contiguous()
        dy16 = ops.dropout_apply(dy2, seeds[1], p if seeds[1] else 0.0)[0]
        dw2 = ops.gemm(ops.TN, ops.EPI_STORE_F32, dy16, h)
        db2 = ops.colsum_bf16(dy16)
        du = ops.gemm(ops.NN, ops.EPI_DGELU, dy16, w2_16, aux_in=u, drop_seed=seeds[0], drop_p=p if seeds[0] else 0.0)
        dw1 = ops.gemm(ops.TN, ops.EPI_STORE_F32, du, xn)
        db1 = ops.colsum_bf16(du)
        dxn = ops.gemm(ops.NN, ops.EPI_STORE_F32, du, w1_16)
        dx, _, dgamma, dbeta, _ = ops.ln_bwd(dxn, x2, st, gamma, want_g16=False)
        return dx.view(ctx.shape), dgamma, dbeta, dw1, db1, dw2, db2, None, None


class _AttentionFn(torch.autograd.Function):
    """vit_3d.py:48-60 standalone: LayerNorm -> to_qkv -> softmax(q k^T * scale) (+Dropout) -> attn v -> to_out (+Dropout)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, wqkv, wo, bo, heads, dim_head, p, seeds):
        if x.dim() != 3:
            raise ValueError("neurovit_amd.Attention: expected x of shape [batch, tokens, dim]")
        B, n, _ = x.shape
        x2, d = _as_rows(x)
        xn, st = ops.ln_fwd(x2, gamma.detach(), beta.detach())
        wqkv16, wo16 = _w16(wqkv), _w16(wo)
        qkv = ops.gemm(ops.NT, ops.EPI_STORE_BF16, xn, wqkv16)
        ao, lse = ops.attn_fwd(qkv, B, n, heads, dim_head, drop_seed=seeds[0], drop_p=p if seeds[0] else 0.0)
        y = ops.gemm(ops.NT, ops.EPI_BIAS_F32, ao, wo16, bias=bo.detach())
        if seeds[1]:
            y = ops.dropout_apply(y, seeds[1], p, want16=False, want32=True)[1]
        ctx.save_for_backward(x2, xn, st, qkv, ao, lse, wqkv16, wo16, gamma.detach())
        ctx.meta = (B, n, heads, dim_head, p, seeds, x.shape)
        return y.view(B, n, wo.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, xn, st, qkv, ao, lse, wqkv16, wo16, gamma = ctx.saved_tensors
        B, n, heads, dim_head, p, seeds, shape = ctx.meta
        dy2 = dy.reshape(-1, dy.shape[-1]).float().contiguous()
        dy16 = ops.dropout_apply(dy2, seeds[1], p if seeds[1] else 0.0)[0]
        dwo = ops.gemm(ops.TN, ops.EPI_STORE_F32, dy16, ao)
        dbo = ops.colsum_bf16(dy16)
        dao = ops.gemm(ops.NN, ops.EPI_STORE_BF16, dy16, wo16)
        dqkv, _ = ops.attn_bwd(qkv, ao, dao, lse, B, n, heads, dim_head, drop_seed=seeds[0], drop_p=p if seeds[0] else 0.0)
        dwqkv = ops.gemm(ops.TN, ops.EPI_STORE_F32, dqkv, xn)
        dxn = ops.gemm(ops.NN, ops.EPI_STORE_F32, dqkv, wqkv16)
        dx, _, dgamma, dbeta, _ = ops.ln_bwd(dxn, x2, st, gamma, want_g16=False)
        return dx.view(shape), dgamma, dbeta, dwqkv, dwo, dbo, None, None, None, None


class FeedForward(nn.Module):
    """vit_3d.py:14-26 - parameter container (net.0 LayerNorm, net.1 Linear, net.4 Linear)."""

    def __init__(self, dim, hidden_dim, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(
            nn.LayerNorm(dim),
            nn.Linear(dim, hidden_dim),
            nn.GELU(),
            nn.Dropout(dropout),
            nn.Linear(hidden_dim, dim),
            nn.Dropout(dropout)
        )

    def forward(self, x):
        """Standalone use (vit_3d.py:25-26; inside ViT.forward the block runs fused in the native engine): x [..., dim] fp32
        on the device -> net(x), differentiable, dropout honoured in train mode."""
        ln, fc1, fc2 = self.net[0], self.net[1], self.net[4]
        p = float(self.net[3].p)
        seeds = (_new_seed(p, self.training), _new_seed(p, self.training))
        return _FeedForwardFn.apply(x, ln.weight, ln.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias, p, seeds)


class Attention(nn.Module):
    """vit_3d.py:28-60 - parameter container (norm, to_qkv without bias, to_out.0)."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        project_out = not (heads == 1 and dim_head == dim)
        self.heads = heads
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.norm = nn.LayerNorm(dim)
        self.attend = nn.Softmax(dim=-1)
        self.dropout = nn.Dropout(dropout)
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(
            nn.Linear(inner_dim, dim),
            nn.Dropout(dropout)
        ) if project_out else nn.Identity()

    def forward(self, x):
        """Standalone use (vit_3d.py:48-60): x [batch, tokens, dim] fp32 on the device -> to_out(attention(norm(x)))."""
        if self.dim_head % 8 or not 8 <= self.dim_head <= 128:
            raise NotImplementedError("neurovit_amd: dim_head must be a multiple of 8 up to 128 (64, the vit_3d.py:78 default and "
                                      "the only value the NeuroEncoder path uses, runs the MFMA attention kernels; the others scalar ones)")
        p = float(self.dropout.p)
        if isinstance(self.to_out, nn.Identity):
            # heads == 1, dim_head == dim: no projection and no trailing dropout (vit_3d.py:43-46) - the same kernels with the
            # identity as the weight (x I + 0 is exact) and the output-dropout site off
            dim = self.heads * self.dim_head
            eye, zero = torch.eye(dim, device=x.device), torch.zeros(dim, device=x.device)
            return _AttentionFn.apply(x, self.norm.weight, self.norm.bias, self.to_qkv.weight, eye, zero, self.heads, self.dim_head, p,
                                      (_new_seed(p, self.training), 0))
        seeds = (_new_seed(p, self.training), _new_seed(p, self.training))
        return _AttentionFn.apply(x, self.norm.weight, self.norm.bias, self.to_qkv.weight, self.to_out[0].weight, self.to_out[0].bias,
                                  self.heads, self.dim_head, p, seeds)


class Transformer(nn.Module):
    """vit_3d.py:62-75."""

    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.):
        super().__init__()
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout),
                FeedForward(dim, mlp_dim, dropout=dropout)
            ]))

    def forward(self, x):
        """Standalone use (vit_3d.py:72-75): pre-norm residual blocks, no final LayerNorm."""
        for attn, ff in self.layers:
            x = attn(x) + x
            x = ff(x) + x
        return x


class PatchRearrange(nn.Module):
    """Placeholder for einops `Rearrange('b c (f pf) (h p1) (w p2) -> b (f h w) (p1 p2 pf c)')`
    (vit_3d.py:92): keeps `to_patch_embedding.{1,2,3}` state_dict indices.  The index map itself is
    executed inside the patch-gather kernel (csrc/norm.hip::patch_ln_fwd_kernel)."""

    def __init__(self, p1, p2, pf):
        super().__init__()
        self.p1, self.p2, self.pf = p1, p2, pf

    def extra_repr(self):
        return f"'b c (f pf) (h p1) (w p2) -> b (f h w) (p1 p2 pf c)', p1={self.p1}, p2={self.p2}, pf={self.pf}"


class _ViTFunction(torch.autograd.Function):
    """Whole-encoder autograd node.  Parameters are passed as inputs only so autograd knows the output
    depends on them; their gradients are written by the engine directly into the module's gradient
    arena (which `param.grad` views), so backward returns None for them (no per-tensor accumulate copies)."""

    @staticmethod
    def forward(ctx, module, video, need_grad, extra, *params):
        # need_grad is decided by the caller: grad mode is always off inside Function.forward, and ctx.needs_input_grad
        # reflects requires_grad alone (it stays True under torch.no_grad()), so neither tells whether a graph is being built
        ctx.module = module
        out = module._run_forward(video, need_grad, extra)
        ctx.rec = module._rt._cur if need_grad else None      # THIS pass's workspace and input: kept until its backward has run
        return out

    @staticmethod
    def backward(ctx, dlogits):
        rt, rec = ctx.module._rt, ctx.rec
        if rec is None or not rt.pass_is_live(rec):
            raise RuntimeError(
                "neurovit_amd.ViT: backward() of a forward pass whose activations have been overwritten - a pass keeps its workspace "
                "until one whole backward of it has run; a second backward (retain_graph) after another training forward or a train "
                "step of the same module finds it refilled.")
        rt._cur = rec                 # several passes may be pending (siamese / two-forward losses): each runs against its own workspace
        ctx.module._run_backward(dlogits)
        if rt._last is not None and rt._last[2] is rec.ws:
            rt.backward_done = True   # the Grad-CAM taps read the MOST RECENT forward's workspace
        return (None, None, None, None) + (None,) * len(ctx.module._plist)


class ViT(nn.Module):
    """vit_3d.py:77-126, MI355X-native.  forward(video[B, C, F, H, W]) -> [B, num_classes] (fp32)."""

    def __init__(self, *, image_size, image_patch_size, frames, frame_patch_size, num_classes, dim, depth, heads, mlp_dim,
                 pool='cls', channels=3, dim_head=64, dropout=0., emb_dropout=0.):
        super().__init__()
        image_height, image_width = pair(image_size)
        patch_height, patch_width = pair(image_patch_size)

        assert image_height % patch_height == 0 and image_width % patch_width == 0, 'Image dimensions must be divisible by the patch size.'
        assert frames % frame_patch_size == 0, 'Frames must be divisible by frame patch size'

        num_patches = (image_height // patch_height) * (image_width // patch_width) * (frames // frame_patch_size)
        patch_dim = channels * patch_height * patch_width * frame_patch_size

        assert pool in {'cls', 'mean'}, 'pool type must be either cls (cls token) or mean (mean pooling)'

        self.to_patch_embedding = nn.Sequential(
            PatchRearrange(patch_height, patch_width, frame_patch_size),
            nn.LayerNorm(patch_dim),
            nn.Linear(patch_dim, dim),
            nn.LayerNorm(dim),
        )

        self.pos_embedding = nn.Parameter(torch.randn(1, num_patches + 1, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.dropout = nn.Dropout(emb_dropout)

        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout)

        self.pool = pool
        self.to_latent = nn.Identity()

        self.mlp_head = nn.Sequential(
            nn.LayerNorm(dim),
            nn.Linear(dim, num_classes)
        )

        # ---- native engine state (not part of the reference surface) ----
        # heads == 1 with dim_head == dim: the reference drops to_out (nn.Identity, vit_3d.py:32,43-46).  The engine's parameter
        # table always carries the projection, so the arena keeps those slots as CONSTANTS no nn.Parameter views - weight = identity,
        # bias = 0: x + I ao + 0 is exactly x + ao (products with 1.0 and sums with 0.0 are exact), its data gradient g I is
        # exactly g, and the optimizer never sees them (see _build_arena / mark_shadow_fresh)
        self._no_proj = (heads == 1 and dim_head == dim)
        self._phantom = []         # [(offset, fp32 constant)] of those slots
        self._dropout_p = (float(dropout), float(emb_dropout))
        self._cfg = engine.make_config(image_size=image_height, image_patch_size=patch_height, image_width=image_width, patch_width=patch_width, frames=frames,
                                       frame_patch_size=frame_patch_size, num_classes=num_classes, dim=dim, depth=depth,
                                       heads=heads, mlp_dim=mlp_dim, channels=channels, dim_head=dim_head,
                                       pool=pool)
        self._rt = engine.VitRuntime(self._cfg)
        self._arena: Optional[torch.Tensor] = None      # flat fp32 master parameters
        self._shadow: Optional[torch.Tensor] = None     # flat bf16 copy read by the MFMA kernels
        self._grads: Optional[torch.Tensor] = None      # flat fp32 gradients (param.grad are views)
        self._layout = None
        self._plist: List[nn.Parameter] = []
        self._shadow_key = None
        self._last_logits = None   # most recent forward's logits (the Trainer shell reads them without a second forward)
        self._grad_sync = None     # parallel.GradSync: all-reduce gradient buckets while backward still runs
        self._fp8 = None           # enable_fp8(): e4m3 weights + scales for inference forwards
        self.fp8_training = False  # enable_fp8(training=True): training forwards on e4m3 operands too
        self._param_generation = 0 # bumped whenever the fused optimizer rewrites the arena (FusedAdamW.step / step_range)
        # Arithmetic of eval-mode forwards that record no graph: "bf16" (bf16 MFMA operands, the training arithmetic) or "fp32"
        # (every operand fp32 on the fp32 MFMA: what the reference's validate computes, Trainer.py:101-118 - logits within 1e-5
        # of its CPU forward, about 3x the time).  Set directly, through `precision(...)`, or by the config key
        # TRAINING_VIT_EVAL_PRECISION of ViT3DEncoder.
        self.eval_precision = "bf16"
        # 16-bit MFMA operand format: "bf16" (default, BASELINE.json's dtype) or "fp16" - the reference's own training arithmetic
        # (torch.autocast(float16), Trainer.py:68): 11 instead of 8 significand bits at the same MFMA rate, which puts the logits
        # within 1e-3 of the reference's fp32 CPU forward; TrainStep then scales the loss (GradScaler, Trainer.py:29,74-76).  set_operands().
        self.operands = "bf16"
        # inference forwards (eval mode or no dropout, no graph recorded) run their blocks with the LayerNorms folded into the GEMMs around them
        # (engine.VitRuntime.forward_lnfold: 21 of ViT3D-base's 24 LayerNorm launches gone); NEUROVIT_LN_FOLD=0 / fold_layernorm = False: the plain launches
        import os as _os
        self.fold_layernorm = _os.environ.get("NEUROVIT_LN_FOLD", "1") != "0"
        self._fold = None

    def _fresh_fold(self):
        """folded weights of the current parameters (recomputed in place when the stock or the fused optimizer changed them)"""
        key = (self._param_key(), self.operands)
        if self._fold is None or self._fold["key"] != key:
            self._fold = dict(self._rt.lnfold_prepare(self._arena, reuse=self._fold), key=key)
        return self._fold

    def set_operands(self, fmt: str):
        """Switch the operand format of the shadow arena, the activations and the MFMA kernels: "bf16" or "fp16"."""
        if fmt not in _cabi.OPERAND_FORMATS:
            raise ValueError(f"neurovit_amd.ViT: operands must be 'bf16' or 'fp16', got {fmt!r}")
        if fmt != self.operands:
            if fmt == "fp16" and self._fp8 is not None:
                raise RuntimeError("neurovit_amd.ViT: the fp8 path is built beside bf16 operands - disable_fp8() first")
            self.operands = fmt
            self._rt.operands = fmt
            if self._shadow is not None:
                self._shadow = torch.empty(self._shadow.numel(), dtype=self._dtype16(), device=self._shadow.device)
            self._shadow_key = None
        return self

    def _dtype16(self) -> torch.dtype:
        return torch.float16 if self.operands == "fp16" else torch.bfloat16

    # ------------------------------------------------------------------ arena management
    def _build_arena(self):
        """(Re)pack all parameters into one contiguous fp32 arena on their current device and make every
        nn.Parameter a view of it.  Called lazily: after construction, after .to(device), after foreign code
        replaced a parameter's storage."""
        plist = [p for _, p in self.named_parameters()]
        if self._layout is None:
            off, num, total = engine.param_layout(self._cfg)
            self._phantom_slots = []
            if self._no_proj:      # entries 8 + 11 l + {3, 4} of the table are to_out.0.weight / .bias of block l: no module parameter
                d = self._cfg.dim
                drop = {8 + 11 * l + k for l in range(self._cfg.depth) for k in (3, 4)}
                for i in sorted(drop):
                    self._phantom_slots.append((off[i], num[i], (i - 8) % 11 == 3))
                off = [o for i, o in enumerate(off) if i not in drop]
                num = [n for i, n in enumerate(num) if i not in drop]
            assert len(off) == len(plist) and all(p.numel() == n for p, n in zip(plist, num)), \
                "parameter table of the native engine does not match the module tree"
            self._layout = (off, num, total)
        off, num, total = self._layout
        dev = plist[0].device
        arena = torch.zeros(total, dtype=torch.float32, device=dev)
        self._phantom = []
        for o, n, is_weight in self._phantom_slots:
            if is_weight:
                arena[o:o + n].copy_(torch.eye(self._cfg.dim, dtype=torch.float32, device=dev).reshape(-1))
            self._phantom.append((o, arena[o:o + n].clone()))
        grads_alive = self._grads is not None and self._grads.device == dev
        with torch.no_grad():
            for p, o, n in zip(plist, off, num):
                arena[o:o + n].copy_(p.detach().reshape(-1).float())
                p.data = arena[o:o + n].view(p.shape)
        self._arena, self._plist = arena, plist
        self._shadow = torch.empty(total, dtype=self._dtype16(), device=dev)
        self._shadow_key = None
        if not grads_alive:
            self._grads = None

    def _arena_ok(self) -> bool:
        if self._arena is None:
            return False
        off, num, _ = self._layout
        base = self._arena.data_ptr()
        for p, o in zip(self._plist, off):
            if p.data_ptr() != base + 4 * o:
                return False
        return True

    def flat_parameters(self):
        """(arena fp32, shadow bf16) - used by the fused optimizer and the DP gradient all-reduce."""
        if not self._arena_ok():
            self._build_arena()
        return self._arena, self._shadow

    def flat_gradients(self) -> torch.Tensor:
        self.flat_parameters()
        if self._grads is None:
            self._grads = torch.zeros_like(self._arena)
        return self._grads

    def _grad_view(self, i: int) -> torch.Tensor:
        off, num, _ = self._layout
        return self._grads[off[i]:off[i] + num[i]].view(self._plist[i].shape)

    def mark_shadow_fresh(self):
        """Called by the fused AdamW, which writes the arena and the bf16 shadow itself (through raw pointers: no tensor
        `_version` moves, so the generation counter is what tells derived copies - the fp8 weights - that they are stale)."""
        for o, const in self._phantom:           # the fused step ran over the whole arena: put the constant slots back
            self._arena[o:o + const.numel()].copy_(const)
            self._shadow[o:o + const.numel()].copy_(const)
        self._shadow_key = tuple(p._version for p in self._plist)
        self._param_generation += 1

    def _param_key(self):
        return (self._param_generation, tuple(p._version for p in self._plist))

    def _refresh_shadow(self):
        key = tuple(p._version for p in self._plist)
        if key != self._shadow_key:
            _cabi.set_operand_format(self.operands)
            ops.cast_bf16(self._arena.view(1, -1), out=self._shadow.view(1, -1))
            self._shadow_key = key

    # ------------------------------------------------------------------ fp8 inference (BASELINE.json configs[4])
    def enable_fp8(self, calibration_video: torch.Tensor, headroom: float = 2.0, out_proj: bool = True, training: bool = False):
        """Switch inference forwards (no grad being recorded) to the fp8 path: qkv / out-projection / FC1 / FC2 of every block on OCP
        e4m3 MFMA operands (out_proj = False keeps the out-projection, 8 % of the linear FLOPs, on bf16).  `calibration_video` ([B, C, F, H, W] on the device) fixes the per-tensor activation scales; weights are
        re-quantised from the fp32 master parameters whenever they have changed.
        training = True additionally runs TRAINING forwards with qkv / FC1 / FC2 on e4m3 operands (nv_vit_forward_fp8_train; the
        backward pass stays on bf16 operands and reads the bf16 activations the same forward kernels write): the weights are
        re-quantised after every optimizer step (in place), the activation scales stay those of the calibration batch - call
        enable_fp8 again to recalibrate.  Dropout works as in the bf16 forward (same masks).  Default: training forwards keep using bf16."""
        if self.operands != "bf16":
            raise RuntimeError("neurovit_amd.ViT: the fp8 path is built beside bf16 operands - set_operands('bf16') first")
        self.flat_parameters()
        self._refresh_shadow()
        scales = self._rt.calibrate_fp8(calibration_video.float(), self._arena, self._shadow, headroom, out_proj)
        self._fp8 = self._rt.quantize_fp8(self._arena, scales)
        self._fp8["key"] = self._param_key()
        self.fp8_training = bool(training)
        return scales

    def disable_fp8(self):
        self._fp8 = None
        self.fp8_training = False

    def _fresh_fp8(self):
        """the e4m3 weights of the current parameters (re-quantised in place when the stock or the fused optimizer changed them)"""
        if self._fp8["key"] != self._param_key():
            self._fp8 = dict(self._rt.quantize_fp8(self._arena, self._fp8["act_list"], reuse=self._fp8), key=self._param_key())
        return self._fp8

    def precision(self, mode: str):
        """Context manager: eval-mode no-grad forwards inside it run in `mode`: "fp32" (every operand fp32), or "bf16" / "fp16" - both
        name the 16-bit operand path, which runs in this module's operand format (`operands`)."""
        import contextlib
        if mode not in ("bf16", "fp16", "fp32"):
            raise ValueError(f"neurovit_amd.ViT: precision must be 'bf16', 'fp16' or 'fp32', got {mode!r}")

        @contextlib.contextmanager
        def scope():
            before, self.eval_precision = self.eval_precision, mode
            try:
                yield self
            finally:
                self.eval_precision = before
        return scope()

    # ------------------------------------------------------------------ execution
    def draw_dropout(self):
        """(p of the blocks, p of the embedding, seed) of the next forward: (0, 0, 0) in eval mode or without dropout."""
        if self.training and (self._dropout_p[0] > 0 or self._dropout_p[1] > 0):
            # nn.Dropout semantics (vit_3d.py:21,23,39,45,100) with a counter-based mask: a fresh seed per forward from
            # torch's CPU generator (so torch.manual_seed reproduces runs); backward recomputes the same masks.
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            return (self._dropout_p[0], self._dropout_p[1], seed)
        return (0.0, 0.0, 0)

    def _run_forward(self, video, need_grad, extra=(None, 0)):
        vol_sigma, time_points = extra
        drop = self.draw_dropout()
        if self.eval_precision not in ("bf16", "fp16", "fp32"):
            raise ValueError(f"neurovit_amd.ViT: eval_precision must be 'bf16', 'fp16' or 'fp32', got {self.eval_precision!r}")
        if self.eval_precision == "fp32" and not need_grad and not self.training:
            self._last_logits = self._rt.forward_f32(video, self._arena, vol_sigma=vol_sigma, time_points=time_points)
            return self._last_logits
        self._refresh_shadow()
        if self._fp8 is not None and not need_grad and not self.training:
            self._last_logits = self._rt.forward_fp8(video, self._arena, self._shadow, self._fresh_fp8(), vol_sigma=vol_sigma, time_points=time_points)
            return self._last_logits
        if self._fp8 is not None and self.fp8_training and need_grad and not time_points:
            self._last_logits = self._rt.forward_fp8_train(video, self._arena, self._shadow, self._fresh_fp8(), dropout=drop, vol_sigma=vol_sigma)
            return self._last_logits
        if self.fold_layernorm and not need_grad and drop[0] == 0.0 and drop[1] == 0.0:
            self._last_logits = self._rt.forward_lnfold(video, self._arena, self._shadow, self._fresh_fold(), vol_sigma=vol_sigma, time_points=time_points)
            return self._last_logits
        self._last_logits = self._rt.forward(video, self._arena, self._shadow, training=need_grad, dropout=drop, vol_sigma=vol_sigma,
                                             time_points=time_points)
        return self._last_logits

    def _run_backward(self, dlogits):
        grads = self.flat_gradients()
        trainable = [i for i, p in enumerate(self._plist) if p.requires_grad]
        state = [self._plist[i].grad for i in trainable]
        if all(g is None for g in state):
            self._backward_into(dlogits, grads, accumulate=False)
            for i in trainable:
                self._plist[i].grad = self._grad_view(i)
        elif all(g is not None and g.data_ptr() == self._grad_view(i).data_ptr() for g, i in zip(state, trainable)):
            self._backward_into(dlogits, grads, accumulate=True)
        else:   # foreign .grad tensors: compute into a scratch arena and add
            scratch = torch.empty_like(grads)
            self._rt.backward(dlogits, self._arena, self._shadow, scratch, accumulate=False)
            off, num, _ = self._layout
            for i in trainable:
                g = scratch[off[i]:off[i] + num[i]].view(self._plist[i].shape)
                p = self._plist[i]
                p.grad = g.clone() if p.grad is None else p.grad.add_(g)

    def mirrored_ranges(self):
        """Arena element ranges whose gradients nv_vit_backward_stages16 also writes, rounded to bf16, into `grads16`: the weights of
        the Linear layers (to_qkv, to_out, FC1, FC2 of every block; the patch embedding's when patch_dim % 8 == 0) - 99.4 % of
        ViT3D-base's gradient bytes.  Sorted, disjoint."""
        if getattr(self, "_mirrored", None) is not None:
            return self._mirrored
        off, num, _ = self._layout
        P = self._cfg.channels * self._cfg.image_patch_size * (self._cfg.patch_width or self._cfg.image_patch_size) * self._cfg.frame_patch_size
        out = []
        for (name, _), o, n in zip(self.named_parameters(), off, num):
            if name.startswith("transformer.layers.") and name.endswith((".to_qkv.weight", ".to_out.0.weight", ".net.1.weight", ".net.4.weight")):
                out.append((o, o + n))
            elif name == "to_patch_embedding.2.weight" and P % 8 == 0:
                out.append((o, o + n))
        self._mirrored = sorted(out)          # the layout never changes for a constructed module
        return self._mirrored

    def _backward_into(self, dlogits, grads, accumulate):
        sync = self._grad_sync
        if sync is None:
            self._rt.backward(dlogits, self._arena, self._shadow, grads, accumulate=accumulate)
            return
        from .parallel import bucket_stages
        sync.begin()
        last_stage = self._cfg.depth + 1
        # bf16 messages: the weight-gradient GEMMs write their share of the message buffer themselves (no cast pass over it)
        msg = sync.message_buffer(grads) if (grads.is_cuda and sync.world > 1) else None
        sync.mirrored = self.mirrored_ranges() if msg is not None else None
        plan = getattr(self, "_bucket_plan", None)
        if plan is None or plan[0] != sync.n_buckets:    # stage groups and their arena ranges: fixed per module, computed once
            groups = list(bucket_stages(self._cfg.depth + 2, sync.n_buckets))
            plan = self._bucket_plan = (sync.n_buckets, [(f, l) + tuple(self._rt.stage_range(f, l)) for f, l in groups])
        for first, last, begin, end in plan[1]:
            # intermediate buckets do not stall the main stream on the auxiliary (weight-gradient) stream: the bucket's
            # all-reduce is ordered after both streams instead
            self._rt.backward(dlogits, self._arena, self._shadow, grads, accumulate=accumulate, stages=(first, last),
                              join_aux=(last == last_stage), grads16=msg)
            sync.bucket_ready(grads, begin, end, also_after=None if last == last_stage else self._rt.aux_stream_object(grads.device))
        sync.finish()

    def check_video(self, video, time_points=0, arena_checked=False):
        """Device, extents and arena placement of an input batch (raises as the reference's einops / pos_embedding add would)."""
        if not video.is_cuda:
            raise RuntimeError("neurovit_amd.ViT: input must live on the MI355X (cuda) device - there is no CPU fallback")
        c = self._cfg
        width = c.image_width or c.image_size
        if time_points:
            if video.dim() != 5 or tuple(video.shape[1:]) != (c.image_size, width, c.frames, time_points):
                raise ValueError(f"neurovit_amd.ViT: expected a 4D batch [B, {c.image_size}, {width}, {c.frames}, {time_points}], got {tuple(video.shape)}")
        elif video.dim() != 5 or tuple(video.shape[1:]) != (c.channels, c.frames, c.image_size, width):
            # the reference fails here too (einops Rearrange / the pos_embedding add, vit_3d.py:92,118); the gather kernel
            # takes its extents from the config, so a wrong-sized volume must never reach it
            raise ValueError(f"neurovit_amd.ViT: expected video [B, {c.channels}, {c.frames}, {c.image_size}, {width}] "
                             f"(channels, frames, height, width), got {tuple(video.shape)}")
        if not arena_checked and not self._arena_ok():
            self._build_arena()
        if self._arena.device != video.device:
            raise RuntimeError(f"neurovit_amd.ViT: parameters on {self._arena.device}, input on {video.device}")

    def forward(self, video, vol_sigma=None, time_points=0):
        """video [B, C, F, H, W] -> [B, num_classes] (vit_3d.py:112-126).  Beyond the reference (both optional):
        vol_sigma [B] marks `video` as RAW volumes whose per-volume z-score (std + 1e-8) is folded into the patch LayerNorm;
        time_points = T > 0 takes a contiguous 4D batch [B, H, W, D, T] and encodes its B*T volumes without the regroup copy."""
        self.check_video(video, time_points)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._plist)
        if time_points and need_grad:
            raise NotImplementedError("neurovit_amd.ViT: the fused 4D input form is forward-only (frozen encoder of the 4D model)")
        return _ViTFunction.apply(self, video.float(), need_grad, (vol_sigma, int(time_points)), *self._plist)

    # activations / gradients of the last block's attention LayerNorm output (Grad-CAM contract, NeuroEncoder.py:70-82)
    def last_attn_norm_output_raw(self) -> torch.Tensor:
        """[B, n, d] view into the workspace of the most recent forward (no copy): the operand format, or fp32 after an fp32 inference forward."""
        B = self._rt._last[0]
        n, d = self.pos_embedding.shape[1], self.pos_embedding.shape[2]
        return self._rt.tap("xn1", self._cfg.depth - 1, (B, n, d), torch.float32 if self._rt._last[1] == 2 else self._dtype16())

    def last_attn_norm_grad_raw(self) -> torch.Tensor:
        """fp32 [B, n, d] view into the workspace; valid once a backward of the most recent training forward has run."""
        if not self._rt.backward_done:
            raise RuntimeError("neurovit_amd.ViT: no backward pass has run for the most recent forward - the hook gradient is not available")
        B = self._rt._last[0]
        n, d = self.pos_embedding.shape[1], self.pos_embedding.shape[2]
        return self._rt.tap("hookg", -1, (B, n, d), torch.float32)

    def last_attn_norm_output(self) -> torch.Tensor:
        return self.last_attn_norm_output_raw().float()

    def last_attn_norm_grad(self) -> torch.Tensor:
        return self.last_attn_norm_grad_raw().clone()
