"""Native temporal head of the 4D model (src/models/NeuroEncoder.py:60-66, 207-230).

The reference's `TemporalTransformer` (one nn.TransformerEncoderLayer, d_model 2, nhead 2) and `ProjectionHead` (nn.Linear(2, 2))
stay what they are in the module tree - same classes, same state_dict keys, same initialisation - but their 16 parameters are
packed into one flat fp32 arena (every nn.Parameter a view of it, gradients views of a second arena) and

    per_volume [B, T, 2] -> encoder layer -> mean over time -> projection -> [B, 2]

runs as ONE launch forward and ONE backward (csrc/temporal.hip) instead of ~60 stock launches per train micro-step; the fused
AdamW steps the arena in one more.  `TemporalHead` is a plain object hung on the NeuroEncoder (not an nn.Module: it owns no state).
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from . import ops

MAX_TIMEPOINTS = 64      # csrc/temporal.hip: TH_MAXT
MAX_FEEDFORWARD = 2048   # TH_THREADS * TH_KPT


class _TemporalHeadFunction(torch.autograd.Function):
    """Parameters are inputs only so that autograd sees the dependency; their gradients go straight into the gradient arena
    (which `param.grad` views), like the encoder's node (vit_3d._ViTFunction)."""

    @staticmethod
    def forward(ctx, head, x, *params):
        ctx.head = head
        ctx.drop = head._draw_dropout()
        ctx.save_for_backward(x)
        return ops.temporal_head_fwd(x, head._arena, head.ff, head.eps, ctx.drop[1], ctx.drop[0])

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        dx = ctx.head._run_backward(x, dout.contiguous().float(), ctx.drop, ctx.needs_input_grad[1])
        return (None, dx) + (None,) * len(ctx.head._plist)


class TemporalHead:
    def __init__(self, temporal_transformer: nn.Module, projection_head: nn.Module):
        self.temporal_transformer, self.projection_head = temporal_transformer, projection_head
        self._layer = temporal_transformer.transformer.layers[0]
        self._arena: Optional[torch.Tensor] = None
        self._grads: Optional[torch.Tensor] = None
        self._plist: List[nn.Parameter] = []
        self._param_generation = 0
        lay = self._layer
        self.ff = lay.linear1.out_features
        self.eps = float(lay.norm1.eps)

    # ------------------------------------------------------------------ what the kernel implements
    def supported(self, x: torch.Tensor) -> bool:
        lay, mha, proj = self._layer, self._layer.self_attn, self.projection_head.projection_head
        drops = {float(lay.dropout.p), float(lay.dropout1.p), float(lay.dropout2.p), float(mha.dropout)}
        return (x.is_cuda and x.dim() == 3 and x.shape[2] == 2 and 1 <= x.shape[1] <= MAX_TIMEPOINTS and self.ff <= MAX_FEEDFORWARD
                and self.ff % 4 == 0 and len(self.temporal_transformer.transformer.layers) == 1
                and self.temporal_transformer.transformer.norm is None
                and mha.embed_dim == 2 and mha.num_heads == 2 and mha._qkv_same_embed_dim and mha.in_proj_bias is not None
                and mha.bias_k is None and not mha.add_zero_attn and not lay.norm_first
                and (lay.activation is torch.nn.functional.relu or isinstance(lay.activation, nn.ReLU))
                and float(lay.norm2.eps) == self.eps and len(drops) == 1
                and proj.in_features == 2 and proj.out_features == 2 and proj.bias is not None)

    # ------------------------------------------------------------------ arena (same scheme as ViT._build_arena)
    def _params(self):
        return [p for _, p in self.temporal_transformer.named_parameters()] + [p for _, p in self.projection_head.named_parameters()]

    def _build_arena(self):
        plist = self._params()
        F = self.ff
        sizes = [12, 6, 4, 2, 2 * F, F, 2 * F, 2, 2, 2, 2, 2, 4, 2]
        assert [p.numel() for p in plist] == sizes, "temporal head: parameter list does not match csrc/temporal.hip's arena layout"
        total = ops.temporal_head_param_count(F)
        assert total == sum(sizes)
        dev = plist[0].device
        arena = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        self._offsets = []
        with torch.no_grad():
            for p, n in zip(plist, sizes):
                arena[off:off + n].copy_(p.detach().reshape(-1).float())
                p.data = arena[off:off + n].view(p.shape)
                self._offsets.append(off)
                off += n
        if self._grads is not None and self._grads.device != dev:
            self._grads = None
        self._arena, self._plist = arena, plist
        self._param_generation += 1

    def _arena_ok(self) -> bool:
        if self._arena is None:
            return False
        base = self._arena.data_ptr()
        cur = self._params()
        return len(cur) == len(self._plist) and all(p is q and p.data_ptr() == base + 4 * o for p, q, o in zip(cur, self._plist, self._offsets))

    def flat_parameters(self):
        """(arena fp32, None): the fused optimizer's interface (no bf16 shadow - the head computes in fp32)."""
        if not self._arena_ok():
            self._build_arena()
        return self._arena, None

    def flat_gradients(self) -> torch.Tensor:
        self.flat_parameters()
        if self._grads is None:
            self._grads = torch.zeros_like(self._arena)
        return self._grads

    def _grad_view(self, i: int) -> torch.Tensor:
        o, p = self._offsets[i], self._plist[i]
        return self._grads[o:o + p.numel()].view(p.shape)

    def mark_shadow_fresh(self):
        self._param_generation += 1

    def gather_foreign_grads(self):
        """Before a fused optimizer step: gradients that autograd produced outside the arena (the stock-module path of
        NeuroEncoder.forward) are moved into it; a parameter without a gradient contributes zeros - the fused step runs over the
        whole arena, so such a parameter still sees weight decay and moment decay where torch.optim.AdamW would skip it.  The
        optimizer skips the arena altogether when NO parameter has a gradient (FusedAdamW._no_gradients: the case torch skips
        entirely); a partially used head does not occur on the NeuroEncoder path (every parameter is on the forward path)."""
        grads = self.flat_gradients()
        for i, p in enumerate(self._plist):
            view = self._grad_view(i)
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view
        return grads

    # ------------------------------------------------------------------ forward / backward
    def _draw_dropout(self):
        p = float(self._layer.dropout.p)
        if not self.temporal_transformer.training or p <= 0.0:
            return (0.0, 0)
        return (p, int(torch.randint(0, 2 ** 62, (1,)).item()))        # a fresh seed per forward from torch's CPU generator

    def __call__(self, per_volume: torch.Tensor) -> torch.Tensor:
        x = per_volume.contiguous().float()
        self.flat_parameters()
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self._plist)):
            return _TemporalHeadFunction.apply(self, x, *self._plist)
        p, seed = self._draw_dropout()
        return ops.temporal_head_fwd(x, self._arena, self.ff, self.eps, seed, p)

    def _run_backward(self, x, dout, drop, want_dx):
        grads = self.flat_gradients()
        trainable = [i for i, p in enumerate(self._plist) if p.requires_grad]
        state = [self._plist[i].grad for i in trainable]
        kw = dict(eps=self.eps, drop_seed=drop[1], drop_p=drop[0], want_dx=want_dx)
        if len(trainable) == len(self._plist) and all(g is None for g in state):
            dx = ops.temporal_head_bwd(x, self._arena, self.ff, dout, grads, accumulate=False, **kw)
            for i in trainable:
                self._plist[i].grad = self._grad_view(i)
        elif len(trainable) == len(self._plist) and all(g is not None and g.data_ptr() == self._grad_view(i).data_ptr() for g, i in zip(state, trainable)):
            dx = ops.temporal_head_bwd(x, self._arena, self.ff, dout, grads, accumulate=True, **kw)
        else:   # foreign .grad tensors or a partially frozen head: compute into a scratch arena and add
            scratch = torch.empty_like(grads)
            dx = ops.temporal_head_bwd(x, self._arena, self.ff, dout, scratch, accumulate=False, **kw)
            for i in trainable:
                o, p = self._offsets[i], self._plist[i]
                g = scratch[o:o + p.numel()].view(p.shape)
                p.grad = g.clone() if p.grad is None else p.grad.add_(g)
        return dx
