"""Fused AdamW for the flat parameter arenas (Trainer.py:31,75).

`FusedAdamW(model.parameters(), lr=..., weight_decay=...)` has the constructor of torch.optim.AdamW
(defaults betas (0.9, 0.999), eps 1e-8, decoupled weight decay on every parameter).  Parameters that
are views of a ViT arena are updated by ONE nv_adamw_step launch per arena (which also refreshes the
bf16 shadow the MFMA kernels read); the 4D model's temporal head is a second, 10 k-parameter arena
(temporal.TemporalHead, fp32 only) stepped the same way; any other parameter is delegated to torch.optim.AdamW.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from . import _cabi, ops
from .vit_3d import ViT


class LossScaler:
    """torch.amp.GradScaler (src/Trainer.py:29,74-76: scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()) for
    training on fp16 operands, kept on the DEVICE: the scale, the inf / NaN flag, the growth tracker and the count of applied
    optimizer updates live in one 16-float block (csrc/optim.hip, nv_loss_scale_*), and the fused AdamW reads that block - so a
    step costs no host synchronisation, where the reference's scaler.step() reads found_inf back every step.  Same policy and
    defaults as GradScaler: the loss gradient is multiplied by `scale`; any non-finite gradient skips the update and multiplies the
    scale by backoff_factor; growth_interval clean steps in a row multiply it by growth_factor."""

    def __init__(self, device, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000,
                 start_step: int = 0):
        self.state = torch.zeros(16, dtype=torch.float32, device=device)
        ops.loss_scale_init(self.state, init_scale, growth_factor, backoff_factor, growth_interval, start_step)

    def scale(self, loss: torch.Tensor) -> torch.Tensor:
        """loss * current scale, as a device-side product (GradScaler.scale): backward() of the result carries the scale."""
        return loss * self.state[0]

    def check(self, grads: torch.Tensor) -> None:
        ops.loss_scale_check(grads, self.state)

    def update(self, lr: float, betas) -> None:
        """Decide skip-or-step for the gradients checked since the last update, adjust the scale, prepare AdamW's constants."""
        ops.loss_scale_update(self.state, lr, betas)

    # host-side views (each synchronises: logging / tests only)
    def get_scale(self) -> float:
        return float(self.state[0])

    def steps_applied(self) -> int:
        return int(self.state[5])

    def steps_skipped(self) -> int:
        return int(self.state[11])

    def last_step_skipped(self) -> bool:
        return bool(self.state[3] != 0)


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, *, model: torch.nn.Module = None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._model = model
        self._arenas: List[object] = []      # ViT modules and TemporalHead objects: flat_parameters() / flat_gradients() / mark_shadow_fresh()
        self._state_mv: Dict[int, tuple] = {}
        self._rest = None
        self._bound = False
        self._steps = 0

    def bind(self, model: torch.nn.Module):
        """Tell the optimizer which module tree owns the parameters (needed to find the arenas)."""
        self._model = model
        self._bound = False
        return self

    def _bind(self):
        mine = {id(p) for g in self.param_groups for p in g["params"]}
        claimed = set()
        self._arenas = []
        if self._model is not None:
            for m in self._model.modules():
                if isinstance(m, ViT):
                    m.flat_parameters()
                    ids = [id(p) for p in m._plist]
                    if all(i in mine for i in ids) and all(p.requires_grad for p in m._plist):
                        self._arenas.append(m)
                        claimed.update(ids)
            for m in self._model.modules():             # the 4D model's temporal head: a second, small arena (temporal.TemporalHead)
                th = getattr(m, "_temporal_head", None)
                if th is not None:
                    th.flat_parameters()
                    ids = [id(p) for p in th._plist]
                    if all(i in mine for i in ids) and all(p.requires_grad for p in th._plist):
                        self._arenas.append(th)
                        claimed.update(ids)
        rest = [p for g in self.param_groups for p in g["params"] if id(p) not in claimed and p.requires_grad]
        g0 = self.param_groups[0]
        self._rest = torch.optim.AdamW(rest, lr=g0["lr"], betas=g0["betas"], eps=g0["eps"], weight_decay=g0["weight_decay"]) if rest else None
        self._bound = True

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0, reduced_bf16=None, scaler: "LossScaler" = None):
        """reduced_bf16: {id(vit): flat bf16 gradient buffer} - when the data-parallel all-reduce ran on bf16 messages the
        optimizer reads the reduced gradients straight from that buffer (no cast back into the fp32 gradient arena).
        scaler: a LossScaler whose scale the gradients carry (GradScaler.step + GradScaler.update, Trainer.py:75-76): every arena's
        gradients are checked for inf / NaN, then the update is applied - un-scaled - or skipped, all on the device."""
        if not self._bound:
            self._bind()
        self._steps += 1
        g0 = self.param_groups[0]
        live = [h for h in self._arenas if not self._no_gradients(h)]
        if scaler is not None:
            for holder in live:
                if hasattr(holder, "gather_foreign_grads"):
                    holder.gather_foreign_grads()
                red = None if reduced_bf16 is None else reduced_bf16.get(id(holder))
                scaler.check(holder.flat_gradients() if red is None else red.float())
            if self._rest is not None:
                for g in self._rest.param_groups:
                    for p in g["params"]:
                        if p.grad is not None:
                            scaler.check(p.grad.contiguous().float())
            scaler.update(g0["lr"], g0["betas"])
        for holder in live:
            self._step_arena(holder, grad_scale, None if reduced_bf16 is None else reduced_bf16.get(id(holder)), scaler)
        if self._rest is not None:
            if scaler is not None:
                if scaler.last_step_skipped():          # (stock parameters beside a scaled arena: the one place that reads the flag back)
                    return None
                for g in self._rest.param_groups:
                    for p in g["params"]:
                        if p.grad is not None:
                            p.grad.mul_(scaler.state[1])
            for g in self._rest.param_groups:
                g["lr"] = g0["lr"]
            self._rest.step()
        return None

    def step_range(self, vit: ViT, begin: int, end: int, grad_scale: float = 1.0, max_blocks: int = 0):
        """AdamW on arena elements [begin, end) only (DP: run per gradient bucket behind its all-reduce).
        The caller advances the step counter once per optimizer step with `begin_step()`."""
        arena, shadow = vit.flat_parameters()
        grads = vit.flat_gradients()
        key = id(vit)
        if key not in self._state_mv:
            self._state_mv[key] = (torch.zeros_like(arena), torch.zeros_like(arena))
        m, v = self._state_mv[key]
        g0 = self.param_groups[0]
        _cabi.set_operand_format(getattr(vit, "operands", "bf16"))
        ops.adamw_step(arena[begin:end], grads[begin:end], m[begin:end], v[begin:end], shadow[begin:end], self._steps, g0["lr"],
                       g0["betas"], g0["eps"], g0["weight_decay"], grad_scale, max_blocks)
        vit._param_generation += 1             # the arena changed under derived copies (fp8 weights re-quantise on their next use)

    def begin_step(self):
        if not self._bound:
            self._bind()
        self._steps += 1

    def arenas(self):
        """The flat arenas (ViT modules, TemporalHead objects) this optimizer steps with one fused launch each."""
        if not self._bound:
            self._bind()
        return list(self._arenas)

    def arena_state(self, holder):
        """(exp_avg, exp_avg_sq) arenas of `holder` (a ViT or a TemporalHead), created on first use."""
        arena, _ = holder.flat_parameters()
        key = id(holder)
        if key not in self._state_mv or self._state_mv[key][0].data_ptr() == 0 or self._state_mv[key][0].device != arena.device:
            self._state_mv[key] = (torch.zeros_like(arena), torch.zeros_like(arena))
        return self._state_mv[key]

    @staticmethod
    def _no_gradients(holder) -> bool:
        """torch.optim.AdamW skips a parameter whose .grad is None (no weight decay, no moment decay, no step count): an arena none of
        whose parameters has a gradient - zero_grad(set_to_none=True) followed by step(), or a head no backward reached - is skipped
        as a whole.  (A PARTIALLY populated arena is stepped with zeros for the missing gradients: see TemporalHead.gather_foreign_grads.)"""
        return all(p.grad is None for p in holder._plist)

    def _step_arena(self, holder, grad_scale, reduced=None, scaler=None):
        arena, shadow = holder.flat_parameters()
        grads = holder.flat_gradients()
        if hasattr(holder, "gather_foreign_grads"):
            holder.gather_foreign_grads()
        m, v = self.arena_state(holder)
        g0 = self.param_groups[0]
        _cabi.set_operand_format(getattr(holder, "operands", "bf16"))      # the format of the shadow this launch rewrites
        ops.adamw_step(arena, grads if reduced is None else reduced, m, v, shadow, self._steps, g0["lr"], g0["betas"], g0["eps"], g0["weight_decay"], grad_scale,
                       scale_state=None if scaler is None else scaler.state)
        holder.mark_shadow_fresh()

    @torch.no_grad()
    def step_rest(self, grad_scale: float = 1.0):
        """Parameters outside the ViT arenas (after those ranges were stepped bucket by bucket): the temporal head's arena with
        `grad_scale`, stock parameters as they are."""
        for holder in self._arenas:
            if not isinstance(holder, ViT) and not self._no_gradients(holder):
                self._step_arena(holder, grad_scale)
        if self._rest is not None:
            g0 = self.param_groups[0]
            for g in self._rest.param_groups:
                g["lr"] = g0["lr"]
            self._rest.step()
