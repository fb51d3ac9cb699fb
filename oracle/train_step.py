"""CPU restatement of the reference train step.  TEST INFRASTRUCTURE ONLY.

Restates reference src/Trainer.py:65-79 (forward -> nn.CrossEntropyLoss (mean) ->
zero_grad -> backward -> AdamW step) with the optimizer of Trainer.py:31
(torch.optim.AdamW defaults: betas (0.9, 0.999), eps 1e-8, decoupled weight decay
applied to EVERY parameter - single param group).  fp32, no GradScaler (loss
scaling is an exact power-of-two round trip in fp32).

Pinned against goldens made by composing the imported reference model with stock
nn.CrossEntropyLoss + torch.optim.AdamW (tests/golden/make_golden.py).
"""
from __future__ import annotations

from typing import Dict

import torch

from . import ref_cpu


def cross_entropy(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """nn.CrossEntropyLoss() defaults: mean over batch of -log_softmax[target]  (Trainer.py:30,70)."""
    lse = torch.logsumexp(logits, dim=1)
    picked = logits.gather(1, target.view(-1, 1)).squeeze(1)
    return (lse - picked).mean()


class AdamW:
    """torch.optim.AdamW restated (single-tensor form, amsgrad=False, maximize=False)."""

    def __init__(self, params: Dict[str, torch.Tensor], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.params = params
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.t = 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    @torch.no_grad()
    def step(self, grads: Dict[str, torch.Tensor]):
        self.t += 1
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** self.t
        bc2_sqrt = (1.0 - b2 ** self.t) ** 0.5
        for k, p in self.params.items():
            g = grads.get(k)
            if g is None:
                continue
            p.mul_(1.0 - self.lr * self.wd)
            self.m[k].lerp_(g, 1.0 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1.0 - b2)
            denom = (self.v[k].sqrt() / bc2_sqrt).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-(self.lr / bc1))


def train_step(sd: Dict[str, torch.Tensor], cfg: ref_cpu.ViTCfg, opt: AdamW, video: torch.Tensor,
               target: torch.Tensor, emulate_bf16: bool = False, fp8_scales=None):
    """One Trainer.py:65-79 iteration on a ViT-level state dict.  Returns (loss, logits, grads).
    fp8_scales: the forward restates the HIP path's fp8 training forward (e4m3 LayerNorm outputs / GELU output / per-row weights for
    qkv, FC1, FC2; quantisation is a cast, i.e. a straight-through estimator for autograd)."""
    leaves = {k: v.detach().requires_grad_(True) for k, v in sd.items()}
    logits = ref_cpu.vit_forward(leaves, cfg, video, emulate_bf16, fp8_scales=fp8_scales)
    loss = cross_entropy(logits, target)
    gl = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    grads = {k: g for k, g in zip(leaves.keys(), gl) if g is not None}
    opt.step(grads)
    return loss.detach(), logits.detach(), grads
