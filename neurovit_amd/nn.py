"""Loss module of the train step (Trainer.py:30,70): nn.CrossEntropyLoss() on the gfx950 path."""
from __future__ import annotations

import torch

from . import ops


class _CEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        loss, dlogits = ops.ce_loss(logits.contiguous().float(), target.contiguous().long(), 1.0, want_grad=logits.requires_grad)
        ctx.save_for_backward(dlogits)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None


class CrossEntropyLoss(torch.nn.Module):
    """Drop-in for nn.CrossEntropyLoss() with default arguments (mean reduction, class-index targets):
    forward and d(loss)/d(logits) come from one nv_ce_loss launch."""

    def forward(self, logits, target):
        if not logits.is_cuda:
            raise RuntimeError("neurovit_amd.nn.CrossEntropyLoss runs on MI355X only")
        return _CEFunction.apply(logits, target)
