"""The reference's train step (src/Trainer.py:65-79) on the MI355X-native path, data-parallel capable.

    outputs = model(fMRI); loss = CrossEntropyLoss(outputs, labels)
    optimizer.zero_grad(set_to_none=True); loss.backward(); optimizer.step()

`TrainStep` runs exactly that sequence with the gfx950 kernels: engine forward, fused CE, staged engine backward
(gradient buckets all-reduced over RCCL while later stages still run), fused AdamW (+ bf16 shadow refresh).
bf16 MFMA operands with fp32 master weights need no GradScaler (Trainer.py:29,74-76 exist for fp16 autocast).
No host synchronisation happens inside a step; `loss` is returned as a device tensor.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from .NeuroEncoder import NeuroEncoder
from .nn import CrossEntropyLoss
from .optim import FusedAdamW
from .parallel import GradSync, broadcast_parameters


class TrainStep:
    def __init__(self, model: NeuroEncoder, lr: Optional[float] = None, weight_decay: Optional[float] = None, process_group=None,
                 n_buckets: int = 4, accumulation_steps: int = 1):
        cfg = model.config
        self.model = model
        self.criterion = CrossEntropyLoss()
        lr = cfg.get("TRAINING_LEARNING_RATE", 1e-4) if lr is None else lr
        wd = cfg.get("TRAINING_WEIGHT_DECAY", 1e-2) if weight_decay is None else weight_decay
        self.optimizer = FusedAdamW(model.parameters(), lr=lr, weight_decay=wd, model=model)
        self.accumulation_steps = max(1, int(accumulation_steps))
        self._micro = 0
        self.sync = GradSync(process_group, n_buckets) if dist.is_initialized() else None
        vit = model.volume_encoder.vit3d
        if self.sync is not None and self.sync.world > 1:
            arena, _ = vit.flat_parameters()
            broadcast_parameters(arena, process_group)
            for p in model.parameters():               # parameters outside the arena (4D temporal head)
                if not any(p is q for q in vit._plist):
                    dist.broadcast(p.data, src=0, group=process_group)
            vit._shadow_key = None
        vit._grad_sync = None

    def __call__(self, fmri: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        model, vit = self.model, self.model.volume_encoder.vit3d
        last_micro = (self._micro + 1) % self.accumulation_steps == 0
        # the all-reduce runs only on the micro-step that is followed by the optimizer step
        vit._grad_sync = self.sync if (self.sync is not None and self.sync.world > 1 and last_micro) else None
        outputs = model(fmri)
        loss = self.criterion(outputs, labels)
        if self._micro == 0:
            self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self._micro += 1
        if last_micro:
            scale = (self.sync.grad_scale if self.sync is not None else 1.0)
            if self.sync is not None and self.sync.world > 1:
                for p in model.parameters():           # stragglers outside the arena: tiny, reduce inline
                    if p.grad is not None and not any(p is q for q in vit._plist):
                        dist.all_reduce(p.grad, group=self.sync.pg)
                        p.grad.mul_(scale)
            self.optimizer.step(grad_scale=scale)
            self._micro = 0
        return loss.detach()
