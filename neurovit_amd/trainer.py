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
                 n_buckets: int = 4, accumulation_steps: int = 1, overlap_optimizer: bool = False):
        cfg = model.config
        self.model = model
        self.criterion = CrossEntropyLoss()
        lr = cfg.get("TRAINING_LEARNING_RATE", 1e-4) if lr is None else lr
        wd = cfg.get("TRAINING_WEIGHT_DECAY", 1e-2) if weight_decay is None else weight_decay
        self.optimizer = FusedAdamW(model.parameters(), lr=lr, weight_decay=wd, model=model)
        self.accumulation_steps = max(1, int(accumulation_steps))
        self._micro = 0
        vit = model.volume_encoder.vit3d
        self._vit = vit
        self._arena_trainable = all(p.requires_grad for p in vit.parameters())
        world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.world = world
        # Bucket pipeline: as soon as a group of backward stages has produced its (contiguous) gradient range, a side
        # stream all-reduces it (world > 1) while the main stream continues the backward pass.  overlap_optimizer=True
        # additionally runs the fused AdamW of that range on the side stream; measured on one MI355X this LOSES 5 %
        # (740 vs 777 volumes/s: the HBM-bound optimizer slows the concurrent GEMMs more than it hides), so it is off.
        self.sync = None
        self._overlap_opt = bool(overlap_optimizer)
        if self._arena_trainable and (world > 1 or overlap_optimizer):
            self.sync = GradSync(process_group, n_buckets, after_bucket=self._bucket_update if overlap_optimizer else None)
        if world > 1:
            arena, _ = vit.flat_parameters()
            broadcast_parameters(arena, process_group)
            for p in model.parameters():               # parameters outside the arena (4D temporal head)
                if not any(p is q for q in vit._plist):
                    dist.broadcast(p.data, src=0, group=process_group)
            vit._shadow_key = None
        self._pg = process_group
        vit._grad_sync = None

    def _bucket_update(self, begin: int, end: int):
        self.optimizer.step_range(self._vit, begin, end, grad_scale=1.0 / self.world)

    def __call__(self, fmri: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        model, vit = self.model, self._vit
        last_micro = (self._micro + 1) % self.accumulation_steps == 0
        pipelined = self.sync is not None and last_micro
        # the all-reduce / optimizer pipeline runs only on the micro-step that ends an accumulation window
        vit._grad_sync = self.sync if pipelined else None
        if pipelined and self._overlap_opt:
            self.optimizer.begin_step()
        outputs = model(fmri)
        loss = self.criterion(outputs, labels)
        if self._micro == 0:
            self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self._micro += 1
        if last_micro:
            scale = 1.0 / self.world
            if self.world > 1:
                for p in model.parameters():           # stragglers outside the arena: tiny, reduce inline
                    if p.grad is not None and not any(p is q for q in vit._plist):
                        dist.all_reduce(p.grad, group=self._pg)
                        p.grad.mul_(scale)
            if pipelined and self._overlap_opt:
                vit.mark_shadow_fresh()                # every range was updated (and its bf16 shadow refreshed) by the buckets
                self.optimizer.step_rest()
            else:
                self.optimizer.step(grad_scale=scale)
            self._micro = 0
        return loss.detach()
