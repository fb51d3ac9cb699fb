"""The reference's train step (src/Trainer.py:65-79) on the MI355X-native path, data-parallel capable.

    outputs = model(fMRI); loss = CrossEntropyLoss(outputs, labels)
    optimizer.zero_grad(set_to_none=True); loss.backward(); optimizer.step()

`TrainStep` runs exactly that sequence with the gfx950 kernels: engine forward, fused CE, staged engine backward
(gradient buckets all-reduced over RCCL while later stages still run), fused AdamW (+ 16-bit shadow refresh).
bf16 MFMA operands with fp32 master weights need no GradScaler; a model on fp16 operands (ViT.set_operands("fp16"), the
reference's own autocast arithmetic) gets the reference's GradScaler (Trainer.py:29,74-76) as optim.LossScaler, kept on the device.
No host synchronisation happens inside a step; `loss` is returned as a device tensor.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist

from .NeuroEncoder import NeuroEncoder
from .nn import CrossEntropyLoss
from .optim import FusedAdamW, LossScaler
from .parallel import GradSync, NativeComm, broadcast_parameters


class TrainStep:
    FUSE_MAX_ROWS = 4096      # fuse_update=None: token rows (batch x tokens) up to which the layers' weights are updated during the backward pass

    def __init__(self, model: NeuroEncoder, lr: Optional[float] = None, weight_decay: Optional[float] = None, process_group=None,
                 n_buckets: int = 4, accumulation_steps: int = 1, overlap_optimizer: bool = False,
                 grad_comm_dtype: torch.dtype = torch.float32, grad_comm_algo: Optional[str] = None, fuse_update: Optional[int] = None,
                 loss_scale=None, native_dp: Optional[bool] = None):
        cfg = model.config
        # native_dp: world > 1 - the step stays ONE native call (nv_vit_train_step with an nv_dp_plan: RCCL all-reduce per bucket issued
        # from native code on a communicator of the library's own, AdamW behind it) instead of the Python-driven staged backward;
        # None = on unless NEUROVIT_NATIVE_DP=0 (falls back, loudly, when the communicator cannot be built: gloo groups, CPU tensors)
        # loss_scale: None = by operand format - "dynamic" for fp16 operands (torch.amp.GradScaler semantics, on the device:
        # optim.LossScaler), none for bf16; "dynamic"; or a number = a static scale (no overflow check, no skipped steps - keeps
        # the optimizer update inside the backward pass, fuse_update).  A power of two changes no bit of a finite result.
        # native step only - where AdamW runs for the layers' Linear weights (96 % of the parameters); same bits in every mode:
        #   0  with the rest of the arena, one launch behind the backward pass
        #   3  per layer on the auxiliary stream, behind that layer's weight-gradient GEMMs, beside the main stream's chain
        #   1  inside those GEMMs' epilogues (nv_gemm_bf16_grouped_adamw; .grad of those weights is then None - torch's
        #      optimizer-in-backward trade), 2 = the same with the gradients still stored
        # None (default; NEUROVIT_FUSE_UPDATE unset) = by batch: 3 up to FUSE_MAX_ROWS token rows per step, 0 above.  ViT3D-base, same
        # box, volumes/s mode 0 -> 3: batch 2 641 -> 655 (+2.3 %), 4: 1113 -> 1165 (+4.8 %), 5: 1055 -> 1132, 6: 1152 -> 1191, 7: 1268 -> 1322, 8: 1390 -> 1382, 16: 1780 -> 1749,
        # 32: 1940 -> 1923; ViT3D-large 53.8 -> 53.3 - small batches leave wave slots and memory bandwidth idle beside the chain, large
        # ones do not.  Mode 1 at batch 2 / 4: +8 % / +2.5 % (profiles/r04_adamw_in_wgrad_epilogue.log)
        env = os.environ.get("NEUROVIT_FUSE_UPDATE")
        self.fuse_update = (int(env) if env is not None else None) if fuse_update is None else int(fuse_update)
        assert self.fuse_update in (None, 0, 1, 2, 3), "fuse_update: None (by batch), 0, 1, 2 or 3"
        self.last_fuse_update = 0          # what the most recent native step did
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
        self.last_outputs = None
        self._overlap_opt = bool(overlap_optimizer)
        self._opt_blocks = int(os.environ.get("NEUROVIT_OPT_BLOCKS", "0"))
        if grad_comm_dtype != torch.float32:           # 16-bit messages travel in the model's operand format (what the GEMM epilogues write)
            grad_comm_dtype = vit._dtype16()
        if self._arena_trainable and (world > 1 or overlap_optimizer):
            self.sync = GradSync(process_group, n_buckets, after_bucket=self._bucket_update if overlap_optimizer else None,
                                 comm_dtype=grad_comm_dtype, algo=grad_comm_algo)
            # bf16 messages: the fused AdamW reads the reduced bf16 gradients directly; param.grad then keeps the LOCAL
            # (pre-reduction) gradients, which nothing downstream of this fused step reads
            self.sync.write_back = bool(overlap_optimizer) or grad_comm_dtype == torch.float32
        if world > 1:
            arena, _ = vit.flat_parameters()
            broadcast_parameters(arena, process_group)
            for p in model.parameters():               # parameters outside the arena (4D temporal head)
                if not any(p is q for q in vit._plist):
                    dist.broadcast(p.data, src=0, group=process_group)
            vit._shadow_key = None
        self._pg = process_group
        vit._grad_sync = None
        self._ncomm = None                             # NativeComm of the native data-parallel step
        self._dp_world = world                         # (tests force the collective path on one rank by raising it)
        self._dp_buckets, self._dp_msg16 = n_buckets, grad_comm_dtype != torch.float32
        want_native_dp = (os.environ.get("NEUROVIT_NATIVE_DP", "1") != "0") if native_dp is None else bool(native_dp)
        dev_ = next(model.parameters()).device
        if want_native_dp and (world > 1 or native_dp) and self._arena_trainable and not overlap_optimizer and dev_.type == "cuda" \
                and (not dist.is_initialized() or dist.get_backend(process_group) == "nccl"):
            try:
                self._ncomm = NativeComm(dev_, process_group)
            except RuntimeError as e:
                import warnings
                warnings.warn(f"neurovit_amd: native data-parallel step unavailable ({e}); using the Python-driven bucket pipeline")
        # DP: if collectives would queue behind the current stream's kernels (shared hardware queue), run the step on a stream where
        # they do not, and give the engine an auxiliary stream with the same property (parallel.streams_beside_collectives)
        self._compute_stream = None
        dev0 = next(model.parameters()).device
        if world > 1 and dev0.type == "cuda" and self._ncomm is None:      # (the native data-parallel step issues its collectives on a stream it names itself)
            from .parallel import streams_beside_collectives
            self._compute_stream, aux = streams_beside_collectives(dev0, process_group)
            if aux is not None and vit._rt.use_aux_stream:
                vit._rt._aux[str(dev0)] = aux
        if os.environ.get("NEUROVIT_FORCE_COMPUTE_STREAM") == "1" and dev0.type == "cuda":     # tests: exercise the side-stream step
            self._compute_stream = torch.cuda.Stream(device=dev0)
        # parameters outside the ViT's arena (the 4D temporal head), found once: the per-step straggler loop must not search
        ids = {id(q) for q in vit.parameters()}
        self._outside = [p for p in model.parameters() if id(p) not in ids]
        self._inside = [p for p in model.parameters() if id(p) in ids]
        self._native = None                                            # decided at the first step (see _native_ok)
        self._graphs_on = os.environ.get("NEUROVIT_GRAPH_STEP", "0") == "1"   # replay the native step from captured graphs (see _graph_step)
        self._graphs, self._graph_seen, self._graph_warm = {}, {}, 0
        if loss_scale is None:
            loss_scale = "dynamic" if (vit.operands == "fp16" and self._arena_trainable) else 0.0
        self.scaler = None
        self.static_scale = 0.0
        if loss_scale == "dynamic":
            assert not overlap_optimizer, "a dynamic loss scale decides after the backward pass whether the step is applied: no per-bucket optimizer updates"
            self.scaler = LossScaler(dev0)
            if self.sync is not None:
                self.sync.write_back = True            # the overflow check reads the reduced gradients from the fp32 arena
        else:
            self.static_scale = float(loss_scale)
            assert self.static_scale >= 0.0, "loss_scale: None, 'dynamic' or a non-negative number"
        self.last_path = None                          # "native" | "general": which path the most recent step took (last_fuse_update: where AdamW ran)
        self._all_modules = list(model.modules())
        self._head = getattr(model, "_temporal_head", None)            # 4D: its 16 parameters are one arena (temporal.TemporalHead)
        self._head_ids = set()
        if self._head is not None:
            self._head.flat_parameters()
            self._head_ids = {id(p) for p in self._head._plist}

    def _bucket_update(self, begin: int, end: int):
        gs = 1.0 / self.world
        if self.static_scale > 0:
            gs /= self.static_scale
        self.optimizer.step_range(self._vit, begin, end, grad_scale=gs, max_blocks=self._opt_blocks)

    def __call__(self, fmri: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        cs = self._compute_stream
        if cs is None:
            return self._step(fmri, labels)
        cur = torch.cuda.current_stream(cs.device)
        cs.wait_stream(cur)
        with torch.cuda.stream(cs):
            loss = self._step(fmri, labels)
        cur.wait_stream(cs)
        for t in (loss, self.last_outputs):
            if t is not None:
                t.record_stream(cur)
        return loss

    # ------------------------------------------------------------------ the step as ONE native call (world = 1, 3D model)
    def _native_ok(self, fmri, labels) -> bool:
        """nv_vit_train_step runs the whole step - forward, CrossEntropyLoss, backward, AdamW - from native code when nothing needs
        the autograd graph or a collective in between: the 3D model with its whole arena trainable, one process, every parameter in
        the ViT's arena, no module hooks.  Everything else (4D, data parallel, partially frozen encoders, foreign .grad tensors)
        takes the general path below; both give the same parameters after a step (tests/test_modules_gpu.py)."""
        if self._native is None:
            model, vit = self.model, self._vit
            self._native = bool(
                os.environ.get("NEUROVIT_NATIVE_STEP", "1") != "0" and model.config.get('TRAINING_DIM') == 3
                and ((self.sync is None and self.world == 1) or self._ncomm is not None) and not self._outside and self._compute_stream is None)
        if not self._native or not (torch.is_tensor(fmri) and fmri.is_cuda and fmri.dtype == torch.float32 and fmri.dim() == 4):
            return False
        # checked on EVERY step (hooks registered, parameters frozen after the first step must not be bypassed: the native call
        # runs no module forward and updates the whole arena)
        if not all(p.requires_grad for p in self._vit._plist or self._vit.parameters()):
            return False
        if any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None) for m in self._all_modules):
            return False
        if not (torch.is_tensor(labels) and labels.is_cuda and labels.dtype == torch.int64 and labels.dim() == 1 and labels.shape[0] == fmri.shape[0]):
            return False
        vit = self._vit
        if vit.fp8_training and vit._fp8 is not None:
            return False                                                 # fp8 training forwards go through ViT._run_forward (general path)
        if not vit._arena_ok():
            vit._build_arena()
        # .grad tensors that are not views of the gradient arena (set by foreign code) need the general path's accumulate-and-copy;
        # the first and the last parameter stand for all (the arena path sets or clears every .grad together)
        for i in (0, len(vit._plist) - 1):
            g = vit._plist[i].grad
            if g is not None and (vit._grads is None or g.data_ptr() != vit._grad_view(i).data_ptr()):
                return False
        return True

    # ------------------------------------------------------------------ ... replayed from a captured graph
    def _graph_step(self, fmri: torch.Tensor, labels: torch.Tensor):
        """The native step's forward + loss + backward as a captured HIP graph, replayed with ONE host call (the ~225 launches of a step
        cost the host 1.6 ms to enqueue one by one, ~20 us as a graph); the AdamW update, whose bias-correction constants change every
        step, is launched behind it.  One graph per (input address, label address, shape): a loader that hands its batches over in a
        few recycled device buffers (DevicePrefetcher: two) hits the cache after the first pass over them; an address seen for the
        first time runs the step eagerly and is captured the NEXT time it shows up, so tensors that never repeat never pay a capture.
        Only with dropout off (the masks' seeds are kernel arguments) and without accumulation.  Returns None when this step is not
        for the graph.  The returned loss / logits tensors are the graph's own outputs: the next replay of the same graph overwrites
        them (clone what must outlive the next step - as with any captured graph)."""
        vit, opt = self._vit, self.optimizer
        if not self._graphs_on or self.accumulation_steps != 1 or vit._dropout_p != (0.0, 0.0) or self.scaler is not None:
            return None
        if vit._grads is None:
            vit._grads = torch.zeros_like(vit._arena)
        m_, v_ = opt.arena_state(vit)
        # everything a captured launch holds by address: a rebuilt gradient arena, re-created optimizer state or another form of the
        # last block at the same input address must not replay a graph that writes stale or freed buffers
        key = (fmri.data_ptr(), labels.data_ptr(), tuple(fmri.shape), tuple(fmri.stride()), vit._arena.data_ptr(), vit._shadow.data_ptr(),
               vit._grads.data_ptr(), m_.data_ptr(), v_.data_ptr(), vit._rt.workspace(fmri.shape[0], True, fmri.device).data_ptr(), vit._rt.rows_form,
               vit.operands, self.static_scale)
        ent = self._graphs.get(key)
        if ent is None:
            self._graph_seen[key] = self._graph_seen.get(key, 0) + 1
            if self._graph_seen[key] < 2 or len(self._graphs) >= 4 or self._graph_warm < 3:
                self._graph_warm += 1
                if len(self._graph_seen) > 64:
                    self._graphs_on = False                              # addresses never repeat: stop looking
                return None
            video = fmri.permute(0, 3, 1, 2).unsqueeze(1)
            vit.check_video(video, arena_checked=True)
            if vit._grads is None:
                vit._grads = torch.zeros_like(vit._arena)
            m, v = opt.arena_state(vit)
            g0 = opt.param_groups[0]
            torch.cuda.synchronize(fmri.device)
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph):
                    loss, logits = vit._rt.train_step(video, labels, vit._arena, vit._shadow, vit._grads, m, v, step=1, lr=g0["lr"], betas=g0["betas"],
                                                      eps=g0["eps"], weight_decay=g0["weight_decay"], accumulate=False, update=False,
                                                      loss_scale=self.static_scale)
            except Exception as e:                                       # capture refused (an unsupported call inside it): eager from now on
                self._graphs_on = False
                import warnings
                warnings.warn(f"neurovit_amd: the train step could not be captured as a graph ({e}); continuing with eager launches")
                return None
            ent = self._graphs[key] = (graph, loss, logits, fmri, labels)     # (the tensors are kept alive with the graph that reads them)
            # the capture itself executed nothing: fall through to the replay below
        graph, loss, logits = ent[0], ent[1], ent[2]
        vit._refresh_shadow()
        graph.replay()
        vit._rt.note_step_replayed(fmri.shape[0], fmri.permute(0, 3, 1, 2).unsqueeze(1))
        opt._steps += 1
        opt._step_arena(vit, 1.0 / self.static_scale if self.static_scale > 0 else 1.0)      # AdamW over the arena + 16-bit shadow (mark_shadow_fresh inside)
        vit._last_logits = logits
        self.last_outputs = logits
        if vit._plist[0].grad is None:
            for i, p in enumerate(vit._plist):
                p.grad = vit._grad_view(i)
        return loss.reshape(())

    def _native_step(self, fmri: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        got = self._graph_step(fmri, labels)
        if got is not None:
            return got
        vit, opt = self._vit, self.optimizer
        last_micro = (self._micro + 1) % self.accumulation_steps == 0
        video = fmri.permute(0, 3, 1, 2).unsqueeze(1)                    # NeuroEncoder.py:200-202 as a VIEW (the gather kernel reads the strides)
        vit.check_video(video, arena_checked=True)                       # (_native_ok has just validated / rebuilt the arena)
        arena, shadow = vit._arena, vit._shadow
        if vit._grads is None:
            vit._grads = torch.zeros_like(arena)
        grads = vit._grads
        vit._refresh_shadow()
        if not opt._bound:
            opt._bind()
        m, v = opt.arena_state(vit)
        g0 = opt.param_groups[0]
        # the update inside the weight-gradient GEMMs: only a step that overwrites its gradients AND updates (no accumulation window)
        fuse = self.fuse_update
        if fuse is None:
            fuse = 3 if fmri.shape[0] * vit.pos_embedding.shape[1] <= self.FUSE_MAX_ROWS else 0
        if self.accumulation_steps != 1 or vit._phantom or self.scaler is not None:
            fuse = 0                        # (a dynamic loss scale decides after the backward pass whether the update is applied)
        dp = None
        if self._ncomm is not None:
            fuse = 0                        # the update follows the all-reduce: per bucket on the communication stream, or once at the end
            dp = self._dp_plan(fmri, grads)
        self.last_fuse_update = fuse
        self.last_path = "native" if dp is None else "native-dp"
        accumulate = self._micro > 0        # the first micro-step of a window overwrites (zero_grad(set_to_none=True), Trainer.py:72), the others add
        loss, logits = vit._rt.train_step(video, labels.contiguous(), arena, shadow, grads, m, v, step=opt._steps + 1, lr=g0["lr"], betas=g0["betas"],
                                          eps=g0["eps"], weight_decay=g0["weight_decay"], grad_scale=1.0, accumulate=accumulate, update=last_micro,
                                          fuse_update=fuse, dropout=vit.draw_dropout(), loss_scale=self.static_scale,
                                          loss_scale_state=None if self.scaler is None else self.scaler.state, dp=dp)
        if last_micro:
            opt._steps += 1                 # (after the call: a refused step leaves the counter where it was)
        vit._last_logits = logits
        self.last_outputs = logits
        if vit._plist[0].grad is None:                                   # .grad = views of the gradient arena (once; they stay valid)
            for i, p in enumerate(vit._plist):
                p.grad = vit._grad_view(i)
        if fuse == 1 and vit._plist[10].grad is not None:
            # the layers' Linear weights were updated inside their gradient GEMMs and their gradients never written: no .grad rather
            # than a stale one (entries 8 + 11 l + {2, 3, 7, 9} of the parameter table: to_qkv, to_out, FC1, FC2 weights of block l)
            for l in range(vit._cfg.depth):
                for k in (2, 3, 7, 9):
                    vit._plist[8 + 11 * l + k].grad = None
        elif fuse != 1 and vit._plist[10].grad is None:
            for i, p in enumerate(vit._plist):
                p.grad = vit._grad_view(i)
        self._micro += 1
        if last_micro:
            vit.mark_shadow_fresh()
            self._micro = 0
        return loss.reshape(())

    def _dp_plan(self, fmri, grads):
        """nv_dp_plan of this step: communicator, a communication stream with a hardware queue of its own, bucket count, message format,
        and where AdamW runs - per bucket behind its all-reduce while the chip has room beside the backward pass (the same batch rule
        as fuse_update), else once when every bucket is in; a dynamic loss scale needs the single update (and fp32 messages)."""
        import ctypes
        from ._cabi import DpPlan
        nc, vit = self._ncomm, self._vit
        if nc.stream is None:
            from .parallel import independent_stream
            cur = torch.cuda.current_stream(fmri.device)
            aux = vit._rt._aux_stream(fmri.device) and vit._rt.aux_stream_object(fmri.device)
            nc.stream = independent_stream(fmri.device, [cur, aux])
        msg = None
        if self._dp_msg16 and self.scaler is None:
            if getattr(self, "_dp_msgbuf", None) is None or self._dp_msgbuf.numel() != grads.numel() or self._dp_msgbuf.dtype != vit._dtype16():
                self._dp_msgbuf = torch.empty(grads.numel(), dtype=vit._dtype16(), device=grads.device)
            msg = self._dp_msgbuf
        per_bucket = self.scaler is None and self.accumulation_steps == 1 and fmri.shape[0] * vit.pos_embedding.shape[1] <= self.FUSE_MAX_ROWS
        per_bucket = int(per_bucket)
        if os.environ.get("NEUROVIT_DP_UPDATE_PER_BUCKET") is not None and per_bucket:      # A/B aid: 0 = once at the end, 1 = comm stream, 2 = auxiliary stream one bucket late
            per_bucket = int(os.environ["NEUROVIT_DP_UPDATE_PER_BUCKET"])
        self.last_dp = dict(buckets=self._dp_buckets, messages="16-bit" if msg is not None else "fp32", update_per_bucket=per_bucket, world=self._dp_world)
        self._dp_keep = DpPlan(ctypes.sizeof(DpPlan), int(self._dp_world), nc.handle, nc.stream.cuda_stream, int(self._dp_buckets), int(per_bucket),
                               None if msg is None else msg.data_ptr())
        return self._dp_keep

    def _step(self, fmri: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        if self._native_ok(fmri, labels):
            return self._native_step(fmri, labels)
        model, vit = self.model, self._vit
        last_micro = (self._micro + 1) % self.accumulation_steps == 0
        pipelined = self.sync is not None and last_micro
        # the all-reduce / optimizer pipeline runs only on the micro-step that ends an accumulation window
        vit._grad_sync = self.sync if pipelined else None
        if pipelined and self._overlap_opt:
            self.optimizer.begin_step()
        self.last_path = "general"
        outputs = model(fmri)
        self.last_outputs = outputs.detach()           # the Trainer shell counts accuracy from these (3D and 4D alike)
        loss = self.criterion(outputs, labels)
        if self._micro == 0:
            self.optimizer.zero_grad(set_to_none=True)
        if self.scaler is not None:
            self.scaler.scale(loss).backward()         # Trainer.py:74: scaler.scale(loss).backward()
        elif self.static_scale > 0:
            (loss * self.static_scale).backward()
        else:
            loss.backward()
        self._micro += 1
        if last_micro:
            scale = 1.0 / self.world
            if self.world > 1:
                # stragglers, reduced inline: parameters outside the arena (the 10 k-parameter temporal head), and - when the
                # ViT is only PARTIALLY trainable, so that no bucket pipeline runs over the arena - its trainable parameters
                arena_synced = self.sync is not None
                inline = self._outside if arena_synced else self._outside + self._inside
                head = self._head
                if head is not None and head._grads is not None and all(p.grad is not None and p.grad.data_ptr() == head._grad_view(i).data_ptr()
                                                                        for i, p in enumerate(head._plist)):
                    dist.all_reduce(head._grads, group=self._pg)       # the temporal head's gradient arena: one message
                    if not arena_synced:
                        head._grads.mul_(scale)                        # (otherwise the fused step applies grad_scale)
                    inline = [p for p in inline if id(p) not in self._head_ids]
                # a head parameter whose gradient is NOT an arena view (stock-module path) is copied into the head's arena by
                # gather_foreign_grads and scaled there by the fused step's grad_scale: it must not be scaled here as well
                fused_scales_head = arena_synced and head is not None and any(h is head for h in self.optimizer.arenas())
                for p in inline:
                    if p.grad is not None:
                        dist.all_reduce(p.grad, group=self._pg)
                        if not (fused_scales_head and id(p) in self._head_ids):
                            p.grad.mul_(scale)
                if not arena_synced:
                    scale = 1.0                        # already averaged above
            if self.static_scale > 0:
                scale = scale / self.static_scale
            if pipelined and self._overlap_opt:
                vit.mark_shadow_fresh()                # every range was updated (and its bf16 shadow refreshed) by the buckets
                self.optimizer.step_rest(grad_scale=scale)
            else:
                red = self.sync.reduced_buffer() if (pipelined and not self.sync.write_back) else None
                self.optimizer.step(grad_scale=scale, reduced_bf16=None if red is None else {id(vit): red}, scaler=self.scaler)
            self._micro = 0
        return loss.detach()


class DevicePrefetcher:
    """Iterates a DataLoader one batch ahead: batch i+1 is copied host -> device on a side stream while step i computes.
    A 4-volume 128^3 batch is 33.5 MB = 0.53 ms over PCIe, 12 % of a 4.35 ms train step if it sits on the compute stream
    (the reference's `fMRI.to(device)` in the loop, Trainer.py:66).  Tensors in the batch move to the device, everything else
    (subject ids, strings) passes through.  With a CPU device it degenerates to the plain iterator."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self._stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def __len__(self):
        return len(self.loader)

    def _move(self, batch):
        if self._stream is None:
            return [b.to(self.device) if torch.is_tensor(b) else b for b in batch]
        with torch.cuda.stream(self._stream):
            return [b.to(self.device, non_blocking=True) if torch.is_tensor(b) else b for b in batch]

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._move(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur = nxt
            if self._stream is not None:
                torch.cuda.current_stream(self.device).wait_stream(self._stream)
                for b in cur:
                    if torch.is_tensor(b):
                        b.record_stream(torch.cuda.current_stream(self.device))
            try:
                nxt = self._move(next(it))       # overlaps the step the caller runs on `cur`
            except StopIteration:
                nxt = None
            yield cur


class Trainer:
    """The reference's Trainer shell (src/Trainer.py:14-167) on the native path: same constructor and methods
    (`run`, `train`, `validate`, `evaluate_samples`), same checkpoint files (plain `state_dict`s, interchangeable
    with the reference), same logging cadence.  Differences, all deliberate:
      * the step body is `TrainStep` (fused CE / staged backward / fused AdamW; no GradScaler: bf16 needs none);
      * wandb is optional (skipped when the package is absent or config['WANDB_ENABLED'] is false);
      * the host -> device copy of the next batch runs on a side stream under the current step (`DevicePrefetcher`);
      * batches may be the 7-tuples of DatasetADNI or the 6-tuples of DatasetADNI_4D (README.md:100-102 asks users
        to hand-edit the unpacking): the volume is element 2 and the label the last element in both;
      * `validate` / `evaluate_samples` run the encoder in config['VALIDATION_PRECISION'] (default "fp32": the reference validates
        in fp32 without autocast, Trainer.py:101-118; "bf16" = the training arithmetic, about 3x faster);
      * `log_interval = len(dl)//10` is clamped to >= 1 (the reference divides by zero for < 10 batches,
        Trainer.py:34,89); TRAINING_ACCUMULATION_STEP is honoured as in the commented block (Trainer.py:82-86).
    """

    def __init__(self, config, model, dataset_train, dataset_val):
        import os as _os
        self.config = config
        self.device = config['DEVICE']
        self.model = model.to(self.device)
        self.output_dir = config['GLOBAL_OUTPUT_DIR']
        self.epochs = config['TRAINING_EPOCHS']
        self.batch_size = config['TRAINING_BATCH_SIZE']
        self.num_workers = config['TRAINING_NUM_WORKERS']
        self.data, self.val_data = dataset_train, dataset_val
        kw = dict(batch_size=self.batch_size, num_workers=self.num_workers, pin_memory=True)
        if self.num_workers > 0:
            kw["prefetch_factor"] = 2
        self.dataloader = torch.utils.data.DataLoader(self.data, shuffle=True, **kw)
        self.val_dataloader = torch.utils.data.DataLoader(self.val_data, shuffle=False, **kw)
        self.criterion = CrossEntropyLoss()
        self.step = TrainStep(model, accumulation_steps=config.get('TRAINING_ACCUMULATION_STEP', 1) if config.get('USE_ACCUMULATION', False) else 1)
        self.optimizer = self.step.optimizer
        self.log_interval = max(1, len(self.dataloader) // 10)
        self._wandb = None
        if config.get('WANDB_ENABLED', False):
            try:
                import wandb
                self._wandb = wandb
            except ImportError:
                pass
        total_params = sum(p.numel() for p in self.model.parameters())
        trainable_params = sum(p.numel() for p in self.model.parameters() if p.requires_grad)
        print(f'Model total parameters: {total_params/1e6:.2f}M (trainable {trainable_params/1e6:.2f}M and frozen {(total_params-trainable_params)/1e6:.2f}M)')
        self._os = _os
        self.validation_precision = config.get('VALIDATION_PRECISION', 'fp32')

    @staticmethod
    def _unpack(batch):
        return batch[2], batch[-1]

    def _log(self, payload):
        if self._wandb is not None:
            self._wandb.log(payload)

    def run(self):
        import datetime
        path = f"{self.output_dir}/{datetime.datetime.now().strftime('%Y-%m-%d_%H-%M-%S')}"
        self._os.makedirs(path, exist_ok=True)
        self._os.makedirs('./results', exist_ok=True)
        for epoch in range(self.epochs):
            self.train(epoch)
            self.validate(epoch)
            torch.save(self.model.state_dict(), './results/last_model.pth')
            torch.save(self.model.state_dict(), f'{path}/model-e{epoch}.pth')
            print(f"MODEL SAVED to .{path}/model-e{epoch}.pth")

    def train(self, epoch):
        import time
        self.model.train()
        running_loss, correct, total = 0.0, 0, 0
        start_time = time.time()
        for i, batch in enumerate(DevicePrefetcher(self.dataloader, self.device)):     # H2D of batch i+1 under step i
            fMRI, label = self._unpack(batch)
            loss = self.step(fMRI, label)
            # the reference syncs twice per step (.item()); here statistics stay on the device until a log line is due
            running_loss = running_loss + loss
            with torch.no_grad():                      # Trainer.py:78-79, from the step's own outputs (no second forward, no sync)
                correct = correct + (self.step.last_outputs.argmax(dim=1) == label.to(self.step.last_outputs.device)).sum()
            total += label.size(0)
            if i != 0 and i % self.log_interval == 0:
                avg_loss = round(float(running_loss) / self.log_interval, 5)
                accuracy = round(float(correct) / total, 5)
                lr = round(self.optimizer.param_groups[0]['lr'], 5)
                duration = time.time() - start_time
                print(f"epoch {epoch}\t| batch {i}/{len(self.dataloader)}\t| train_loss: {avg_loss:.5f}\t| train_accuracy: {accuracy:.5f}\t| learning_rate: {lr:.5f}\t| duration: {duration:.2f}s")
                self._log({"epoch": epoch, "batch": i, "train_loss": avg_loss, "train_accuracy": accuracy, "learning_rate": lr, "duration": duration})
                correct, total, running_loss = 0, 0, 0.0
                start_time = time.time()

    def validate(self, epoch):
        self.model.eval()
        val_loss, correct, total, i = 0.0, 0, 0, 0
        with torch.no_grad(), self.model.precision(self.validation_precision):
            for i, batch in enumerate(self.val_dataloader):
                fMRI, label = self._unpack(batch)
                fMRI, label = fMRI.to(self.device), label.to(self.device)
                outputs = self.model(fMRI)
                val_loss += self.criterion(outputs, label).item()
                correct += (outputs.argmax(dim=1) == label).sum().item()
                total += label.size(0)
        avg_val_loss = round(val_loss / max(1, len(self.val_dataloader)), 5)
        self.val_loss = avg_val_loss
        accuracy = round(correct / max(1, total), 5)
        print(f"[VALIDATION] epoch {epoch}\t| total_batch {i}\t| val_loss {avg_val_loss:.5f}\t| val_accuracy {accuracy:.5f}")
        self._log({"epoch": epoch, "val_loss": avg_val_loss, "val_accuracy": accuracy})
        return avg_val_loss, accuracy

    def evaluate_samples(self):
        self.model.eval()
        loader = torch.utils.data.DataLoader(self.val_data, batch_size=1, shuffle=False, num_workers=self.num_workers)
        accuracy, wrong = 0, []
        with torch.no_grad(), self.model.precision(self.validation_precision):
            for batch in loader:
                fMRI, label = self._unpack(batch)
                prediction = self.model(fMRI.to(self.device)).argmax(dim=1).item()
                actual = int(label.item())
                if prediction != actual:
                    wrong.append((batch[0][0] if isinstance(batch[0], (list, tuple)) else batch[0], prediction, actual))
                accuracy += prediction == actual
        acc = accuracy / max(1, len(loader)) * 100
        print(f"Accuracy: {acc:.2f}%")
        print(f"Wrong predictions: {wrong}")
        return acc, wrong
