"""Data-parallel gradient synchronisation: one process per GPU, RCCL (torch.distributed "nccl") over xGMI.

The reference is single-device (main.py:41-46); DP is additive.  fMRI volumes are independent, so
the batch is sharded across ranks and the only collective is one SUM all-reduce of the gradients per optimizer
step - issued per BUCKET (a contiguous range of the flat gradient arena that becomes final when a group of
backward stages has run) on a side stream, so it overlaps the rest of the backward pass.  The 1/world_size
averaging is folded into the fused AdamW (grad_scale), not a separate pass.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def bucket_stages(n_stages: int, n_buckets: int) -> List[Tuple[int, int]]:
    """Split backward stages 0..n_stages-1 (head, layers last->first, embedding) into <= n_buckets contiguous
    groups of near-equal stage count.  Returns [(first_stage, last_stage)] in execution order."""
    n_buckets = max(1, min(n_buckets, n_stages))
    base, extra = divmod(n_stages, n_buckets)
    out, s = [], 0
    for b in range(n_buckets):
        cnt = base + (1 if b < extra else 0)
        out.append((s, s + cnt - 1))
        s += cnt
    return out


def _spin(stream: "torch.cuda.Stream", microseconds: int) -> None:
    """Keep `stream` busy for the given time with the library's own spin kernel (no private torch API)."""
    from ._cabi import check, lib
    check(lib.nv_spin_us(int(microseconds), stream.cuda_stream), "nv_spin_us")


def _runs_beside(a: "torch.cuda.Stream", b: "torch.cuda.Stream") -> bool:
    """True when work on stream a is not queued behind work on stream b.  HIP multiplexes its streams onto a few hardware queues
    (four per priority by default); two streams that share one execute in submission order - a wait enqueued on one (a bucket's
    all-reduce waiting for the weight-gradient stream) then stalls every later kernel of the other.  Probe: a spin kernel on b,
    a small fill on a; a's fill finishing first means separate queues."""
    x = torch.empty(64, device=a.device)
    torch.cuda.synchronize(a.device)
    with torch.cuda.stream(b):
        _spin(b, 2000)                               # 2 ms
        eb = torch.cuda.Event()
        eb.record(b)
    with torch.cuda.stream(a):
        x.fill_(1.0)
        ea = torch.cuda.Event()
        ea.record(a)
    ea.synchronize()
    beside = not eb.query()
    torch.cuda.synchronize(a.device)
    return beside


def independent_stream(device, others, tries: int = 12) -> "torch.cuda.Stream":
    """A stream from torch's pool that shares its hardware queue with none of `others` (None entries ignored); the last candidate
    if the probe never succeeds (correct either way - only the overlap is lost).  Priority streams are not an answer: a
    high-priority queue that mostly waits on events slowed the whole step by 80 % (measured)."""
    others = [o for o in others if o is not None]
    cand = torch.cuda.Stream(device=device)
    for _ in range(tries):
        if all(_runs_beside(cand, o) for o in others):
            return cand
        cand = torch.cuda.Stream(device=device)
    return cand


def streams_beside_collectives(device, process_group=None, candidates: int = 8):
    """Data-parallel start-up probe (world > 1): which streams does a collective overlap?  torch runs RCCL kernels on an internal
    stream of its own; a stream that shares that stream's hardware queue executes in submission order with every all-reduce instead
    of beside it.  Returns (compute, aux): `compute` is None when the current stream is fine, else a pool stream that is (the
    caller runs its step there); `aux` is a second such stream with a queue of its own for the engine's weight-gradient work (None:
    let the engine pick).  Every rank issues exactly candidates + 2 collectives, whatever it finds, so the ranks stay in step:
    nothing between the first and the last collective may end a rank's probe early - an error there is raised, never swallowed
    (a rank that silently left would strand its peers inside a collective).  The spin kernel is the library's own (nv_spin_us)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) < 2 or torch.device(device).type != "cuda":
        return None, None
    import os
    if os.environ.get("NEUROVIT_DP_PROBE", "1") == "0":      # `bench.py --no-probe`: no start-up collectives, the current stream and the
        return None, None                                     # engine's own choice of auxiliary stream (every rank must set it alike)
    t = torch.ones(1 << 20, device=device)
    side = torch.cuda.Stream(device=device)
    dist.all_reduce(t, group=process_group)                   # warm-up: communicator and internal stream exist after this
    torch.cuda.synchronize(device)

    def beside(stream) -> bool:
        with torch.cuda.stream(stream):
            _spin(stream, 4000)                               # 4 ms of spinning on the candidate
            es = torch.cuda.Event()
            es.record(stream)
        with torch.cuda.stream(side):
            dist.all_reduce(t, group=process_group)
            ec = torch.cuda.Event()
            ec.record(side)
        ec.synchronize()
        ok = not es.query()                                   # the collective finished while the candidate still spun
        torch.cuda.synchronize(device)
        return ok

    current = torch.cuda.current_stream(device)
    current_ok = beside(current)
    good = []
    for _ in range(candidates):
        cand = torch.cuda.Stream(device=device)
        if beside(cand):
            good.append(cand)
    compute = None
    if not current_ok and good:
        compute = good.pop(0)
    main = compute if compute is not None else current
    aux = next((c for c in good if _runs_beside(c, main)), None)
    return compute, aux


class NativeComm:
    """An RCCL communicator of the C-ABI library's own (csrc/comm.cpp: RCCL bound at run time), for the data-parallel form of the
    NATIVE train step (nv_vit_train_step + nv_dp_plan): the gradient all-reduce of every bucket is issued from native code on
    `stream`, between the backward stages, with no interpreter and no torch.distributed call inside a step.
    Built from a unique id that rank 0 creates and one broadcast on the existing torch.distributed group delivers; with no group
    (or a group of one) it is a one-rank communicator - the form the single-GPU rehearsal and tests use.  The library is pointed
    at the RCCL torch already loaded (torch/lib/librccl.so) so that one copy serves both."""

    def __init__(self, device, process_group=None):
        import ctypes
        import os
        from ._cabi import check, lib
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("neurovit_amd.NativeComm: RCCL communicators exist on the GPU only")
        grouped = dist.is_initialized()
        self.world = dist.get_world_size(process_group) if grouped else 1
        self.rank = dist.get_rank(process_group) if grouped else 0
        path = os.environ.get("NEUROVIT_RCCL_LIB") or os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        check(lib.nv_comm_load(path.encode() if os.path.exists(path) else None), "nv_comm_load")
        uid = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            buf = (ctypes.c_char * 128)()
            check(lib.nv_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)), "nv_comm_unique_id")
            uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if self.world > 1:
            backend = dist.get_backend(process_group)
            carrier = uid.to(self.device) if backend == "nccl" else uid
            dist.broadcast(carrier, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0, group=process_group)
            uid = carrier.cpu()
        raw = (ctypes.c_char * 128).from_buffer_copy(bytes(uid.tolist()))
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.nv_comm_init(ctypes.cast(raw, ctypes.c_void_p), self.world, self.rank, ctypes.byref(handle)), "nv_comm_init")
        self.handle = handle
        self.stream = None          # the stream the step's collectives run on (TrainStep picks one with a hardware queue of its own)

    def all_reduce(self, t: torch.Tensor, stream: Optional["torch.cuda.Stream"] = None) -> torch.Tensor:
        """In-place SUM over the ranks (fp32, or the current 16-bit operand format) on `stream` (default: the current one)."""
        from . import ops
        from ._cabi import check, lib
        assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, ops.op16())
        st = (stream or torch.cuda.current_stream(t.device)).cuda_stream
        check(lib.nv_comm_all_reduce(self.handle, t.data_ptr(), t.numel(), int(t.dtype != torch.float32), st), "nv_comm_all_reduce")
        return t

    def close(self):
        from ._cabi import lib
        if getattr(self, "handle", None):
            lib.nv_comm_destroy(self.handle)
            self.handle = None


class GradSync:
    """All-reduce (SUM) of gradient-arena ranges as they become final, optionally followed by a per-range callback
    (the fused AdamW of that range) on the same side stream, so both overlap the rest of the backward pass.
    Device agnostic: with CUDA tensors the work runs on a dedicated stream behind an event; with CPU tensors
    (gloo, tests) it runs inline.  world_size 1 skips the collective but keeps the overlapped callback."""

    ALGOS = ("allreduce", "rs_ag", "one_hop")

    def __init__(self, process_group=None, n_buckets: int = 4, after_bucket=None, comm_dtype: torch.dtype = torch.float32,
                 algo: Optional[str] = None):
        """comm_dtype = torch.bfloat16 sends each bucket as bf16 (cast on the side stream, sum, cast back): half the xGMI
        bytes (177 MB instead of 354 MB per step for ViT3D-base).  The rounding (2^-9 relative per element) is
        below the bf16 noise the gradients already carry from the MFMA operands; replicas stay bit-identical because every
        rank receives the same reduced values.

        algo (default: $NEUROVIT_DP_ALGO or "allreduce") - how a bucket is summed over the ranks:
          "allreduce"  one dist.all_reduce per bucket: RCCL picks ring / tree itself.  A ring all-reduce of S bytes moves
                       2 (W-1)/W S over ONE xGMI link per GPU (153 GB/s): 2.0 ms for ViT3D-base's 177 MB of bf16 gradients at W = 8;
          "rs_ag"      dist.reduce_scatter_tensor + dist.all_gather_into_tensor on a zero-padded staging copy of the bucket: the
                       same two phases a ring all-reduce is made of, as separate collectives (rank r owns shard r in between - the
                       form a sharded optimizer would hook into);
          "one_hop"    the fully-connected form the 8-GPU node is wired for (7 links per GPU): ONE all-to-all hands every rank its
                       shard of every peer's bucket (each byte crosses exactly one link, all 7 links of a GPU busy at once:
                       (W-1)/W S / 7 per link), the W copies are summed locally in fp32 in RANK ORDER (deterministic, and for bf16
                       messages rounded once instead of W-1 times), a second all-to-all returns the reduced shards.  Per-link bytes
                       2 S / W instead of 2 (W-1)/W S: 0.29 ms instead of 2.0 ms for the 177 MB above.
        Every algo leaves the same values on every rank (replicas stay bit-identical); "rs_ag" and "one_hop" sum in an order of
        their own, so against "allreduce" they agree to fp32 / bf16 rounding, and bit for bit at world 2."""
        assert comm_dtype in (torch.float32, torch.bfloat16, torch.float16)      # 16-bit messages: the model's operand format (the weight-gradient GEMMs write their share of the message buffer in it)
        import os
        self.algo = algo or os.environ.get("NEUROVIT_DP_ALGO", "allreduce")
        if self.algo not in self.ALGOS:
            raise ValueError(f"GradSync: algo must be one of {self.ALGOS}, got {self.algo!r}")
        self._stage = {}            # (dtype, device) -> staging buffers of the rs_ag / one_hop forms
        self.comm_dtype = comm_dtype
        self.write_back = True      # bf16 messages: cast the reduced values back into the fp32 gradients (False: the consumer
                                    # reads `reduced_buffer()` itself, e.g. the fused AdamW - saves a pass over the arena)
        self._comm_buf: Optional[torch.Tensor] = None
        self.mirrored = None        # bf16 messages: sorted element ranges the producer already wrote (rounded) into message_buffer();
                                    # only the rest of a bucket is converted here (ViT.mirrored_ranges, nv_vit_backward_stages16)
        self.pg = process_group
        self.n_buckets = n_buckets
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.after_bucket = after_bucket          # callable(begin, end) run behind the bucket's all-reduce
        self._comm_stream: Optional[torch.cuda.Stream] = None
        self._works = []
        self.bytes_reduced = 0

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def begin(self):
        self._works = []

    def message_buffer(self, flat_grads: torch.Tensor) -> Optional[torch.Tensor]:
        """bf16 messages: the arena-shaped bf16 buffer the buckets are sent from (allocated on first use)."""
        if self.comm_dtype == torch.float32:
            return None
        if self._comm_buf is None or self._comm_buf.numel() != flat_grads.numel() or self._comm_buf.device != flat_grads.device:
            self._comm_buf = torch.empty(flat_grads.numel(), dtype=self.comm_dtype, device=flat_grads.device)
        return self._comm_buf

    def _staging(self, like: torch.Tensor, padded: int, names):
        key = (like.dtype, str(like.device))
        bufs = self._stage.setdefault(key, {})
        out = []
        for name in names:
            b = bufs.get(name)
            if b is None or b.numel() < padded:
                b = bufs[name] = torch.empty(padded, dtype=like.dtype, device=like.device)
            out.append(b[:padded])
        return out

    def _sum_over_ranks(self, buf: torch.Tensor):
        """In-place SUM of the 1-D tensor `buf` over the ranks of the group, by the configured algorithm (see __init__)."""
        if self.algo == "allreduce":
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
            return
        W, n = self.world, buf.numel()
        rank = dist.get_rank(self.pg)
        per = (n + W - 1) // W
        per = (per + 7) // 8 * 8                         # 16-byte aligned shards for either dtype
        padded = per * W
        if self.algo == "rs_ag":
            stage, = self._staging(buf, padded, ("stage",))
            stage[:n].copy_(buf)
            if padded > n:
                stage[n:].zero_()
            shard = torch.empty(per, dtype=buf.dtype, device=buf.device)
            dist.reduce_scatter_tensor(shard, stage, op=dist.ReduceOp.SUM, group=self.pg)
            dist.all_gather_into_tensor(stage, shard, group=self.pg)
            buf.copy_(stage[:n])
            return
        # one_hop: all-to-all of shards, local fixed-order sum in fp32, all-to-all of the reduced shards
        stage, recv = self._staging(buf, padded, ("stage", "recv"))
        stage[:n].copy_(buf)
        if padded > n:
            stage[n:].zero_()
        dist.all_to_all_single(recv, stage, group=self.pg)             # recv row r = rank r's copy of MY shard
        rows = recv.view(W, per)
        acc = rows[0].float()
        for r in range(1, W):                                          # rank order: the same sum on every run and for every world layout
            acc = acc + rows[r].float()
        reduced = acc.to(buf.dtype)
        stage.view(W, per).copy_(reduced.unsqueeze(0).expand(W, per))  # my reduced shard, once per destination
        dist.all_to_all_single(recv, stage, group=self.pg)             # recv row r = rank r's reduced shard = shard r of the sum
        buf.copy_(recv[:n])

    def _reduce(self, flat_grads: torch.Tensor, chunk: torch.Tensor, begin: int, end: int):
        if self.comm_dtype == torch.float32:
            self.bytes_reduced += chunk.numel() * 4
            self._sum_over_ranks(chunk)
            return
        buf = self.message_buffer(flat_grads)[begin:end]
        if self.mirrored is None:
            buf.copy_(chunk)
        else:                          # the large ranges are in the buffer already: convert what lies between them
            from . import ops
            rest, cur = [], begin
            for b, e in self.mirrored:
                if e <= begin or b >= end:
                    continue
                if b > cur:
                    rest.append((cur, b))
                cur = max(cur, e)
            if cur < end:
                rest.append((cur, end))
            ops.cast_ranges_bf16(flat_grads, self.message_buffer(flat_grads), rest)
        self.bytes_reduced += buf.numel() * 2
        self._sum_over_ranks(buf)
        if self.write_back:
            chunk.copy_(buf)

    def bucket_ready(self, flat_grads: torch.Tensor, begin: int, end: int, also_after: Optional["torch.cuda.Stream"] = None):
        """also_after: a second stream whose enqueued work also writes this range (the engine's auxiliary stream when the
        backward call did not join it)."""
        if end <= begin or (self.world == 1 and self.after_bucket is None):
            return
        chunk = flat_grads[begin:end]
        if chunk.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = independent_stream(chunk.device, [torch.cuda.current_stream(chunk.device), also_after])
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._comm_stream.wait_event(ev)
            if also_after is not None:
                self._comm_stream.wait_stream(also_after)
            with torch.cuda.stream(self._comm_stream):
                if self.world > 1:
                    self._reduce(flat_grads, chunk, begin, end)                     # enqueued on the side stream
                if self.after_bucket is not None:
                    self.after_bucket(begin, end)
        else:
            if self.world > 1:
                self._reduce(flat_grads, chunk, begin, end)
            if self.after_bucket is not None:
                self.after_bucket(begin, end)

    def reduced_buffer(self) -> Optional[torch.Tensor]:
        """The flat bf16 buffer holding the reduced gradients of every bucket of this step (bf16 messages only)."""
        return self._comm_buf if self.comm_dtype != torch.float32 else None

    def finish(self):
        """Make the current stream wait for every outstanding bucket."""
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        self._works = []


def broadcast_parameters(flat_params: torch.Tensor, process_group=None, src: int = 0):
    """Identical replicas at start: one broadcast of the flat parameter arena."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.broadcast(flat_params, src=src, group=process_group)
