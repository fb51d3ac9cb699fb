"""Python side of the native encoder engine (csrc/engine.hip): flat parameter arenas + workspace.

Memory design (MI355X-first, 288 GB HBM per GPU): all parameters of a ViT live in ONE contiguous fp32
arena (master weights), mirrored by a bf16 shadow arena of identical element offsets that the MFMA
kernels read, plus one gradient arena and (optimizer) m / v arenas.  nn.Parameters are VIEWS into the
arena, so the fused AdamW kernel, the bf16 refresh and the data-parallel gradient all-reduce all work
on a few large contiguous ranges instead of ~150 small tensors.
"""
from __future__ import annotations

import ctypes
import os
import sys
import weakref
from typing import Dict, List, Optional, Tuple

import torch

from . import _cabi, ops
from ._cabi import TrainHparams, VitConfig, VitInput, check, lib

_BYTES = {torch.float32: 4, torch.bfloat16: 2, torch.float16: 2}


def make_config(*, image_size, image_patch_size, frames, frame_patch_size, num_classes, dim, depth, heads, mlp_dim,
                channels=3, dim_head=64, ln_eps=1e-5, pool='cls', image_width=0, patch_width=0, **_) -> VitConfig:
    """image_size / image_patch_size are the image / patch HEIGHT; image_width / patch_width (0 = same) the widths (vit_3d.py:80-81)."""
    if pool not in ('cls', 'mean'):
        raise ValueError("pool type must be either cls (cls token) or mean (mean pooling)")
    return VitConfig(image_size, image_patch_size, frames, frame_patch_size, channels, num_classes, dim, depth, heads,
                     dim_head, mlp_dim, ln_eps, int(pool == 'mean'), 0 if image_width == image_size else image_width,
                     0 if patch_width == image_patch_size else patch_width, int(heads == 1 and dim_head == dim))


def image_hw(cfg: VitConfig):
    """(image height, image width, patch height, patch width)"""
    return cfg.image_size, cfg.image_width or cfg.image_size, cfg.image_patch_size, cfg.patch_width or cfg.image_patch_size


def param_layout(cfg: VitConfig) -> Tuple[List[int], List[int], int]:
    """(offsets, numels, total) of the arena, in the reference's ViT.state_dict() order."""
    total = lib.nv_vit_param_count(ctypes.byref(cfg))
    if total < 0:
        check(-1, "nv_vit_param_count")
    cap = 8 + 11 * cfg.depth + 4 + 8
    off = (ctypes.c_long * cap)()
    num = (ctypes.c_long * cap)()
    cnt = lib.nv_vit_param_table(ctypes.byref(cfg), off, num, cap)
    if cnt < 0:
        check(cnt, "nv_vit_param_table")
    return list(off[:cnt]), list(num[:cnt]), int(total)


def flops_forward(cfg: VitConfig) -> float:
    """Algorithmic FLOPs per volume of one ViT forward (2 x MAC of the patch embedding, the four linears and the two
    attention products of every block, and the head; LayerNorm / softmax / GELU / bias excluded) - SURVEY.md 8(d).
    A train step counts 3 x this (backward = 2 x forward)."""
    H, Wd, p1, p2 = image_hw(cfg)
    N = (cfg.frames // cfg.frame_patch_size) * (H // p1) * (Wd // p2)
    n = N + 1
    P = cfg.channels * p1 * p2 * cfg.frame_patch_size
    d, inner, m = cfg.dim, cfg.heads * cfg.dim_head, cfg.mlp_dim
    per_layer = 2 * n * d * 3 * inner + 2 * n * n * inner + 2 * n * n * inner + 2 * n * inner * d + 4 * n * d * m
    return 2.0 * N * P * d + cfg.depth * per_layer + 2 * d * cfg.num_classes


class _Pass:
    """One training-layout forward whose activations a backward may still read: its workspace, its input (the backward re-gathers
    the patches), the form of the last block, the dropout draw.  `done`: a whole backward has run - the workspace may be refilled."""
    __slots__ = ("B", "ws", "video", "keep", "rows_form", "dropout", "dlogits", "done", "__weakref__")

    def __init__(self, B, ws, video, keep, rows_form, dropout, done=False):
        self.B, self.ws, self.video, self.keep, self.rows_form, self.dropout, self.dlogits, self.done = B, ws, video, keep, rows_form, dropout, None, done


class VitRuntime:
    """Executes ViT forward / backward through the native engine on caller-provided arenas.
    Training-layout workspaces are handed out per forward pass: a pass whose backward has not run yet (and whose autograd node is still
    alive) keeps its workspace, and the next training forward takes another one - as many live sets of activations as the caller's graph
    holds, the reference's autograd semantics (siamese / two-forward losses); the usual forward-backward loop never needs a second."""

    def __init__(self, cfg: VitConfig):
        self.cfg = cfg
        self._ws: Dict[Tuple[int, int, str], torch.Tensor] = {}
        self._last = None   # (B, training, workspace, video) of the most recent forward
        self._pool: Dict[Tuple[int, int, str], list] = {}   # further training workspaces (beyond self._ws[key]) for overlapping passes
        self._holder = {}   # workspace address -> weakref of the _Pass that filled it last
        self._cur: Optional[_Pass] = None     # the pass backward() runs against (the most recent training forward unless autograd names another)
        self.generation = 0          # counts forwards: a backward may only run against the forward that filled the workspace
        self.backward_done = False   # a backward of the most recent forward has run (gates the Grad-CAM gradient tap)
        self._aux = {}      # device -> auxiliary stream for the weight-gradient GEMMs
        self.use_aux_stream = os.environ.get("NEUROVIT_AUX_STREAM", "1") != "0"
        # last block under pool='cls': 0 = the library's process default (cls rows), 1 = every row (A/B runs, debugging)
        self.rows_form = 1 if os.environ.get("NEUROVIT_CLS_TAIL") == "0" else 0
        self._rows_form = 0
        self.operands = "bf16"       # 16-bit operand format of this runtime's calls (ViT.set_operands); set in the library before each of them

    def workspace(self, B: int, training, device) -> torch.Tensor:
        """training: False / True, or 2 for the fp32 inference layout."""
        key = (B, int(training), str(device))
        ws = self._ws.get(key)
        if ws is None:
            nbytes = lib.nv_vit_workspace_bytes(ctypes.byref(self.cfg), B, int(training))
            if nbytes < 0:
                check(-1, "nv_vit_workspace_bytes")
            ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
            pad = (-ws.data_ptr()) % 256
            ws = ws[pad:pad + nbytes]
            self._ws[key] = ws
        return ws

    def _training_workspace(self, B: int, device) -> torch.Tensor:
        """A training-layout workspace no pending pass still needs: the first one whose last pass has had its backward or is gone."""
        key = (B, 1, str(device))
        first = self.workspace(B, True, device)
        more = self._pool.setdefault(key, [])
        for ws in [first] + more:
            ref = self._holder.get(ws.data_ptr())
            rec = ref() if ref is not None else None
            if rec is None or rec.done:
                return ws
            # a pass only THIS runtime still refers to (self._cur, + `rec` here + the call's argument = 3): no autograd node holds it, and a direct
            # rt.backward() always means the most recent forward - which the caller is about to replace: its workspace is free (forward-only loops in
            # training layout, fp8 calibration: one workspace, not two)
            if rec is self._cur and sys.getrefcount(rec) <= 3:
                return ws
        nbytes = first.numel()
        ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
        pad = (-ws.data_ptr()) % 256
        ws = ws[pad:pad + nbytes]
        more.append(ws)
        return ws

    def _open_pass(self, B, ws, video, keep, rows_form, dropout, done=False) -> _Pass:
        rec = _Pass(B, ws, video, keep, rows_form, dropout, done)
        self._holder[ws.data_ptr()] = weakref.ref(rec)
        self._cur = rec
        return rec

    def pass_is_live(self, rec: _Pass) -> bool:
        """the workspace still holds THIS pass's activations (no later forward or train step has refilled it)"""
        ref = self._holder.get(rec.ws.data_ptr())
        return ref is not None and ref() is rec

    def _input_form(self, video: torch.Tensor, vol_sigma, time_points: int, rows_form: int = 0):
        """(B, nv_vit_input or None) after checking the extents: plain [B, C, F, H, W] view, or (time_points > 0) a contiguous
        4D batch [B / T, H, W, D, T]; vol_sigma marks RAW volumes (z-score folded into the patch LayerNorm)."""
        if not video.is_cuda:
            raise RuntimeError("neurovit_amd: ViT forward needs a CUDA/HIP tensor on MI355X - there is no CPU fallback")
        assert video.dtype == torch.float32 and video.dim() == 5
        c = self.cfg
        if time_points:
            want = (c.image_size, c.image_width or c.image_size, c.frames, time_points)
            if tuple(video.shape[1:]) != want or not video.is_contiguous() or time_points % 4 or c.channels != 1:
                raise ValueError(f"neurovit_amd: 4D batch of shape {tuple(video.shape)} (contiguous: {video.is_contiguous()}) does not match "
                                 f"[B, H, W, D, T] = [B, {', '.join(map(str, want))}] with T % 4 == 0")
            B = video.shape[0] * time_points
        else:
            want = (c.channels, c.frames, c.image_size, c.image_width or c.image_size)
            if tuple(video.shape[1:]) != want:
                raise ValueError(f"neurovit_amd: video of shape {tuple(video.shape)} does not match the model's "
                                 f"[B, channels, frames, height, width] = [B, {', '.join(map(str, want))}]")
            B = video.shape[0]
        if vol_sigma is None and not time_points and not rows_form:
            return B, None
        if vol_sigma is not None:
            assert vol_sigma.is_cuda and vol_sigma.dtype == torch.float32 and vol_sigma.numel() == video.shape[0] and vol_sigma.is_contiguous()
        return B, VitInput(None if vol_sigma is None else vol_sigma.data_ptr(), int(time_points), int(rows_form))

    def forward(self, video: torch.Tensor, params: torch.Tensor, params16: torch.Tensor, training: bool,
                dropout: Tuple[float, float, int] = (0.0, 0.0, 0), vol_sigma=None, time_points: int = 0, rows_form: Optional[int] = None) -> torch.Tensor:
        """video: [B, C, F, H, W] fp32 view (any strides).  Returns logits [B, num_classes] fp32.
        dropout = (p of the blocks, p of the embedding, seed) - (0, 0, *) in eval mode.
        vol_sigma / time_points: the optional input forms of nv_vit_forward_in (raw volumes; contiguous 4D batch).
        rows_form: 1 = the last block on every row (as the reference computes it), 2 = on the cls rows when eligible,
        None = this runtime's default (`self.rows_form`: the process default unless NEUROVIT_CLS_TAIL=0)."""
        rows_form = self.rows_form if rows_form is None else int(rows_form)
        _cabi.set_operand_format(self.operands)
        B, inp = self._input_form(video, vol_sigma, time_points, rows_form)
        ws = self._training_workspace(B, video.device) if training else self.workspace(B, training, video.device)
        logits = torch.empty((B, self.cfg.num_classes), dtype=torch.float32, device=video.device)
        check(lib.nv_vit_forward_in(ctypes.byref(self.cfg), B, video.data_ptr(), ops.shape5(video), ops.strides5(video),
                                    None if inp is None else ctypes.cast(ctypes.pointer(inp), ctypes.c_void_p),
                                    params.data_ptr(), params16.data_ptr(), ws.data_ptr(), ws.numel(), int(training), float(dropout[0]),
                                 float(dropout[1]), int(dropout[2]), logits.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream), "nv_vit_forward_in")
        self._keep = (vol_sigma, inp)                    # the backward re-gathers the same (raw) input
        self._rows_form = rows_form                      # ... and takes the same form of the last block
        self._last = (B, training, ws, video)
        if training:
            self._open_pass(B, ws, video, self._keep, rows_form, dropout)
        self.generation += 1
        self.backward_done = False
        self._dropout = dropout
        return logits

    # ------------------------------------------------------------------ inference forward with the LayerNorms folded into the GEMMs around them
    def lnfold_prepare(self, params: torch.Tensor, reuse=None):
        """W_qkv diag(gamma1) / W_1 diag(gamma2) of every block in the operand format (an arena with the parameter arena's element offsets), their
        column sums and the folded biases (nv_vit_lnfold_prepare).  reuse: a dict this function returned earlier - overwritten in place."""
        _cabi.set_operand_format(self.operands)
        n32 = lib.nv_vit_lnfold_floats(ctypes.byref(self.cfg))
        if n32 < 0:
            check(-1, "nv_vit_lnfold_floats")
        dt = torch.float16 if self.operands == "fp16" else torch.bfloat16
        if reuse is not None and reuse["fold16"].numel() == params.numel() and reuse["fold16"].dtype == dt and reuse["fold16"].device == params.device:
            f16, f32 = reuse["fold16"], reuse["fold32"]
        else:
            f16 = torch.zeros(params.numel(), dtype=dt, device=params.device)
            f32 = torch.empty(n32, dtype=torch.float32, device=params.device)
        check(lib.nv_vit_lnfold_prepare(ctypes.byref(self.cfg), params.data_ptr(), f16.data_ptr(), f32.data_ptr(), torch.cuda.current_stream().cuda_stream),
              "nv_vit_lnfold_prepare")
        return dict(fold16=f16, fold32=f32)

    def forward_lnfold(self, video: torch.Tensor, params: torch.Tensor, params16: torch.Tensor, fold, vol_sigma=None, time_points: int = 0,
                       rows_form: Optional[int] = None) -> torch.Tensor:
        """Inference forward (no dropout) whose blocks run without LayerNorm launches: nv_vit_forward_lnfold."""
        rows_form = self.rows_form if rows_form is None else int(rows_form)
        _cabi.set_operand_format(self.operands)
        B, inp = self._input_form(video, vol_sigma, time_points, rows_form)
        ws = self.workspace(B, False, video.device)
        logits = torch.empty((B, self.cfg.num_classes), dtype=torch.float32, device=video.device)
        check(lib.nv_vit_forward_lnfold(ctypes.byref(self.cfg), B, video.data_ptr(), ops.shape5(video), ops.strides5(video),
                                        None if inp is None else ctypes.cast(ctypes.pointer(inp), ctypes.c_void_p), params.data_ptr(), params16.data_ptr(),
                                        fold["fold16"].data_ptr(), fold["fold32"].data_ptr(), ws.data_ptr(), ws.numel(), logits.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream), "nv_vit_forward_lnfold")
        self._keep = (vol_sigma, inp)
        self._rows_form = rows_form
        self._last = (B, False, ws, video)
        self.generation += 1
        self.backward_done = False
        self._dropout = (0.0, 0.0, 0)
        return logits

    # ------------------------------------------------------------------ fp32 inference (the reference's fp32 validate, Trainer.py:101-118)
    def forward_f32(self, video: torch.Tensor, params: torch.Tensor, vol_sigma=None, time_points: int = 0) -> torch.Tensor:
        """Inference forward with every operand in fp32 (weights straight from the fp32 arena, contractions on the fp32 MFMA):
        logits within 1e-5 of the reference's CPU fp32 forward instead of the bf16 path's 1e-3 ... 7e-3.  Eval mode only."""
        B, inp = self._input_form(video, vol_sigma, time_points, self.rows_form)
        ws = self.workspace(B, 2, video.device)
        logits = torch.empty((B, self.cfg.num_classes), dtype=torch.float32, device=video.device)
        check(lib.nv_vit_forward_f32(ctypes.byref(self.cfg), B, video.data_ptr(), ops.shape5(video), ops.strides5(video),
                                     None if inp is None else ctypes.cast(ctypes.pointer(inp), ctypes.c_void_p), params.data_ptr(),
                                     ws.data_ptr(), ws.numel(), logits.data_ptr(), torch.cuda.current_stream().cuda_stream), "nv_vit_forward_f32")
        self._last = (B, 2, ws, video)          # 2 = fp32 inference layout (truthy as `training` only for the layout queries)
        self.generation += 1
        self.backward_done = False
        return logits

    # ------------------------------------------------------------------ fp8 inference (BASELINE.json configs[4])
    def calibrate_fp8(self, video: torch.Tensor, params: torch.Tensor, params16: torch.Tensor, headroom: float = 2.0, out_proj: bool = True):
        """One bf16 forward of `video` with every layer's activations kept; returns the per-layer activation scales
        [depth][4] = 448 / (headroom * amax) of the LN1 output, the LN2 output, the GELU output and the attention output.  e4m3 is a floating
        format: headroom costs no relative precision, it only moves the subnormal floor."""
        self.forward(video, params, params16, training=True, rows_form=1)    # the fp8 forward quantises every row of every block: calibrate on every row
        B = video.shape[0]
        H, Wd, p1, p2 = image_hw(self.cfg)
        n = (self.cfg.frames // self.cfg.frame_patch_size) * (H // p1) * (Wd // p2) + 1
        scales = []
        for l in range(self.cfg.depth):
            row = []
            for name, width in (("xn1", self.cfg.dim), ("xn2", self.cfg.dim), ("h", self.cfg.mlp_dim), ("ao", self.cfg.heads * self.cfg.dim_head)):
                amax = float(self.tap(name, l, (B * n, width), torch.bfloat16).abs().max())     # calibration only: off the hot path
                row.append(448.0 / (headroom * max(amax, 1e-6)))
            if not out_proj:
                row[3] = 0.0          # the out-projection stays on bf16 operands (measured: fp8 there buys 1-2 % and adds ~15 % to the logits' error)
            scales.append(row)
        return scales

    def quantize_fp8(self, params: torch.Tensor, act_scales, reuse=None):
        """fp8 weight arena (bytes at the parameter arena's element offsets) + column scales for the given activation scales.
        reuse: a dict this function returned earlier - its device buffers are overwritten in place (the per-step re-quantisation of
        an fp8 training run must not allocate and clear a parameter-sized arena every step)."""
        cnt = lib.nv_vit_fp8_scale_count(ctypes.byref(self.cfg))
        if cnt < 0:
            check(-1, "nv_vit_fp8_scale_count")
        flat = [float(v) for row in act_scales for v in row]
        host = (ctypes.c_float * len(flat))(*flat)
        if reuse is not None and reuse["params8"].numel() == params.numel() and reuse["params8"].device == params.device:
            p8, cs = reuse["params8"], reuse["colscales"]
        else:
            p8 = torch.zeros(params.numel(), dtype=torch.uint8, device=params.device)
            cs = torch.empty(cnt, dtype=torch.float32, device=params.device)
        check(lib.nv_vit_quantize_fp8(ctypes.byref(self.cfg), params.data_ptr(), ctypes.cast(host, ctypes.c_void_p), p8.data_ptr(), cs.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream), "nv_vit_quantize_fp8")
        return dict(params8=p8, colscales=cs, act_scales=host, act_list=act_scales)

    def forward_fp8_train(self, video: torch.Tensor, params: torch.Tensor, params16: torch.Tensor, f8, dropout: Tuple[float, float, int] = (0.0, 0.0, 0),
                          vol_sigma=None, rows_form: Optional[int] = None) -> torch.Tensor:
        """TRAINING forward with qkv / FC1 / FC2 of every block on e4m3 operands (nv_vit_forward_fp8_train): fills the training
        workspace exactly as forward(training=True) does, so backward() follows as usual (on bf16 operands); dropout as in forward()."""
        rows_form = self.rows_form if rows_form is None else int(rows_form)
        _cabi.set_operand_format(self.operands)
        B, inp = self._input_form(video, vol_sigma, 0, rows_form)
        ws = self._training_workspace(B, video.device)
        logits = torch.empty((B, self.cfg.num_classes), dtype=torch.float32, device=video.device)
        check(lib.nv_vit_forward_fp8_train(ctypes.byref(self.cfg), B, video.data_ptr(), ops.shape5(video), ops.strides5(video),
                                           None if inp is None else ctypes.cast(ctypes.pointer(inp), ctypes.c_void_p), params.data_ptr(),
                                           params16.data_ptr(), f8["params8"].data_ptr(), f8["colscales"].data_ptr(),
                                           ctypes.cast(f8["act_scales"], ctypes.c_void_p), ws.data_ptr(), ws.numel(), float(dropout[0]), float(dropout[1]), int(dropout[2]),
                                           logits.data_ptr(), torch.cuda.current_stream().cuda_stream), "nv_vit_forward_fp8_train")
        self._keep = (vol_sigma, inp)
        self._rows_form = rows_form
        self._last = (B, True, ws, video)
        self._open_pass(B, ws, video, self._keep, rows_form, dropout)
        self.generation += 1
        self.backward_done = False
        self._dropout = dropout
        return logits

    def forward_fp8(self, video: torch.Tensor, params: torch.Tensor, params16: torch.Tensor, f8, vol_sigma=None, time_points: int = 0) -> torch.Tensor:
        _cabi.set_operand_format(self.operands)
        B, inp = self._input_form(video, vol_sigma, time_points, self.rows_form)
        ws = self.workspace(B, False, video.device)
        logits = torch.empty((B, self.cfg.num_classes), dtype=torch.float32, device=video.device)
        check(lib.nv_vit_forward_fp8(ctypes.byref(self.cfg), B, video.data_ptr(), ops.shape5(video), ops.strides5(video),
                                     None if inp is None else ctypes.cast(ctypes.pointer(inp), ctypes.c_void_p), params.data_ptr(),
                                     params16.data_ptr(), f8["params8"].data_ptr(), f8["colscales"].data_ptr(),
                                     ctypes.cast(f8["act_scales"], ctypes.c_void_p), ws.data_ptr(), ws.numel(), logits.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream), "nv_vit_forward_fp8")
        self._last = (B, False, ws, video)
        self.generation += 1
        self.backward_done = False
        return logits

    def backward(self, dlogits: torch.Tensor, params: torch.Tensor, params16: torch.Tensor, grads: torch.Tensor,
                 accumulate: bool, stages: Optional[Tuple[int, int]] = None, join_aux: bool = True,
                 grads16: Optional[torch.Tensor] = None) -> None:
        """Whole backward, or only stages [first, last] (0 = head, 1+k = layer depth-1-k, depth+1 = embedding).
        join_aux=False (only for ranges before the last stage): the current stream is not made to wait for the auxiliary
        stream - order the consumer of the range's gradients after `aux_stream_object()` as well."""
        rec = self._cur
        assert rec is not None, "backward needs a preceding forward(training=True)"
        if not self.pass_is_live(rec):
            raise RuntimeError("neurovit_amd: backward() of a forward pass whose activations have been overwritten - a later forward or train step "
                               "has refilled its workspace (a pass keeps its workspace only until a whole backward of it has run)")
        B, ws, video = rec.B, rec.ws, rec.video
        if rec.keep[1] is not None and rec.keep[1].time_points:
            raise NotImplementedError("neurovit_amd: the fused 4D input form is forward-only (the 4D model's encoder is frozen, NeuroEncoder.py:34-36)")
        first, last = (0, self.cfg.depth + 1) if stages is None else stages
        _cabi.set_operand_format(self.operands)
        if first == 0:
            rec.dlogits = dlogits.contiguous().float()
        # grads16: bf16 arena (element offsets of `grads`) that also receives the Linear weight gradients, rounded, straight from
        # their GEMMs - the data-parallel message buffer (mirrored_ranges() lists what lands there)
        check(lib.nv_vit_backward_stages16(ctypes.byref(self.cfg), B, video.data_ptr(), ops.strides5(video), params.data_ptr(),
                                           params16.data_ptr(), ws.data_ptr(), ws.numel(), rec.dlogits.data_ptr(),
                                           grads.data_ptr(), None if grads16 is None else grads16.data_ptr(), int(accumulate), first, last,
                                           float(rec.dropout[0]), float(rec.dropout[1]), int(rec.dropout[2]),
                                           torch.cuda.current_stream().cuda_stream, self._aux_stream(video.device), int(join_aux), int(rec.rows_form)),
              "nv_vit_backward_stages16")
        if last == self.cfg.depth + 1:
            rec.done = True              # the workspace may be refilled by the next training forward

    def train_step(self, video: torch.Tensor, labels: torch.Tensor, params: torch.Tensor, params16: torch.Tensor, grads: torch.Tensor,
                   adam_m: torch.Tensor, adam_v: torch.Tensor, *, step: int, lr: float, betas, eps: float, weight_decay: float,
                   grad_scale: float = 1.0, accumulate: bool = False, update: bool = True, fuse_update: int = 0,
                   dropout: Tuple[float, float, int] = (0.0, 0.0, 0), vol_sigma=None, rows_form: Optional[int] = None,
                   loss_scale: float = 0.0, loss_scale_state: Optional[torch.Tensor] = None, dp=None):
        """The reference's whole train step (Trainer.py:65-79) as ONE native call (nv_vit_train_step): forward, CrossEntropyLoss,
        backward of every stage, AdamW over the arena + bf16 shadow refresh.  Returns (loss [1], logits [B, C]) on the device.
        Same launches, streams and arithmetic as forward() + ops.ce_loss + backward() + ops.adamw_step.
        fuse_update (only with update and not accumulate): 1 = the transformer layers' Linear weights are updated by their
        weight-gradient GEMMs (their gradients are then not left in `grads`), 2 = the same but they are; same bits either way.
        loss_scale: static factor on d(loss)/d(logits), divided out again by the update; loss_scale_state: the device block of a
        dynamic loss scale (optim.LossScaler) - GradScaler semantics on the device, needs fuse_update = 0."""
        rows_form = self.rows_form if rows_form is None else int(rows_form)
        _cabi.set_operand_format(self.operands)
        B, inp = self._input_form(video, vol_sigma, 0, rows_form)
        ws = self.workspace(B, True, video.device)
        dev = video.device
        logits = torch.empty((B, self.cfg.num_classes), dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dlogits = torch.empty((B, self.cfg.num_classes), dtype=torch.float32, device=dev)
        assert labels.is_cuda and labels.dtype == torch.int64 and labels.numel() == B and labels.is_contiguous()
        hp = TrainHparams(ctypes.sizeof(TrainHparams), int(step), float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                          float(grad_scale), int(bool(accumulate)), int(bool(update)), int(fuse_update), float(loss_scale),
                          None if dp is None else ctypes.cast(ctypes.pointer(dp), ctypes.c_void_p),      # _cabi.DpPlan (kept alive by the caller)
                          None if loss_scale_state is None else loss_scale_state.data_ptr())
        check(lib.nv_vit_train_step(ctypes.byref(self.cfg), B, video.data_ptr(), ops.shape5(video), ops.strides5(video),
                                    None if inp is None else ctypes.cast(ctypes.pointer(inp), ctypes.c_void_p),
                                    params.data_ptr(), params16.data_ptr(), grads.data_ptr(), adam_m.data_ptr(), adam_v.data_ptr(),
                                    ws.data_ptr(), ws.numel(), labels.data_ptr(), logits.data_ptr(), loss.data_ptr(), dlogits.data_ptr(),
                                    ctypes.byref(hp), float(dropout[0]), float(dropout[1]), int(dropout[2]),
                                    torch.cuda.current_stream().cuda_stream, self._aux_stream(dev)), "nv_vit_train_step")
        self._keep = (vol_sigma, inp)
        self._rows_form = rows_form
        self._last = (B, True, ws, video)
        self._open_pass(B, ws, video, self._keep, rows_form, dropout, done=True).dlogits = dlogits     # (a pending pass in this workspace is now stale)
        self.generation += 1
        self.backward_done = True                        # the Grad-CAM gradient tap holds this step's gradient
        self._dropout = dropout
        return loss, logits

    def note_step_replayed(self, B: int, video: torch.Tensor) -> None:
        """A captured train step (trainer.py) has been replayed: the primary training workspace holds ITS activations and gradient taps."""
        ws = self.workspace(B, True, video.device)
        self._last = (B, True, ws, video)
        self._open_pass(B, ws, video, getattr(self, "_keep", (None, None)), self._rows_form, (0.0, 0.0, 0), done=True)
        self.generation += 1
        self.backward_done = True
        self._dropout = (0.0, 0.0, 0)

    def aux_stream_object(self, device) -> Optional[torch.cuda.Stream]:
        """The torch stream object behind the engine's auxiliary stream (None when the engine runs single-stream)."""
        return self._aux.get(str(device)) if self.use_aux_stream else None

    def _aux_stream(self, device):
        if not self.use_aux_stream:
            return None
        st = self._aux.get(str(device))
        if st is None:
            from .parallel import independent_stream     # a stream that does not share the current stream's hardware queue
            st = self._aux[str(device)] = independent_stream(device, [torch.cuda.current_stream(device)])
        return st.cuda_stream

    def stage_range(self, first: int, last: int) -> Tuple[int, int]:
        """Arena element range [begin, end) that is final after backward stages first..last have run."""
        lo, hi = None, None
        b, e = ctypes.c_long(), ctypes.c_long()
        for s in range(first, last + 1):
            check(lib.nv_vit_stage_param_range(ctypes.byref(self.cfg), s, ctypes.byref(b), ctypes.byref(e)), "nv_vit_stage_param_range")
            lo = b.value if lo is None else min(lo, b.value)
            hi = e.value if hi is None else max(hi, e.value)
        return lo, hi

    def tap(self, name: str, layer: int, shape, dtype) -> torch.Tensor:
        """View of a named activation inside the last forward's workspace (tests, Grad-CAM hooks)."""
        B, training, ws, _ = self._last
        off = lib.nv_vit_workspace_offset(ctypes.byref(self.cfg), B, int(training), name.encode(), layer)
        if off < 0:
            raise KeyError(f"no workspace buffer {name!r} (layer {layer})")
        n = 1
        for s in shape:
            n *= s
        return ws[off:off + n * _BYTES[dtype]].view(dtype).view(*shape)
