"""Input contract of the encoder on the GPU: crop of the raw scanner volume + per-sample z-score.

The reference does this with numpy inside `Dataset.__getitem__` (src/data/DatasetADNI.py:211-214 for one 3D
timepoint, src/data/DatasetADNI_4D.py:85-88 for a whole 4D run): `raw[1:, 10:-9, 1:]` turns the 91 x 109 x 91 MNI grid
into 90^3 and `(x - x.mean()) / (x.std() + 1e-8)` normalises with the population standard deviation.  `zscore_crop`
is that operation as one HIP op (statistics accumulated in double, like numpy), for pipelines that keep raw volumes
on the device; it produces exactly the `[B, H, W, D]` / `[B, H, W, D, T]` float32 tensor `NeuroEncoder.forward` takes.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import torch

from ._cabi import check, lib

ADNI_CROP = ((1, None), (10, -9), (1, None))      # DatasetADNI.py:212


def _span(sl, size):
    lo, hi = sl
    lo = 0 if lo is None else (lo + size if lo < 0 else lo)
    hi = size if hi is None else (hi + size if hi < 0 else hi)
    if not (0 <= lo < hi <= size):
        raise ValueError(f"crop {sl} does not fit an axis of {size}")
    return lo, hi - lo


def zscore_crop(raw: torch.Tensor, crop: Sequence[Tuple[Optional[int], Optional[int]]] = ADNI_CROP, eps: float = 1e-8,
                return_stats: bool = False):
    """raw: [B, X, Y, Z] or [B, X, Y, Z, T], float32 or int16, any strides, on the MI355X.
    Returns float32 [B, Sx, Sy, Sz(, T)] (dense); with return_stats also [B, 2] = (mean, population std)."""
    if not raw.is_cuda:
        raise RuntimeError("neurovit_amd.preprocess.zscore_crop: input must live on the MI355X (cuda) device - there is no CPU fallback")
    if raw.dtype not in (torch.float32, torch.int16):
        raise TypeError(f"zscore_crop: dtype {raw.dtype} unsupported (float32 or int16)")
    four_d = raw.dim() == 5
    if raw.dim() not in (4, 5):
        raise ValueError("zscore_crop: expected [B, X, Y, Z] or [B, X, Y, Z, T]")
    v = raw if four_d else raw.unsqueeze(-1)
    B, X, Y, Z, T = v.shape
    (x0, sx), (y0, sy), (z0, sz) = (_span(c, s) for c, s in zip(crop, (X, Y, Z)))
    out = torch.empty((B, sx, sy, sz, T), dtype=torch.float32, device=raw.device)
    stats = torch.empty((B, 2), dtype=torch.float32, device=raw.device)
    nb = lib.nv_zscore_crop_workspace_bytes(B)
    ws = torch.empty(nb, dtype=torch.uint8, device=raw.device)
    strides = (ctypes.c_long * 5)(*v.stride())
    crop8 = (ctypes.c_int * 8)(x0, y0, z0, 0, sx, sy, sz, T)
    check(lib.nv_zscore_crop(v.data_ptr(), 0 if raw.dtype == torch.float32 else 1, ctypes.cast(strides, ctypes.c_void_p),
                             B, ctypes.cast(crop8, ctypes.c_void_p), eps, out.data_ptr(), stats.data_ptr(), ws.data_ptr(), nb,
                             torch.cuda.current_stream().cuda_stream), "nv_zscore_crop")
    out = out if four_d else out.squeeze(-1)
    return (out, stats) if return_stats else out


def crop_view(raw: torch.Tensor, crop: Sequence[Tuple[Optional[int], Optional[int]]] = ADNI_CROP) -> torch.Tensor:
    """The cropped volume as a strided VIEW of raw [B, X, Y, Z] (no copy)."""
    (x0, sx), (y0, sy), (z0, sz) = (_span(c, s) for c, s in zip(crop, raw.shape[1:4]))
    return raw[:, x0:x0 + sx, y0:y0 + sy, z0:z0 + sz]


def volume_sigma(raw: torch.Tensor, crop: Sequence[Tuple[Optional[int], Optional[int]]] = ADNI_CROP, eps: float = 1e-8) -> torch.Tensor:
    """[B] = population std of each cropped volume + eps (statistics in double; one pass over the raw volumes, nothing written back)."""
    if not raw.is_cuda:
        raise RuntimeError("neurovit_amd.preprocess.volume_sigma: input must live on the MI355X (cuda) device - there is no CPU fallback")
    if raw.dim() != 4 or raw.dtype not in (torch.float32, torch.int16):
        raise ValueError("volume_sigma: expected raw [B, X, Y, Z] float32 or int16")
    B, X, Y, Z = raw.shape
    (x0, sx), (y0, sy), (z0, sz) = (_span(c, s) for c, s in zip(crop, (X, Y, Z)))
    sigma = torch.empty(B, dtype=torch.float32, device=raw.device)
    nb = lib.nv_zscore_crop_workspace_bytes(B)
    ws = torch.empty(nb, dtype=torch.uint8, device=raw.device)
    strides = (ctypes.c_long * 5)(*raw.stride(), 0)
    crop8 = (ctypes.c_int * 8)(x0, y0, z0, 0, sx, sy, sz, 1)
    check(lib.nv_volume_sigma(raw.data_ptr(), 0 if raw.dtype == torch.float32 else 1, ctypes.cast(strides, ctypes.c_void_p), B,
                              ctypes.cast(crop8, ctypes.c_void_p), eps, sigma.data_ptr(), None, ws.data_ptr(), nb,
                              torch.cuda.current_stream().cuda_stream), "nv_volume_sigma")
    return sigma
