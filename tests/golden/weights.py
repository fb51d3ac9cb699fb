"""Deterministic weights / inputs for the golden fixtures (own code, no reference import).

Both ``make_golden.py`` (which loads these tensors INTO the imported reference model
to produce expected outputs) and the tests (which load them into the oracle / the
HIP modules) regenerate the same tensors from a seed with numpy's legacy
``RandomState`` stream, which is stable across numpy versions and platforms.  So the
committed fixtures only need to hold expected OUTPUTS (plus per-tensor checksums of
the regenerated weights).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np
import torch


def vit_param_spec(*, image_size, image_patch_size, frames, frame_patch_size, num_classes, dim, depth,
                   heads, mlp_dim, channels=3, dim_head=64, **_) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) in the reference's ViT.state_dict() order (vit_3d.py:91-110).
    kind: 'w' weight matrix, 'b' bias, 'g' LN gain, 'e' embedding."""
    (H, Wd), (p1, p2) = (image_size if isinstance(image_size, tuple) else (image_size, image_size)), \
        (image_patch_size if isinstance(image_patch_size, tuple) else (image_patch_size, image_patch_size))   # vit_3d.py:80-81 pairs
    N = (H // p1) * (Wd // p2) * (frames // frame_patch_size)
    P = channels * p1 * p2 * frame_patch_size
    inner = heads * dim_head
    project_out = not (heads == 1 and dim_head == dim)
    spec = [("pos_embedding", (1, N + 1, dim), "e"), ("cls_token", (1, 1, dim), "e"),
            ("to_patch_embedding.1.weight", (P,), "g"), ("to_patch_embedding.1.bias", (P,), "b"),
            ("to_patch_embedding.2.weight", (dim, P), "w"), ("to_patch_embedding.2.bias", (dim,), "b"),
            ("to_patch_embedding.3.weight", (dim,), "g"), ("to_patch_embedding.3.bias", (dim,), "b")]
    for i in range(depth):
        a, f = f"transformer.layers.{i}.0.", f"transformer.layers.{i}.1."
        spec += [(a + "norm.weight", (dim,), "g"), (a + "norm.bias", (dim,), "b"),
                 (a + "to_qkv.weight", (3 * inner, dim), "w")]
        if project_out:
            spec += [(a + "to_out.0.weight", (dim, inner), "w"), (a + "to_out.0.bias", (dim,), "b")]
        spec += [(f + "net.0.weight", (dim,), "g"), (f + "net.0.bias", (dim,), "b"),
                 (f + "net.1.weight", (mlp_dim, dim), "w"), (f + "net.1.bias", (mlp_dim,), "b"),
                 (f + "net.4.weight", (dim, mlp_dim), "w"), (f + "net.4.bias", (dim,), "b")]
    spec += [("mlp_head.0.weight", (dim,), "g"), ("mlp_head.0.bias", (dim,), "b"),
             ("mlp_head.1.weight", (num_classes, dim), "w"), ("mlp_head.1.bias", (num_classes,), "b")]
    return spec


def temporal_param_spec() -> List[Tuple[str, Tuple[int, ...], str]]:
    """TemporalTransformer + ProjectionHead parameters (NeuroEncoder.py:207-230)."""
    t = "temporal_transformer.transformer.layers.0."
    return [(t + "self_attn.in_proj_weight", (6, 2), "w"), (t + "self_attn.in_proj_bias", (6,), "b"),
            (t + "self_attn.out_proj.weight", (2, 2), "w"), (t + "self_attn.out_proj.bias", (2,), "b"),
            (t + "linear1.weight", (2048, 2), "w"), (t + "linear1.bias", (2048,), "b"),
            (t + "linear2.weight", (2, 2048), "w"), (t + "linear2.bias", (2,), "b"),
            (t + "norm1.weight", (2,), "g"), (t + "norm1.bias", (2,), "b"),
            (t + "norm2.weight", (2,), "g"), (t + "norm2.bias", (2,), "b"),
            ("projection_head.projection_head.weight", (2, 2), "w"),
            ("projection_head.projection_head.bias", (2,), "b")]


def make_tensors(spec, seed: int, prefix: str = "") -> "OrderedDict[str, torch.Tensor]":
    """Deterministic fp32 tensors: weights ~ N(0, 1/fan_in) (so activations stay O(1)),
    biases ~ 0.1 N(0,1), LN gains ~ 1 + 0.1 N(0,1), embeddings ~ N(0,1).  One RandomState
    stream per tensor (seeded by crc32 of its name) so specs can grow without reshuffling."""
    out = OrderedDict()
    for name, shape, kind in spec:
        rs = np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 32))
        a = rs.standard_normal(size=shape).astype(np.float32)
        if kind == "w":
            a *= np.float32(1.0 / np.sqrt(shape[-1]))
        elif kind == "b":
            a *= np.float32(0.1)
        elif kind == "g":
            a = np.float32(1.0) + np.float32(0.1) * a
        out[prefix + name] = torch.from_numpy(a)
    return out


def make_volume(shape, seed: int) -> torch.Tensor:
    """Synthetic ADNI-shaped input: N(0,1) voxels, z-scored per volume as the dataset does
    ((x-mean)/(std+1e-8), reference src/data/DatasetADNI.py:213).  shape = (B, S, S, S[, T])."""
    rs = np.random.RandomState(seed)
    a = rs.standard_normal(size=shape).astype(np.float32)
    flat = a.reshape(shape[0], -1)
    flat = (flat - flat.mean(axis=1, keepdims=True)) / (flat.std(axis=1, keepdims=True) + np.float32(1e-8))
    return torch.from_numpy(flat.reshape(shape).astype(np.float32))


def checksums(sd: Dict[str, torch.Tensor]) -> Dict[str, np.ndarray]:
    """Per-tensor (sum, sum of squares) in float64 - stored in fixtures to prove regeneration."""
    return {k: np.array([v.double().sum().item(), (v.double() ** 2).sum().item()]) for k, v in sd.items()}


# Named configurations used by fixtures and tests -------------------------------------------------

TINY = dict(image_size=64, image_patch_size=16, frames=64, frame_patch_size=16, num_classes=2, dim=192,
            depth=4, heads=3, mlp_dim=384, channels=1, dim_head=64, pool="cls")          # BASELINE.json configs[0]
MICRO = dict(image_size=32, image_patch_size=8, frames=32, frame_patch_size=8, num_classes=2, dim=128,
             depth=2, heads=2, mlp_dim=256, channels=1, dim_head=64, pool="cls")         # fast unit tests
BASE = dict(image_size=128, image_patch_size=16, frames=128, frame_patch_size=16, num_classes=2, dim=768,
            depth=12, heads=12, mlp_dim=3072, channels=1, dim_head=64, pool="cls")       # BASELINE.json configs[1]


RECT = dict(image_size=(16, 24), image_patch_size=(8, 4), frames=12, frame_patch_size=4, num_classes=3, dim=64,
            depth=2, heads=2, mlp_dim=128, channels=2, dim_head=64, pool="cls")          # non-square images / patches, 2 channels


NOPROJ = dict(image_size=16, image_patch_size=8, frames=16, frame_patch_size=8, num_classes=2, dim=64, depth=2, heads=1,
              mlp_dim=128, channels=1, dim_head=64, pool="cls")                          # heads 1, dim_head == dim: no to_out (vit_3d.py:32)


def neuro_config(S: int, p: int, dim: int = 3, dataset: str = "adni", **extra) -> dict:
    """Minimal reference-style config dict (keys of configs/config.yaml that the model reads)."""
    cfg = dict(DEVICE="cpu", TRAINING_DIM=dim, TRAINING_DROPOUT=0.0, TRAINING_VIT_INPUT_SIZE=S,
               TRAINING_VIT_PATCH_SIZE=p, GRADCAM_CUBE_SIZE=8, DATASET_NAME=dataset, GRADCAM_THRESHOLD=5,
               GRADCAM_SLICE_DIM=2, GRADCAM_SLICE_IDX=S // 2, GLOBAL_BASE_PATH="", BEST_MODEL_PATH="")
    cfg.update(extra)
    return cfg
