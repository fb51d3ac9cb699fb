"""Synthetic cube-localisation task of the reference's DatasetGradCAM (src/data/DatasetGradCAM.py:84-131), generated in memory.

A volume is `grid_noise` everywhere except one cube of ones whose corner is a multiple of `cube_size`; the label is the cube's
position index  x/c + (y/c) * k + (z/c) * k^2  with k = grid_size / cube_size (DatasetGradCAM.py:111-117).  The reference writes
pickles and reads lower-case config keys that its YAML does not have, so it cannot run as shipped; this module
restates the generator only - it is the self-checking workload for the train step + Grad-CAM path
(tests/test_cube_demo_gpu.py): a model that learns the task must light up the cube."""
from __future__ import annotations

import numpy as np
import torch


def cube_volumes(num_samples: int, grid_size: int, cube_size: int, grid_noise: float = 0.0, seed: int = 0):
    """-> (volumes float32 [num, S, S, S], labels int64 [num], corners int64 [num, 3])."""
    assert grid_size % cube_size == 0
    rs = np.random.RandomState(seed)
    k = grid_size // cube_size
    corners = rs.randint(0, k, size=(num_samples, 3)) * cube_size
    volumes = np.full((num_samples, grid_size, grid_size, grid_size), grid_noise, dtype=np.float32)
    for i, (x, y, z) in enumerate(corners):
        volumes[i, x:x + cube_size, y:y + cube_size, z:z + cube_size] = 1.0
    idx = corners // cube_size
    labels = idx[:, 0] + idx[:, 1] * k + idx[:, 2] * k * k
    return torch.from_numpy(volumes), torch.from_numpy(labels.astype(np.int64)), torch.from_numpy(corners.astype(np.int64))


def cam_mass_in_cube(cam: torch.Tensor, corner, cube_size: int, margin: int = 0) -> float:
    """Fraction of a Grad-CAM volume's mass that lies inside the cube grown by `margin` voxels on every side (clipped to the
    volume).  Patches that lie wholly inside or outside the cube are constant, and the patch LayerNorm (vit_3d.py:93) maps every
    constant patch to the same token: only patches that straddle a cube face carry signal, so a map is expected on the cube's
    faces - count the patches the cube touches (margin = patch size rounds the cube out to patch borders)."""
    S = cam.shape[0]
    lo = [max(0, int(v) - margin) for v in corner]
    hi = [min(S, int(v) + cube_size + margin) for v in corner]
    total = float(cam.sum())
    return float(cam[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].sum()) / total if total > 0 else 0.0
