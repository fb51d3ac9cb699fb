"""Config loading: the reference's flat UPPER_CASE YAML (configs/config.yaml, config4D.yaml; main.py:49-62)
plus the optional model-size keys this build adds (defaults = the reference's hard-coded constants)."""
from __future__ import annotations

from typing import Optional

import yaml

OPTIONAL_MODEL_KEYS = {
    "TRAINING_VIT_DIM": 1024,       # NeuroEncoder.py:187
    "TRAINING_VIT_DEPTH": 6,        # NeuroEncoder.py:188
    "TRAINING_VIT_HEADS": 8,        # NeuroEncoder.py:189
    "TRAINING_VIT_DIM_HEAD": 64,    # vit_3d.py:78 default
    "TRAINING_VIT_MLP_DIM": 2048,   # NeuroEncoder.py:190
    "TRAINING_VIT_EVAL_PRECISION": "bf16",   # arithmetic of model.eval() no-grad forwards: "bf16" | "fp32"
    "VALIDATION_PRECISION": "fp32",          # Trainer.validate / evaluate_samples (the reference validates in fp32: Trainer.py:101-118)
}

REQUIRED_MODEL_KEYS = ("TRAINING_DIM", "TRAINING_DROPOUT", "TRAINING_VIT_INPUT_SIZE", "TRAINING_VIT_PATCH_SIZE",
                       "GRADCAM_CUBE_SIZE", "DATASET_NAME")


def load_config(path: str, device: Optional[str] = None, **overrides) -> dict:
    """YAML -> dict the way main.py:get_config does (DEVICE injected by the caller's CLI there)."""
    with open(path) as f:
        config = yaml.safe_load(f)
    for k, v in OPTIONAL_MODEL_KEYS.items():
        config.setdefault(k, v)
    config.update(overrides)
    if device is not None:
        config["DEVICE"] = device
    missing = [k for k in REQUIRED_MODEL_KEYS if k not in config]
    if missing:
        raise KeyError(f"config {path} lacks keys read by the model: {missing}")
    return config


def preset(name: str) -> dict:
    """Model-size presets of BASELINE.json's configs (mlp_dim = 4*dim for base/large, 2x for tiny - SURVEY 0)."""
    presets = {
        "tiny": dict(TRAINING_VIT_INPUT_SIZE=64, TRAINING_VIT_PATCH_SIZE=16, TRAINING_VIT_DIM=192, TRAINING_VIT_DEPTH=4,
                     TRAINING_VIT_HEADS=3, TRAINING_VIT_MLP_DIM=384),
        "base": dict(TRAINING_VIT_INPUT_SIZE=128, TRAINING_VIT_PATCH_SIZE=16, TRAINING_VIT_DIM=768, TRAINING_VIT_DEPTH=12,
                     TRAINING_VIT_HEADS=12, TRAINING_VIT_MLP_DIM=3072),
        "large": dict(TRAINING_VIT_INPUT_SIZE=128, TRAINING_VIT_PATCH_SIZE=8, TRAINING_VIT_DIM=1024, TRAINING_VIT_DEPTH=24,
                      TRAINING_VIT_HEADS=16, TRAINING_VIT_MLP_DIM=4096),
        # what the reference ships: configs/config.yaml:39-40 (90^3, patch 9) with the constants of NeuroEncoder.py:187-190
        "reference": dict(TRAINING_VIT_INPUT_SIZE=90, TRAINING_VIT_PATCH_SIZE=9, TRAINING_VIT_DIM=1024, TRAINING_VIT_DEPTH=6,
                          TRAINING_VIT_HEADS=8, TRAINING_VIT_MLP_DIM=2048),
    }
    return dict(presets[name])
