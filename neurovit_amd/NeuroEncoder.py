"""Drop-in for the reference's src/models/NeuroEncoder.py on MI355X.

Same classes / constructor signatures / forward signatures / state_dict keys / attribute paths
(NeuroEncoder.py:15-230): NeuroEncoder(config), ViT3DEncoder(config), TemporalTransformer(config),
ProjectionHead(config); `model.activations`, `model.gradients`, `get_attention_map`, `visualize_slice`.

Config: the reference hard-codes dim=1024, depth=6, heads=8, mlp_dim=2048 (NeuroEncoder.py:181-195).
Optional keys TRAINING_VIT_DIM / _DEPTH / _HEADS / _DIM_HEAD / _MLP_DIM override them and default to
those constants, so an unmodified configs/config.yaml builds the same model.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .vit_3d import ViT


class NeuroEncoder(nn.Module):
    """3D or 4D encoder for MRI / fMRI volumes (NeuroEncoder.py:15-68)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.device = config['DEVICE']

        self.volume_encoder = ViT3DEncoder(config)

        if config['TRAINING_DIM'] == 4:
            # Extract only ViT3D weights by filtering keys (NeuroEncoder.py:23-36)
            best_model_path = os.path.join(config['GLOBAL_BASE_PATH'], config['BEST_MODEL_PATH'])
            full_state_dict = torch.load(best_model_path, map_location='cpu', weights_only=True)
            vit3d_state_dict = {
                k.replace("volume_encoder.vit3d.", "vit3d."): v
                for k, v in full_state_dict.items()
                if k.startswith("volume_encoder.vit3d.")
            }
            self.volume_encoder.load_state_dict(vit3d_state_dict, strict=True)

            for param in self.volume_encoder.parameters():
                param.requires_grad = False
            self.volume_encoder.eval()

            self.temporal_transformer = TemporalTransformer(config)
            self.projection_head = ProjectionHead(config)

        self.to(self.device)  # Move entire model to device

        self.register_hooks()

    def forward(self, fmri):
        if self.config['TRAINING_DIM'] == 3:
            fmri_encoding = self.volume_encoder(fmri)
        elif self.config['TRAINING_DIM'] == 4:
            fmri = fmri.to(self.device).permute(0, 4, 1, 2, 3)   # [B, H, W, D, T] -> [B, T, H, W, D]
            B, T, H, W, D = fmri.shape
            volumes = fmri.reshape(B * T, H, W, D)
            volumes_encoding = self.volume_encoder(volumes)       # [B*T, 2]
            volumes_encoding = volumes_encoding.reshape(B, T, -1)

            fmri_encodings = self.temporal_transformer(volumes_encoding)
            fmri_encoding = fmri_encodings.mean(dim=1)
            fmri_encoding = self.projection_head(fmri_encoding)   # [B, 2]

        return fmri_encoding

    # ---- Grad-CAM contract (NeuroEncoder.py:70-82): activation / gradient of the last block's attention-LN output.
    # The reference copies both to the CPU on EVERY forward / backward (a blocking D2H sync per step); here they stay
    # in the engine workspace and are materialised on attribute access only.
    def register_hooks(self):
        self._hook_override = {}

    @property
    def activations(self):
        if 'activations' in self._hook_override:
            return self._hook_override['activations']
        vit = self.volume_encoder.vit3d
        if vit._rt._last is None:
            return {}
        return vit.last_attn_norm_output().detach().cpu()

    @activations.setter
    def activations(self, value):
        self._hook_override['activations'] = value

    @property
    def gradients(self):
        if 'gradients' in self._hook_override:
            return self._hook_override['gradients']
        vit = self.volume_encoder.vit3d
        if vit._rt._last is None or not vit._rt._last[1]:
            return {}
        return vit.last_attn_norm_grad().detach().cpu()

    @gradients.setter
    def gradients(self, value):
        self._hook_override['gradients'] = value

    def get_attention_map(self, x):
        """NeuroEncoder.py:84-133."""
        grid_size = self.config['TRAINING_VIT_INPUT_SIZE']
        patch_size = self.config['TRAINING_VIT_PATCH_SIZE']
        threshold = self.config['GRADCAM_THRESHOLD']

        output = self.forward(x)
        class_idx = output.argmax(dim=1)

        one_hot = torch.zeros_like(output)
        one_hot[torch.arange(output.size(0)), class_idx] = 1

        output.backward(gradient=one_hot, retain_graph=True)
        gradients = self.gradients
        activations = self.activations

        weights = gradients.mean(dim=2, keepdim=True)
        cam = (weights * activations).sum(dim=2)
        cam = cam[:, 1:]

        cam_size = grid_size // patch_size
        cam = cam.reshape(1, cam_size, cam_size, cam_size)

        cam = F.relu(cam)
        cam = (cam - cam.min()) / (cam.max() - cam.min() + 1e-8)
        threshold_value = np.percentile(cam, 100 - threshold)
        thresholded_map = np.where(cam >= threshold_value, cam, 0)
        thresholded_map = torch.from_numpy(thresholded_map).unsqueeze(0)

        cam_3d = F.interpolate(
            thresholded_map,
            size=(grid_size, grid_size, grid_size),
            mode='trilinear',
            align_corners=False
        ).squeeze()

        return cam_3d, class_idx

    def visualize_slice(self, cam_3d, original_volume):
        """NeuroEncoder.py:135-168."""
        slice_dim = self.config['GRADCAM_SLICE_DIM']
        slice_idx = self.config['GRADCAM_SLICE_IDX']

        if cam_3d is None:
            print("Error: No CAM computed")
            return

        original = original_volume.squeeze()
        original = original.detach().cpu().numpy()

        if original.ndim != 3 or cam_3d.ndim != 3:
            print(f"Shape mismatch: original {original.shape}, CAM {cam_3d.shape}")
            return

        if slice_dim == 0:
            img = original[slice_idx]
            attn = cam_3d[slice_idx]
        elif slice_dim == 1:
            img = original[:, slice_idx]
            attn = cam_3d[:, slice_idx]
        elif slice_dim == 2:
            img = original[:, :, slice_idx]
            attn = cam_3d[:, :, slice_idx]
        else:
            print(f"Invalid slice dimension: {slice_dim}")
            return

        return img, attn


class ViT3DEncoder(nn.Module):
    """NeuroEncoder.py:171-205: config -> ViT, and the [B,H,W,D] -> [B,1,D,H,W] input view."""

    def __init__(self, config):
        super().__init__()
        self.device = config['DEVICE']
        self.dropout = config['TRAINING_DROPOUT']
        self.grid_size = config['TRAINING_VIT_INPUT_SIZE']
        self.cube_size = config['GRADCAM_CUBE_SIZE']
        self.patch_size = config['TRAINING_VIT_PATCH_SIZE']
        number_classes = (self.grid_size // self.cube_size) ** 3 if config['DATASET_NAME'] == 'gradcam' else 2

        self.vit3d = ViT(
            channels=1,
            image_size=self.grid_size,
            image_patch_size=self.patch_size,
            frames=self.grid_size,
            frame_patch_size=self.patch_size,
            num_classes=number_classes,
            dim=config.get('TRAINING_VIT_DIM', 1024),
            depth=config.get('TRAINING_VIT_DEPTH', 6),
            heads=config.get('TRAINING_VIT_HEADS', 8),
            dim_head=config.get('TRAINING_VIT_DIM_HEAD', 64),
            mlp_dim=config.get('TRAINING_VIT_MLP_DIM', 2048),
            dropout=self.dropout,
            emb_dropout=self.dropout,
            pool='cls'
        ).to(self.device)

    def forward(self, x):
        # x: (batch, H, W, D).  The permuted tensor is only a VIEW: the patch-gather kernel reads the original
        # [B,H,W,D] memory through its strides, so the reference's permute never costs a copy.
        timepoint = x.to(self.device)
        timepoint = timepoint.permute(0, 3, 1, 2)
        timepoint = timepoint.unsqueeze(1)
        return self.vit3d(timepoint)


class TemporalTransformer(nn.Module):
    """NeuroEncoder.py:207-217.  d_model = 2 (the frozen ViT3D emits 2 logits): 10 274 parameters and ~0 % of
    the FLOPs, so it stays on stock torch modules (which also keeps the reference's state_dict keys)."""

    def __init__(self, config):
        super().__init__()
        self.device = config['DEVICE']
        encoder_layer = nn.TransformerEncoderLayer(d_model=2, nhead=2, batch_first=True)
        self.transformer = nn.TransformerEncoder(encoder_layer, num_layers=1).to(self.device)

    def forward(self, x):
        return self.transformer(x)


class ProjectionHead(nn.Module):
    """NeuroEncoder.py:219-230."""

    def __init__(self, config):
        super().__init__()
        self.device = config['DEVICE']
        self.projection_head = nn.Linear(2, 2).to(self.device)

    def forward(self, x):
        return self.projection_head(x)
