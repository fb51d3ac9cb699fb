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

import torch
import torch.nn as nn
import torch.nn.functional as F

from .vit_3d import ViT


def _encoder_weights_of(checkpoint: dict, prefix: str = "volume_encoder.") -> dict:
    """The entries of a 3D NeuroEncoder checkpoint that belong to its ViT3DEncoder, re-keyed for that sub-module: the 4D model is
    built around the encoder of a trained 3D model (NeuroEncoder.py:23-31) and nothing else of that file is used."""
    wanted = prefix + "vit3d."
    return {key[len(prefix):]: tensor for key, tensor in checkpoint.items() if key.startswith(wanted)}


class NeuroEncoder(nn.Module):
    """3D or 4D encoder for MRI / fMRI volumes (NeuroEncoder.py:15-68).  Attribute names and the ORDER in which sub-modules are created
    are the contract (state_dict keys; the RNG stream of a seeded construction) - tests/test_boundary_cpu.py pins both."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.device = config['DEVICE']
        self.volume_encoder = ViT3DEncoder(config)
        four_d = config['TRAINING_DIM'] == 4
        if four_d:
            self._adopt_trained_encoder(os.path.join(config['GLOBAL_BASE_PATH'], config['BEST_MODEL_PATH']))
            self.temporal_transformer = TemporalTransformer(config)
            self.projection_head = ProjectionHead(config)
            # the two modules above hold the parameters (reference classes, keys and initialisation); the computation
            # transformer -> mean over time -> projection is one native launch per direction (temporal.py, csrc/temporal.hip)
            from .temporal import TemporalHead
            self._temporal_head = TemporalHead(self.temporal_transformer, self.projection_head)
        self.to(self.device)
        self.register_hooks()

    def _adopt_trained_encoder(self, checkpoint_path: str) -> None:
        """4D model (NeuroEncoder.py:23-36): the spatial encoder is the ViT3D of a trained 3D checkpoint - loaded strictly, frozen and
        kept in eval mode; only the temporal head trains.  The file is read with weights_only=True (a plain state_dict, Trainer.py:54-55)."""
        saved = torch.load(checkpoint_path, map_location='cpu', weights_only=True)
        self.volume_encoder.load_state_dict(_encoder_weights_of(saved), strict=True)
        self.volume_encoder.requires_grad_(False)
        self.volume_encoder.eval()

    def forward(self, fmri):
        if self.config['TRAINING_DIM'] == 3:
            return self.volume_encoder(fmri)                          # NeuroEncoder.py:50-51
        if self.config['TRAINING_DIM'] != 4:
            raise ValueError(f"TRAINING_DIM must be 3 or 4, got {self.config['TRAINING_DIM']!r}")
        # 4D (NeuroEncoder.py:53-66): every timepoint is an independent volume for the frozen ViT3D, so the time axis is
        # folded into the batch; the B*T logit pairs then form a length-T sequence for the temporal transformer.
        series = fmri.to(self.device)
        n_samples, n_time = series.shape[0], series.shape[-1]
        vit = self.volume_encoder.vit3d
        frozen = not (torch.is_grad_enabled() and any(p.requires_grad for p in vit.parameters()))
        if frozen and n_time % 4 == 0 and n_time <= 64 and series.dtype == torch.float32 and series.is_contiguous():
            # fused 4D gather (csrc/norm.hip::patch_ln_fwd_t_kernel): the B*T volumes are read in place, no regroup copy
            per_volume = vit(series, time_points=n_time).unflatten(0, (n_samples, n_time))
        else:
            as_volumes = series.movedim(-1, 1).flatten(0, 1)          # [B, H, W, D, T] -> [B*T, H, W, D]  (strided copy)
            per_volume = self.volume_encoder(as_volumes).unflatten(0, (n_samples, n_time))
        head = self._temporal_head
        if head.supported(per_volume):                                # NeuroEncoder.py:60-66 in one launch
            return head(per_volume)
        # geometries outside the kernel's (more than 64 timepoints, an edited encoder layer): the stock modules, on the device
        pooled = self.temporal_transformer(per_volume).mean(dim=1)
        return self.projection_head(pooled)

    def precision(self, mode: str):
        """Context manager: eval-mode no-grad forwards of the ViT3D encoder inside it run in `mode` ("bf16" / "fp16": the 16-bit
        operand path in the encoder's operand format, or "fp32")."""
        return self.volume_encoder.vit3d.precision(mode)

    def set_operands(self, fmt: str):
        """16-bit MFMA operand format of the ViT3D encoder: "bf16" (default) or "fp16" (the reference's autocast arithmetic)."""
        self.volume_encoder.vit3d.set_operands(fmt)
        return self

    def forward_raw(self, raw, crop=None):
        """3D model fed RAW scanner volumes [B, X, Y, Z] float32 on the device: the dataset's crop (DatasetADNI.py:212) is a strided
        view and its z-score (:213) is folded into the patch LayerNorm - one statistics pass, no normalised copy."""
        if self.config['TRAINING_DIM'] != 3:
            raise NotImplementedError("forward_raw: 3D model only (4D samples are normalised over all timepoints: DatasetADNI_4D.py:86-87)")
        return self.volume_encoder.forward_raw(raw, crop)

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
        # the buffer holds the gradient only once a backward pass of the most recent training-mode forward has run
        if vit._rt._last is None or not vit._rt._last[1] or not vit._rt.backward_done:
            return {}
        return vit.last_attn_norm_grad().detach().cpu()

    @gradients.setter
    def gradients(self, value):
        self._hook_override['gradients'] = value

    def get_attention_map(self, x):
        """Grad-CAM of the predicted class on the patch grid, thresholded and upsampled to the volume
        (same contract as NeuroEncoder.py:84-133: returns (cam[S,S,S] on the CPU, class_idx)).

        The [B,n,d] activation and gradient never leave the device: `nv_gradcam_reduce` (csrc/gradcam.hip) turns them
        into the normalised G^3 map in one launch, and only those G^3 floats cross PCIe for the percentile / upsampling."""
        from . import ops
        size = self.config['TRAINING_VIT_INPUT_SIZE']
        cells = size // self.config['TRAINING_VIT_PATCH_SIZE']
        keep_percent = self.config['GRADCAM_THRESHOLD']

        logits = self.forward(x)
        predicted = logits.argmax(dim=1)
        logits.backward(gradient=F.one_hot(predicted, logits.shape[1]).to(logits.dtype), retain_graph=True)

        vit = self.volume_encoder.vit3d
        if 'activations' in self._hook_override or 'gradients' in self._hook_override:      # user-assigned hook tensors win
            act, grad = self.activations.float(), self.gradients.float()
            token_map = torch.relu(grad.mean(dim=2) * act.sum(dim=2))[:, 1:]
            token_map = (token_map - token_map.min()) / (token_map.max() - token_map.min() + 1e-8)
        else:
            token_map, _ = ops.gradcam_reduce(vit.last_attn_norm_output_raw(), vit.last_attn_norm_grad_raw())
            token_map = token_map.cpu()
        grid = token_map.reshape(1, cells, cells, cells)               # one sample, as in the reference

        # keep the top `keep_percent` % of the cells (linear-interpolated percentile, as numpy.percentile), zero the rest
        cut = torch.quantile(grid.double().flatten(), 1.0 - keep_percent / 100.0).to(grid.dtype)
        sparse = torch.where(grid >= cut, grid, torch.zeros_like(grid))
        volume = F.interpolate(sparse[None], size=(size,) * 3, mode='trilinear', align_corners=False)
        return volume[0, 0], predicted

    def visualize_slice(self, cam_3d, original_volume):
        """One 2-D slice of the volume and of its CAM along GRADCAM_SLICE_DIM at GRADCAM_SLICE_IDX
        (contract of NeuroEncoder.py:135-168: returns (img, attn), or None after printing why not)."""
        axis, index = self.config['GRADCAM_SLICE_DIM'], self.config['GRADCAM_SLICE_IDX']
        if cam_3d is None:
            print("Error: No CAM computed")
            return None
        volume = original_volume.squeeze().detach().cpu().numpy()
        if volume.ndim != 3 or cam_3d.ndim != 3:
            print(f"Shape mismatch: original {volume.shape}, CAM {cam_3d.shape}")
            return None
        if axis not in (0, 1, 2):
            print(f"Invalid slice dimension: {axis}")
            return None
        pick = tuple(index if a == axis else slice(None) for a in range(3))
        return volume[pick], cam_3d[pick]


class ViT3DEncoder(nn.Module):
    """NeuroEncoder.py:171-205: config -> ViT, and the [B,H,W,D] -> [B,1,D,H,W] input view."""

    # the transformer size the reference hard-codes (NeuroEncoder.py:187-190); optional TRAINING_VIT_* keys override it
    REFERENCE_SIZE = {'TRAINING_VIT_DIM': 1024, 'TRAINING_VIT_DEPTH': 6, 'TRAINING_VIT_HEADS': 8, 'TRAINING_VIT_DIM_HEAD': 64,
                      'TRAINING_VIT_MLP_DIM': 2048}

    def __init__(self, config):
        super().__init__()
        self.device = config['DEVICE']
        self.dropout = config['TRAINING_DROPOUT']
        self.grid_size = config['TRAINING_VIT_INPUT_SIZE']
        self.cube_size = config['GRADCAM_CUBE_SIZE']
        self.patch_size = config['TRAINING_VIT_PATCH_SIZE']
        size = {key: config.get(key, default) for key, default in self.REFERENCE_SIZE.items()}
        # the synthetic cube-localisation task has one class per cube position (NeuroEncoder.py:179); everything else is binary
        cubes_per_axis = self.grid_size // self.cube_size
        classes = cubes_per_axis ** 3 if config['DATASET_NAME'] == 'gradcam' else 2
        # a cubic single-channel volume: frames (depth) and image sides share one extent and one patch edge
        self.vit3d = ViT(image_size=self.grid_size, image_patch_size=self.patch_size, frames=self.grid_size, frame_patch_size=self.patch_size,
                         channels=1, num_classes=classes, dim=size['TRAINING_VIT_DIM'], depth=size['TRAINING_VIT_DEPTH'],
                         heads=size['TRAINING_VIT_HEADS'], dim_head=size['TRAINING_VIT_DIM_HEAD'], mlp_dim=size['TRAINING_VIT_MLP_DIM'],
                         pool='cls', dropout=self.dropout, emb_dropout=self.dropout).to(self.device)
        # arithmetic of eval-mode no-grad forwards: "bf16" (default; the training arithmetic) or "fp32" (the reference's validate,
        # Trainer.py:101-118).  The Trainer shell's validate / evaluate_samples use VALIDATION_PRECISION (default "fp32").
        self.vit3d.eval_precision = config.get('TRAINING_VIT_EVAL_PRECISION', 'bf16')
        # 16-bit MFMA operand format of the training arithmetic: "bf16" (default) or "fp16" - what the reference's autocast(float16)
        # computes in (Trainer.py:68); TrainStep then scales the loss as its GradScaler does (Trainer.py:29,74-76)
        self.vit3d.set_operands(config.get('TRAINING_VIT_OPERANDS', 'bf16'))

    def forward_raw(self, raw, crop=None):
        from .preprocess import ADNI_CROP, crop_view, volume_sigma
        crop = ADNI_CROP if crop is None else crop
        raw = raw.to(self.device)
        if raw.dtype != torch.float32:                                 # int16 scanner data: the gather kernel reads float32 -> separate pass
            from .preprocess import zscore_crop
            return self.forward(zscore_crop(raw, crop))
        sigma = volume_sigma(raw, crop)                                # std + 1e-8 per cropped volume (statistics in double)
        view = crop_view(raw, crop)                                    # [B, Sx, Sy, Sz] strided view of the raw tensor
        return self.vit3d(view.permute(0, 3, 1, 2).unsqueeze(1), vol_sigma=sigma)

    def forward(self, x):
        # x: (batch, H, W, D).  The permuted tensor is only a VIEW: the patch-gather kernel reads the original
        # [B,H,W,D] memory through its strides, so the reference's permute never costs a copy.
        volume = x.to(self.device)
        return self.vit3d(volume.permute(0, 3, 1, 2).unsqueeze(1))      # (batch, channel = 1, frames = D, height = H, width = W)


class TemporalTransformer(nn.Module):
    """NeuroEncoder.py:207-217.  d_model = 2 (the frozen ViT3D emits 2 logits): 10 274 parameters.  The stock module is the
    parameter holder (reference state_dict keys and initialisation); NeuroEncoder.forward computes it through temporal.TemporalHead."""

    D_MODEL, HEADS, LAYERS = 2, 2, 1      # NeuroEncoder.py:210-211: the sequence elements are the encoder's two logits

    def __init__(self, config):
        super().__init__()
        self.device = config['DEVICE']
        block = nn.TransformerEncoderLayer(d_model=self.D_MODEL, nhead=self.HEADS, batch_first=True)
        self.transformer = nn.TransformerEncoder(block, num_layers=self.LAYERS).to(self.device)

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
