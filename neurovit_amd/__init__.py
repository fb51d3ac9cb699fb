"""neurovit_amd - MI355X (gfx950) native ViT3D / NeuroEncoder hot path of gillet-thomas/NeuroViT.

Drop-in nn.Modules (same constructors, forward signatures, state_dict keys and config.yaml keys as the
reference's src/models/vit_3d.py and src/models/NeuroEncoder.py) over hand-written HIP kernels reached
through a C-ABI shared library (include/neurovit_hip.h).  There is no CPU / eager fallback.
"""
__version__ = "0.1.0"
