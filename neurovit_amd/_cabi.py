"""ctypes binding of libneurovit_hip.so (the gfx950 C-ABI, include/neurovit_hip.h).

The prototypes are parsed from the header itself so the binding cannot drift from it.
There is NO fallback: if the library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "neurovit_hip.h")
# NEUROVIT_HIP_LIB: another build of the same library (same-box A/B runs of two builds in one gpurun call: tools/_ab/); the tree's own otherwise
LIB_PATH = os.environ.get("NEUROVIT_HIP_LIB") or os.path.join(_HERE, "lib", "libneurovit_hip.so")

_SCALARS = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double}


class VitConfig(ctypes.Structure):
    """struct nv_vit_config (neurovit_hip.h) == ctor arguments of the reference ViT (vit_3d.py:78)."""
    _fields_ = [("image_size", ctypes.c_int), ("image_patch_size", ctypes.c_int), ("frames", ctypes.c_int),
                ("frame_patch_size", ctypes.c_int), ("channels", ctypes.c_int), ("num_classes", ctypes.c_int),
                ("dim", ctypes.c_int), ("depth", ctypes.c_int), ("heads", ctypes.c_int), ("dim_head", ctypes.c_int),
                ("mlp_dim", ctypes.c_int), ("ln_eps", ctypes.c_float), ("pool_mean", ctypes.c_int),
                ("image_width", ctypes.c_int), ("patch_width", ctypes.c_int),      # 0 = square
                ("no_proj_dropout", ctypes.c_int)]      # heads == 1 and dim_head == dim: to_out is nn.Identity(), no Dropout behind it (vit_3d.py:43-46)


class VitInput(ctypes.Structure):
    """struct nv_vit_input (neurovit_hip.h): optional input forms of nv_vit_forward_in / nv_vit_forward_fp8."""
    _fields_ = [("vol_sigma", ctypes.c_void_p), ("time_points", ctypes.c_int), ("rows_form", ctypes.c_int)]


class TrainHparams(ctypes.Structure):
    """struct nv_train_hparams (neurovit_hip.h): optimizer constants and accumulation flags of nv_vit_train_step."""
    _fields_ = [("struct_size", ctypes.c_int), ("step", ctypes.c_int), ("lr", ctypes.c_double), ("beta1", ctypes.c_double),
                ("beta2", ctypes.c_double), ("eps", ctypes.c_double), ("weight_decay", ctypes.c_double), ("grad_scale", ctypes.c_float),
                ("accumulate", ctypes.c_int), ("update", ctypes.c_int), ("fuse_update", ctypes.c_int),
                ("loss_scale", ctypes.c_float), ("dp", ctypes.c_void_p), ("loss_scale_state", ctypes.c_void_p)]


class DpPlan(ctypes.Structure):
    """struct nv_dp_plan (neurovit_hip.h): the data-parallel form of nv_vit_train_step's backward pass + update."""
    _fields_ = [("struct_size", ctypes.c_int), ("world", ctypes.c_int), ("comm", ctypes.c_void_p), ("comm_stream", ctypes.c_void_p),
                ("n_buckets", ctypes.c_int), ("update_per_bucket", ctypes.c_int), ("grads16", ctypes.c_void_p)]


class AdamwArena(ctypes.Structure):
    """struct nv_adamw_arena (neurovit_hip.h): AdamW constants + the five arenas that share element offsets."""
    _fields_ = [("struct_size", ctypes.c_int), ("step", ctypes.c_int), ("lr", ctypes.c_double), ("beta1", ctypes.c_double),
                ("beta2", ctypes.c_double), ("eps", ctypes.c_double), ("weight_decay", ctypes.c_double), ("grad_scale", ctypes.c_float),
                ("keep_grads", ctypes.c_int), ("params", ctypes.c_void_p), ("grads", ctypes.c_void_p), ("adam_m", ctypes.c_void_p),
                ("adam_v", ctypes.c_void_p), ("params16", ctypes.c_void_p)]


ABI_VERSION = 6      # NV_ABI_VERSION of the header this binding was written against (checked at load time)


def parse_header(path: str = HEADER) -> Dict[str, Tuple[object, List[object]]]:
    """{symbol: (restype, [argtypes])} for every `nv_*` prototype declared in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|long|const char\s*\*)\s+(nv_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        restype = ctypes.c_char_p if "char" in ret else _SCALARS[ret]
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_char_p if re.match(r"const char\s*\*", a) else ctypes.c_void_p)
                elif a.startswith("unsigned long"):
                    argtypes.append(ctypes.c_ulong)
                else:
                    argtypes.append(_SCALARS[a.split()[0]])
        protos[name] = (restype, argtypes)
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self.protos = parse_header()

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"neurovit_amd: {LIB_PATH} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(or `make -C neurovit_amd/csrc`). There is no CPU / PyTorch fallback for this path.")
            dll = ctypes.CDLL(LIB_PATH)
            for name, (restype, argtypes) in self.protos.items():
                fn = getattr(dll, name)          # AttributeError if the .so lacks a declared symbol
                fn.restype, fn.argtypes = restype, argtypes
            got = dll.nv_abi_version()
            if got != ABI_VERSION:
                raise RuntimeError(f"neurovit_amd: {LIB_PATH} has ABI revision {got}, this binding expects {ABI_VERSION} - rebuild the library "
                                   "(make -C neurovit_amd/csrc); see INTEGRATION.md 'ABI revisions'")
            if os.environ.get("NEUROVIT_ATTN_MODE"):           # A/B aid: nv_attn_set_mode for the whole process (see the header)
                dll.nv_attn_set_mode(int(os.environ["NEUROVIT_ATTN_MODE"]))
            if os.environ.get("NEUROVIT_ADAMW_WGS"):           # A/B aid: workgroups of the weight-gradient launch with the AdamW epilogue
                dll.nv_gemm_set_tile(12, int(os.environ["NEUROVIT_ADAMW_WGS"]))
            if os.environ.get("NEUROVIT_HEAD_STEP"):           # A/B aid: 0 = the native step runs the head as three calls
                dll.nv_vit_set_head_step(int(os.environ["NEUROVIT_HEAD_STEP"]))
            if os.environ.get("NEUROVIT_WGRAD_WGS"):           # A/B aid: workgroups of the grouped weight-gradient launch
                dll.nv_gemm_set_tile(14, int(os.environ["NEUROVIT_WGRAD_WGS"]))
            if os.environ.get("NEUROVIT_PATCH_MODE"):          # A/B aid: 2 = the LDS-staged slab form of the patch gather instead of the per-token one
                dll.nv_patch_set_mode(int(os.environ["NEUROVIT_PATCH_MODE"]))
            if os.environ.get("NEUROVIT_ADAMW_CAP"):
                dll.nv_gemm_set_tile(13, int(os.environ["NEUROVIT_ADAMW_CAP"]))
            self._dll = dll
        return self._dll

    def __getattr__(self, name):
        return getattr(self.load(), name)


lib = _Lib()


OPERAND_FORMATS = {"bf16": 0, "fp16": 1}      # NV_OPERAND_BF16 / NV_OPERAND_FP16
_operand_format = "bf16"                      # what the library is set to (its own default)


def set_operand_format(fmt: str) -> None:
    """Select what every 16-bit buffer of the following calls holds and which MFMA contracts it (nv_set_operand_format): "bf16" (default)
    or "fp16" - the reference's autocast arithmetic (src/Trainer.py:68).  Process-wide in the library; modules set their own format
    in front of their calls, so models of both formats can live in one process.  Costs a C call only when the format changes."""
    global _operand_format
    if fmt != _operand_format:
        if fmt not in OPERAND_FORMATS:
            raise ValueError(f"neurovit_amd: operand format must be 'bf16' or 'fp16', got {fmt!r}")
        check(lib.nv_set_operand_format(OPERAND_FORMATS[fmt]), "nv_set_operand_format")
        _operand_format = fmt


def operand_format() -> str:
    return _operand_format


def last_error() -> str:
    return (lib.nv_last_error() or b"").decode()


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise RuntimeError(f"neurovit_amd C-ABI call failed ({what}, rc={rc}): {last_error()}")


def require_gpu() -> None:
    """Fail loudly unless a gfx950 device and the HIP library are usable."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("neurovit_amd: no HIP device available - this path only runs on MI355X (gfx950); "
                           "there is no CPU fallback.")
    if lib.nv_arch_ok() != 1:
        raise RuntimeError(f"neurovit_amd: current device is not gfx950: {last_error()}")
