"""Host-side / boundary tests that need no GPU: state_dict contract, same-seed init parity with the reference,
config handling, the C-ABI library (loads, exports every declared symbol, host-only entry points), and that
the product path refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os

import numpy as np
import pytest
import torch

import weights as W
from conftest import ROOT


def test_cabi_exports_every_declared_symbol():
    from neurovit_amd._cabi import LIB_PATH, parse_header
    assert os.path.exists(LIB_PATH), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    protos = parse_header()
    assert len(protos) >= 30
    dll = ctypes.CDLL(LIB_PATH)
    missing = [name for name in protos if not hasattr(dll, name)]
    assert not missing, missing
    assert dll.nv_version() >= 1


def test_param_table_matches_reference_state_dict_order():
    from neurovit_amd import engine
    for cfgdict in (W.MICRO, W.TINY, W.BASE):
        spec = W.vit_param_spec(**cfgdict)
        off, num, total = engine.param_layout(engine.make_config(**cfgdict))
        assert len(off) == len(spec)
        assert [int(np.prod(s)) for _, s, _ in spec] == num
        assert all(o % 8 == 0 for o in off) and total % 8 == 0
        assert all(o2 >= o1 + n1 for o1, n1, o2 in zip(off, num, off[1:]))     # no overlap, state_dict order
    # parameter count of ViT3D-base matches SURVEY 8d (88.58 M)
    _, num, _ = engine.param_layout(engine.make_config(**W.BASE))
    assert abs(sum(num) - 88.58e6) < 0.01e6


def test_workspace_and_error_paths():
    from neurovit_amd import engine
    from neurovit_amd._cabi import last_error, lib
    cfg = engine.make_config(**W.BASE)
    inf, trn = (lib.nv_vit_workspace_bytes(ctypes.byref(cfg), 4, t) for t in (0, 1))
    assert 0 < inf < trn
    bad = engine.make_config(**dict(W.BASE, dim_head=36))
    assert lib.nv_vit_workspace_bytes(ctypes.byref(bad), 4, 1) < 0 and "dim_head" in last_error()
    bad = engine.make_config(**dict(W.BASE, image_size=100))
    assert lib.nv_vit_param_count(ctypes.byref(bad)) < 0 and "divisible" in last_error()
    b, e = ctypes.c_long(), ctypes.c_long()
    total = lib.nv_vit_param_count(ctypes.byref(cfg))
    covered = []
    for s in range(cfg.depth + 2):
        assert lib.nv_vit_stage_param_range(ctypes.byref(cfg), s, ctypes.byref(b), ctypes.byref(e)) == 0
        covered.append((b.value, e.value))
    covered.sort()
    assert covered[0][0] == 0 and covered[-1][1] == total
    assert all(a[1] == c[0] for a, c in zip(covered, covered[1:]))             # stages tile the arena exactly


def test_state_dict_keys_and_shapes_match_reference(golden):
    from neurovit_amd.NeuroEncoder import NeuroEncoder
    g = golden("neuro3d.npz")
    m = NeuroEncoder(W.neuro_config(32, 8))
    sd = m.state_dict()
    assert list(sd.keys()) == list(g["keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["shapes"])
    g4 = golden("neuro4d.npz")
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        torch.save(m.state_dict(), os.path.join(td, "c.pth"))
        # 4D needs the 3D checkpoint at S=16 (fixture shape)
        m3 = NeuroEncoder(W.neuro_config(16, 8))
        torch.save(m3.state_dict(), os.path.join(td, "c16.pth"))
        m4 = NeuroEncoder(W.neuro_config(16, 8, dim=4, GLOBAL_BASE_PATH=td, BEST_MODEL_PATH="c16.pth"))
    assert list(m4.state_dict().keys()) == list(g4["keys"])
    assert [str(tuple(v.shape)) for v in m4.state_dict().values()] == list(g4["shapes"])
    assert sorted(k for k, p in m4.named_parameters() if p.requires_grad) == sorted(g4["trainable"])
    assert not m4.volume_encoder.training                                          # NeuroEncoder.py:36
    for k in m3.state_dict():                                                      # strict load of the filtered 3D ckpt
        assert torch.equal(m3.state_dict()[k], m4.state_dict()[k])


def test_same_seed_gives_reference_init(golden):
    """torch.manual_seed(42) (main.py:86-88) -> bit-identical initial parameters to the reference's modules."""
    from neurovit_amd.NeuroEncoder import NeuroEncoder
    from neurovit_amd.vit_3d import ViT
    g = golden("init.npz")
    torch.manual_seed(42)
    m = ViT(**W.MICRO)
    for k, v in W.checksums(m.state_dict()).items():
        np.testing.assert_array_equal(v, g["init." + k], err_msg=k)
    torch.manual_seed(42)
    n = NeuroEncoder(W.neuro_config(16, 8))
    for k, v in W.checksums(n.state_dict()).items():
        np.testing.assert_array_equal(v, g["init_neuro." + k], err_msg=k)


def test_optional_size_keys_default_to_reference_constants():
    from neurovit_amd.NeuroEncoder import NeuroEncoder
    m = NeuroEncoder(W.neuro_config(16, 8))
    vit = m.volume_encoder.vit3d
    assert vit.pos_embedding.shape == (1, 9, 1024) and len(vit.transformer.layers) == 6
    assert vit.transformer.layers[0][0].to_qkv.weight.shape == (3 * 8 * 64, 1024)
    assert vit.transformer.layers[0][1].net[1].weight.shape == (2048, 1024)
    t = NeuroEncoder(W.neuro_config(64, 16, TRAINING_VIT_DIM=192, TRAINING_VIT_DEPTH=4, TRAINING_VIT_HEADS=3, TRAINING_VIT_MLP_DIM=384))
    assert abs(sum(p.numel() for p in t.parameters()) - 1.994e6) < 1e3                 # tiny = 1.994 M
    gc = NeuroEncoder(W.neuro_config(32, 8, dataset="gradcam"))
    assert gc.volume_encoder.vit3d.mlp_head[1].weight.shape[0] == (32 // 8) ** 3      # NeuroEncoder.py:179


def test_reference_yaml_configs_load(tmp_path):
    import yaml
    from neurovit_amd import config as C
    y = dict(TRAINING_DIM=3, TRAINING_DROPOUT=0.1, TRAINING_VIT_INPUT_SIZE=90, TRAINING_VIT_PATCH_SIZE=9, GRADCAM_CUBE_SIZE=8,
             DATASET_NAME="adni", TRAINING_LEARNING_RATE=1e-4)
    p = tmp_path / "config.yaml"
    p.write_text(yaml.safe_dump(y))
    cfg = C.load_config(str(p), device="cpu")
    assert cfg["TRAINING_VIT_DIM"] == 1024 and cfg["TRAINING_VIT_MLP_DIM"] == 2048 and cfg["DEVICE"] == "cpu"
    p.write_text(yaml.safe_dump({k: v for k, v in y.items() if k != "DATASET_NAME"}))
    with pytest.raises(KeyError):
        C.load_config(str(p))
    assert C.preset("base")["TRAINING_VIT_MLP_DIM"] == 3072


def test_ctor_asserts_like_reference():
    from neurovit_amd.vit_3d import ViT
    with pytest.raises(AssertionError, match="divisible"):
        ViT(**dict(W.MICRO, image_size=30))
    with pytest.raises(AssertionError, match="Frames"):
        ViT(**dict(W.MICRO, frames=30))
    with pytest.raises(AssertionError, match="pool"):
        ViT(**dict(W.MICRO, pool="max"))


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU - never route through the oracle or eager torch."""
    from neurovit_amd.NeuroEncoder import NeuroEncoder
    m = NeuroEncoder(W.neuro_config(16, 8, TRAINING_VIT_DIM=64, TRAINING_VIT_DEPTH=1, TRAINING_VIT_HEADS=2, TRAINING_VIT_MLP_DIM=64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 16, 16, 16))
    import neurovit_amd
    src = "".join(open(os.path.join(os.path.dirname(neurovit_amd.__file__), f)).read()
                  for f in os.listdir(os.path.dirname(neurovit_amd.__file__)) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src


def test_constructions_off_the_neuroencoder_path():
    """Geometries the NeuroEncoder never builds: no output projection (same keys as the reference, constants in the arena), mean
    pooling, the reference's divisibility assert."""
    from neurovit_amd.vit_3d import ViT
    m = ViT(**dict(W.MICRO, dim=64, heads=1))            # to_out becomes Identity in the reference (vit_3d.py:32)
    assert m._no_proj and not any("to_out" in k for k in m.state_dict())
    assert set(m.state_dict()) == set(W.make_tensors(W.vit_param_spec(**dict(W.MICRO, dim=64, heads=1)), 0))
    arena, _ = m.flat_parameters()                        # (the table itself comes from the library: no GPU needed)
    assert len(m._phantom) == 2 * W.MICRO["depth"] and torch.equal(arena[m._phantom[0][0]:m._phantom[0][0] + 64 * 64].view(64, 64), torch.eye(64))
    assert m._arena_ok() and all(q.data_ptr() == arena.data_ptr() + 4 * o for q, o in zip(m._plist, m._layout[0]))
    assert ViT(**dict(W.MICRO, pool="mean"))._cfg.pool_mean == 1 and ViT(**W.MICRO)._cfg.pool_mean == 0
    with pytest.raises(AssertionError):
        ViT(**dict(W.MICRO, image_size=30))              # vit_3d.py:83 divisibility assert


def test_arena_views_survive_load_state_dict_and_track_to():
    from neurovit_amd.vit_3d import ViT
    m = ViT(**W.MICRO)
    arena, shadow = m.flat_parameters()
    sd = W.make_tensors(W.vit_param_spec(**W.MICRO), 1)
    m.load_state_dict(sd)
    assert m._arena_ok() and m.flat_parameters()[0] is arena
    off, num, _ = m._layout
    for (k, v), o, n in zip(sd.items(), off, num):
        assert torch.equal(arena[o:o + n], v.reshape(-1)), k
    m.double().float()                                    # storage replaced behind our back
    assert not m._arena_ok()
    a2, _ = m.flat_parameters()
    assert m._arena_ok() and torch.equal(a2, arena)


def test_device_prefetcher_preserves_order_and_passthrough():
    """Host-side logic of the Trainer's prefetcher (CPU device: plain iteration): every batch, in order, tensors intact,
    non-tensor fields untouched, empty loaders and single-batch loaders handled."""
    from neurovit_amd.trainer import DevicePrefetcher
    batches = [("sub%d" % i, torch.full((2, 3), float(i)), torch.tensor([i, i + 1])) for i in range(5)]
    out = list(DevicePrefetcher(batches, "cpu"))
    assert len(out) == 5 and len(DevicePrefetcher(batches, "cpu")) == 5
    for i, b in enumerate(out):
        assert b[0] == "sub%d" % i and torch.equal(b[1], batches[i][1]) and torch.equal(b[2], batches[i][2])
    assert list(DevicePrefetcher([], "cpu")) == []
    assert len(list(DevicePrefetcher(batches[:1], "cpu"))) == 1


def test_nv_vit_forward_rejects_wrong_extents_before_touching_memory():
    """the gather kernel takes its extents from the config, so a volume of another size must be refused
    (the reference raises in einops / the pos_embedding add, vit_3d.py:92,118).  Host-side check: runs without a GPU."""
    from neurovit_amd import engine
    from neurovit_amd._cabi import last_error, lib
    cfg = engine.make_config(**W.MICRO)                      # 32^3 volumes, 1 channel
    dummy = (ctypes.c_char * 64)()
    ptr = ctypes.addressof(dummy)
    for shape in ((2, 1, 16, 16, 16), (2, 3, 32, 32, 32), (2, 1, 33, 51, 33), (3, 1, 32, 32, 32)):
        shp = (ctypes.c_long * 5)(*shape)
        strides = (ctypes.c_long * 5)(1, 1, 1, 1, 1)
        rc = lib.nv_vit_forward(ctypes.byref(cfg), 2, ptr, shp, strides, ptr, ptr, ptr, 1 << 40, 0, 0.0, 0.0, 0, ptr, None)
        assert rc == -1, shape
        assert "the model was built for" in last_error()


def test_flops_formula_in_package_equals_survey_numbers():
    """bench.py takes its FLOP count from the package (not from oracle/): SURVEY 8(d) values."""
    from neurovit_amd import engine
    from oracle import ref_cpu
    for cfgdict in (W.MICRO, W.TINY, W.BASE):
        c = {k: v for k, v in cfgdict.items() if k != "pool"}
        assert engine.flops_forward(engine.make_config(**c)) == ref_cpu.flops_forward(ref_cpu.ViTCfg(**cfgdict))
    assert abs(engine.flops_forward(engine.make_config(**{k: v for k, v in W.BASE.items() if k != "pool"})) / 1e9 - 100.07) < 0.01

def test_ctypes_structs_match_the_header_layout(tmp_path):
    """The ctypes mirrors of the header's structs (nv_vit_config, nv_vit_input, nv_gemm_problem, nv_reduce_job, nv_train_hparams, nv_adamw_arena) against what a C
    compiler makes of include/neurovit_hip.h: same size, same offset for every field (the header is plain C: gcc compiles it)."""
    import subprocess
    from neurovit_amd import ops
    from neurovit_amd._cabi import HEADER, AdamwArena, DpPlan, TrainHparams, VitConfig, VitInput
    structs = {"nv_vit_config": VitConfig, "nv_vit_input": VitInput, "nv_gemm_problem": ops.GemmProblem, "nv_reduce_job": ops.ReduceJob,
               "nv_train_hparams": TrainHparams, "nv_adamw_arena": AdamwArena, "nv_dp_plan": DpPlan}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    got = {tuple(l.split()[:2]): int(l.split()[2]) for l in out if l.strip()}
    for cname, cls in structs.items():
        assert got[(cname, "size")] == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)



def test_abi_revision_of_library_header_and_binding_agree():
    """nv_abi_version() of the built library == NV_ABI_VERSION of the header == the revision the ctypes binding was written for
    (a caller built against another revision must refuse the library: INTEGRATION.md 'ABI revisions')."""
    import re
    from neurovit_amd import _cabi
    header = int(re.search(r"#define\s+NV_ABI_VERSION\s+(\d+)", open(_cabi.HEADER).read()).group(1))
    assert header == _cabi.ABI_VERSION
    assert _cabi.lib.nv_abi_version() == header            # no GPU needed: loads the library and asks
