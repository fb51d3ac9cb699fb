"""Whole-encoder parity on MI355X: native engine (C-ABI nv_vit_forward / nv_vit_backward) vs the oracle.

G3 (gate): HIP bf16 path vs the bf16-emulating oracle, every stage + logits <= 1e-3 rel, gradients <= GRAD_REL.
G4 (report + loose gate): HIP bf16 logits vs the fp32 golden logits produced by the imported reference.
Measured errors are appended to gpurun_out/parity_report.txt when that directory exists.
"""
import os

import numpy as np
import pytest
import torch

import weights as W
from conftest import ROOT, rel_err
from oracle import ref_cpu, train_step

pytestmark = pytest.mark.gpu
REL = 1e-3        # forward stages / logits vs bf16-emulating oracle
GRAD_REL = 5e-3   # parameter gradients vs bf16-emulating oracle (bf16 operand rounding flips accumulate over depth)


def report(line):
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_report.txt"), "a") as f:
            f.write(line + "\n")


@pytest.fixture(scope="module")
def eng():
    from neurovit_amd import engine
    from neurovit_amd._cabi import require_gpu
    require_gpu()
    return engine


def load_arena(engine, cfgdict, sd):
    cfg = engine.make_config(**cfgdict)
    off, num, total = engine.param_layout(cfg)
    assert len(off) == len(sd)
    arena = torch.zeros(total)
    for (k, v), o, n in zip(sd.items(), off, num):
        assert v.numel() == n, k
        arena[o:o + n] = v.reshape(-1)
    return cfg, off, num, arena


def run_case(engine, tag, cfgdict, seeds, B=2):
    sd = W.make_tensors(W.vit_param_spec(**cfgdict), seeds[0])
    S = cfgdict["image_size"]
    fmri = W.make_volume((B, S, S, S), seeds[1])
    ocfg = ref_cpu.ViTCfg(**cfgdict)
    cfg, off, num, arena = load_arena(engine, cfgdict, sd)
    params = arena.cuda()
    params16 = params.to(torch.bfloat16)
    rt = engine.VitRuntime(cfg)
    video = ref_cpu.fmri_to_video(fmri.cuda())
    logits = rt.forward(video, params, params16, training=True)

    # ---- oracle, bf16-emulating, with autograd for the gradients
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    taps = {}
    ref_logits = ref_cpu.vit_forward(leaves, ocfg, ref_cpu.fmri_to_video(fmri), emulate_bf16=True, taps=taps)
    n, d, M = ocfg.num_patches + 1, ocfg.dim, B * (ocfg.num_patches + 1)
    errs = {"A5": rel_err(rt.tap("x0", -1, (B, n, d), torch.float32), taps["A5"])}
    for i in range(ocfg.depth):
        errs[f"block{i}"] = rel_err(rt.tap("x2", i, (B, n, d), torch.float32), taps[f"block{i}"])
    errs["logits"] = rel_err(logits, ref_logits)
    for k, e in errs.items():
        report(f"{tag} fwd {k}: rel {e:.3e}")
        assert e <= REL, (k, e)

    labels = torch.from_numpy(np.random.RandomState(seeds[1] + 7).randint(0, ocfg.num_classes, size=B)).long()
    loss = train_step.cross_entropy(ref_logits, labels)
    names = list(leaves.keys())
    ref_grads = dict(zip(names, torch.autograd.grad(loss, [leaves[k] for k in names])))
    # dlogits computed by the oracle from the DEVICE logits so the backward comparison is not polluted by forward error
    ld = ref_logits.detach().clone().requires_grad_(True)
    (dlogits,) = torch.autograd.grad(train_step.cross_entropy(ld, labels), ld)
    grads = torch.zeros_like(params)
    rt.backward(dlogits.cuda(), params, params16, grads, accumulate=False)
    gcpu = grads.cpu()
    worst = ("", 0.0)
    for k, o, nn in zip(names, off, num):
        e = rel_err(gcpu[o:o + nn].reshape(ref_grads[k].shape), ref_grads[k])
        report(f"{tag} grad {k}: rel {e:.3e}")
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] <= GRAD_REL, worst
    # accumulate=True doubles every gradient
    rt.backward(dlogits.cuda(), params, params16, grads, accumulate=True)
    assert rel_err(grads.cpu(), 2 * gcpu) < 1e-5
    return logits, rt, errs


def test_micro_vs_emulating_oracle(eng, golden):
    logits, _, _ = run_case(eng, "micro", dict(W.MICRO), (1, 2))
    g = golden("micro_vit.npz")
    e = rel_err(logits, g["logits"])
    report(f"micro G4 logits vs fp32 reference golden: rel {e:.3e}")
    assert e < 3e-2


def test_tiny_vs_emulating_oracle(eng, golden):
    """BASELINE.json configs[0]: ViT3D tiny (64^3, p16, d192, L4, h3), batch 2."""
    logits, _, _ = run_case(eng, "tiny", dict(W.TINY), (3, 4))
    g = golden("tiny_vit.npz")
    e = rel_err(logits, g["logits"])
    report(f"tiny G4 logits vs fp32 reference golden: rel {e:.3e}")
    assert e < 3e-2


def test_inference_mode_matches_training_forward(eng):
    cfgdict = dict(W.MICRO)
    sd = W.make_tensors(W.vit_param_spec(**cfgdict), 1)
    cfg, off, num, arena = load_arena(eng, cfgdict, sd)
    params = arena.cuda()
    p16 = params.to(torch.bfloat16)
    rt = eng.VitRuntime(cfg)
    video = ref_cpu.fmri_to_video(W.make_volume((3, 32, 32, 32), 9).cuda())
    a = rt.forward(video, params, p16, training=True).clone()
    b = rt.forward(video, params, p16, training=False)
    assert torch.equal(a, b)           # same kernels, same order: bitwise identical
    c = rt.forward(video, params, p16, training=False)
    assert torch.equal(b, c)           # run-to-run deterministic


def test_base_config_properties(eng):
    """BASELINE.json configs[1] at full size (128^3, p16, d768, L12, h12, B=4): size-independent properties.
    (a) determinism, (b) batch independence: volume b's logits do not depend on its batch neighbours,
    (c) logits agree with the bf16-emulating oracle on one volume (CPU oracle takes a few seconds)."""
    cfgdict = dict(W.BASE)
    sd = W.make_tensors(W.vit_param_spec(**cfgdict), 5)
    cfg, off, num, arena = load_arena(eng, cfgdict, sd)
    params = arena.cuda()
    p16 = params.to(torch.bfloat16)
    rt = eng.VitRuntime(cfg)
    fmri = W.make_volume((4, 128, 128, 128), 6)
    video = ref_cpu.fmri_to_video(fmri.cuda())
    a = rt.forward(video, params, p16, training=False).clone()
    b = rt.forward(video, params, p16, training=False).clone()
    assert torch.equal(a, b)
    single = rt.forward(ref_cpu.fmri_to_video(fmri[2:3].cuda()), params, p16, training=False)
    assert torch.equal(single[0], a[2])
    with torch.no_grad():
        ref = ref_cpu.vit_forward(sd, ref_cpu.ViTCfg(**cfgdict), ref_cpu.fmri_to_video(fmri[:1]), emulate_bf16=True)
        ref32 = ref_cpu.vit_forward(sd, ref_cpu.ViTCfg(**cfgdict), ref_cpu.fmri_to_video(fmri[:1]))
    e, e32 = rel_err(a[:1], ref), rel_err(a[:1], ref32)
    report(f"base fwd logits vs emulating oracle: rel {e:.3e}; vs fp32 oracle (G4): rel {e32:.3e}")
    assert e <= REL
    assert e32 < 3e-2
