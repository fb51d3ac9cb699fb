"""Whole-encoder parity on MI355X: native engine (C-ABI nv_vit_forward / nv_vit_backward) vs the oracle.

Per-kernel parity (tests/test_kernels_gpu.py) is gated at 1e-3 against the bf16-emulating oracle.  For the WHOLE
encoder that form of gate is not meaningful: bf16 rounding is discontinuous, so two correct implementations whose
fp32 accumulation orders differ by 1e-6 decorrelate at the bf16 quantisation-noise level after a few layers
(measured: HIP vs emulating oracle grows 6e-4 -> 2e-3 over 4 blocks while every kernel matches to <1e-3).  The
whole-model gates are therefore three-way, against the exact fp32 oracle (itself pinned to the reference goldens):

  G3a  err(HIP, fp32) <= 1.5 * err(emulating oracle, fp32) + 2e-4   per stage, logits and per gradient (relative L2):
       the HIP path is as close to the fp32 truth as an exact emulation of its own cast points is;
  G3b  err(HIP, emulating oracle) <= 5e-3 (stages, L2) / 1.5e-2 (gradients, L2): decorrelation bound;
  G4   logits vs the fp32 golden logits produced by the imported reference, max-norm relative, gated PER CASE at 1.5 x the
       value measured on MI355X (the path is run-to-run deterministic): micro 1.92e-3 -> 3.0e-3, tiny 1.49e-3 -> 2.4e-3,
       base (one volume) 1.70e-3 -> 2.6e-3.  The north-star "1e-3" is out of reach of bf16 MFMA operands:
       profiles/r02_cast_point_ablation.txt switches the forward cast points off one at a time in the emulating oracle -
       no single point dominates, and bf16 WEIGHTS ALONE (every activation kept fp32) already cost 1.0e-3 (micro, tiny),
       4.6e-3 (d1024 L6) and 5.9e-3 (base); the reference's own CPU bf16 autocast sits at 0.9-1.3e-2.
Measured errors are appended to gpurun_out/parity_report.txt when that directory exists.
"""
import os

import numpy as np
import pytest
import torch

import weights as W
from conftest import ROOT, rel_err, rel_l2
from oracle import ref_cpu, train_step

pytestmark = pytest.mark.gpu
RATIO, SLACK = 1.5, 2e-4   # G3a
REL = 5e-3                 # G3b forward stages / logits, relative L2
MAXREL = 5e-3              # base-size logits (two numbers) vs emulating oracle
GRAD_REL = 1.5e-2          # G3b parameter gradients, relative L2
FP8_GRAD_REL = 6e-2        # fp8 training forward: first-step gradient arena against the emulating oracle, relative L2 (e4m3 forward noise feeds every gradient)
FORM_TIGHT = 2e-5          # ... of the head and the last block's FeedForward / out-projection parameters alone (no re-rounding downstream): measured <= 4e-7 over all bf16 cases
FORM_REL = 5e-3            # gradient arena of the cls-rows form against the every-row form (see run_case)
OPERANDS = "bf16"          # 16-bit operand format of the runs below; tests/test_fp16_gpu.py re-runs the cases with "fp16" (and tighter G3b / G4 gates)
DT16 = {"bf16": torch.bfloat16, "fp16": torch.float16}
LOSS_SCALE = 1.0           # factor on the loss whose gradients are compared (divided out again): the fp16 runs use a power of two, as training does


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


def run_case(engine, tag, cfgdict, seeds, B=2, dropout=(0.0, 0.0, 0)):
    """Every stage, the logits and every gradient against both oracles with the last block computed on ALL rows (as the reference
    does), then once more in the product's default form - the last block's out-projection / FeedForward on the cls rows only
    (rows_form = 2; the form is an argument of each call) - which must reproduce the logits and every gradient."""
    out = _run_case_full_rows(engine, tag, cfgdict, seeds, B, dropout)
    logits, rt, gcpu, (cfg, params, params16, video, dlogits), (names, off, num, ref_grads, grads32) = out
    rt2 = engine.VitRuntime(cfg)
    rt2.operands = OPERANDS
    logits2 = rt2.forward(video, params, params16, training=True, dropout=dropout, rows_form=2)
    grads2 = torch.zeros_like(params)
    rt2.backward(dlogits, params, params16, grads2, accumulate=False)
    if LOSS_SCALE != 1.0:
        grads2 /= LOSS_SCALE
    assert rel_l2(logits2, logits) < 1e-5, (tag, "cls-rows form: logits")
    # The two forms are the same arithmetic up to the summation order of the last block's Linear layers on the cls rows (weight-streaming
    # against tiled kernels): fp32-rounding differences (~1e-6) in that block.  Whether they STAY there is luck: every layer of the backward pass
    # rounds the residual gradient to bf16, and one flipped rounding is a 4e-3 change of that element which the layers below inherit - the
    # gradient arenas of the two forms then differ by 1e-8 (no flip: most seeds) or by up to ~2e-3 (flips; base(B=1): 1.6e-3 after an
    # unrelated 1-ulp change of the GELU's reciprocal moved one bf16 value of u).  So the cls-rows form is held to the SAME gates against the
    # two oracles as the every-row form (every parameter, three-way), and to the every-row form itself at the size of those gates.
    g2 = grads2.cpu()
    fails = []
    for k, o, nn in zip(names, off, num):
        hip = g2[o:o + nn].reshape(ref_grads[k].shape)
        e_he, e_h32, e_e32 = rel_l2(hip, ref_grads[k]), rel_l2(hip, grads32[k]), rel_l2(ref_grads[k], grads32[k])
        if not (e_h32 <= RATIO * e_e32 + SLACK and e_he <= GRAD_REL):
            fails.append((k, e_he, e_h32, e_e32))
    assert not fails, (tag, "cls-rows form: gradients against the oracles", fails)
    # ... and TIGHTLY where no later bf16 re-rounding can have flipped: the head and the last block's own FeedForward / out-projection
    # parameters are formed from the B cls rows before the attention backward mixes rows, so the two forms differ there by the summation
    # order of one or two products (and the isolated bf16 flips of dU / h that follow from it) - a systematic 1e-3-level defect of the
    # cls-row kernels' backward shows up here, three orders above what is measured (values in the report; gate = FORM_TIGHT)
    last = f"transformer.layers.{cfg.depth - 1}."
    tight = [k for k in names if k.startswith("mlp_head.") or k.startswith(last + "1.net.") or k.startswith(last + "0.to_out.")]
    worst = ("", 0.0)
    for k, o, nn in zip(names, off, num):
        if k in tight:
            d = rel_l2(g2[o:o + nn], gcpu[o:o + nn])
            worst = max(worst, (k, d), key=lambda t: t[1])
    report(f"{tag} cls-rows form vs all rows, tensors ahead of any re-rounding ({len(tight)}): worst {worst[0]} {worst[1]:.2e}")
    assert worst[1] < FORM_TIGHT, (tag, "cls-rows form: head / last-block FeedForward / out-projection gradients", worst)
    e = rel_l2(g2, gcpu)
    report(f"{tag} cls-rows form of the last block vs all rows: logits {rel_l2(logits2, logits):.2e}, gradient arena {e:.2e} (every parameter inside the oracle gates)")
    assert e < FORM_REL, (tag, "cls-rows form: gradients", e)
    return logits, rt, None


def _run_case_full_rows(engine, tag, cfgdict, seeds, B, dropout):
    sd = W.make_tensors(W.vit_param_spec(**cfgdict), seeds[0])
    S = cfgdict["image_size"]
    fmri = W.make_volume((B, S, S, S), seeds[1])
    ocfg = ref_cpu.ViTCfg(**cfgdict)
    cfg, off, num, arena = load_arena(engine, cfgdict, sd)
    params = arena.cuda()
    params16 = params.to(DT16[OPERANDS])
    rt = engine.VitRuntime(cfg)
    rt.operands = OPERANDS
    video = ref_cpu.fmri_to_video(fmri.cuda())
    logits = rt.forward(video, params, params16, training=True, dropout=dropout, rows_form=1)

    # ---- oracles: bf16-emulating (same cast points) and exact fp32, both with autograd for the gradients
    def oracle(emulate):
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        taps = {}
        lg = ref_cpu.vit_forward(leaves, ocfg, ref_cpu.fmri_to_video(fmri), emulate_bf16=emulate, taps=taps, dropout=dropout)
        return leaves, taps, lg

    leaves, taps, ref_logits = oracle(True)
    leaves32, taps32, logits32 = oracle(False)
    n, d = ocfg.num_patches + 1, ocfg.dim
    fails = []

    def three_way(kind, name, hip, emu, f32, limit):
        e_he, e_h32, e_e32 = rel_l2(hip, emu), rel_l2(hip, f32), rel_l2(emu, f32)
        report(f"{tag} {kind} {name}: hip-emu {e_he:.3e}  hip-fp32 {e_h32:.3e}  emu-fp32 {e_e32:.3e}")
        if not (e_h32 <= RATIO * e_e32 + SLACK and e_he <= limit):
            fails.append((kind, name, e_he, e_h32, e_e32))

    three_way("fwd", "A5", rt.tap("x0", -1, (B, n, d), torch.float32), taps["A5"], taps32["A5"], REL)
    for i in range(ocfg.depth):
        three_way("fwd", f"block{i}", rt.tap("x2", i, (B, n, d), torch.float32), taps[f"block{i}"], taps32[f"block{i}"], REL)
    three_way("fwd", "logits", logits, ref_logits, logits32, 2 * REL)     # 2 x B numbers at the end of the chain: noisier than a stage tensor (base: 6.0e-3)

    labels = torch.from_numpy(np.random.RandomState(seeds[1] + 7).randint(0, ocfg.num_classes, size=B)).long()
    names = list(leaves.keys())
    LS = LOSS_SCALE
    ref_grads = dict(zip(names, (g / LS for g in torch.autograd.grad(train_step.cross_entropy(ref_logits, labels) * LS, [leaves[k] for k in names]))))
    grads32 = dict(zip(names, torch.autograd.grad(train_step.cross_entropy(logits32, labels), [leaves32[k] for k in names])))
    # dlogits from the oracle's logits so the backward comparison is not polluted by the forward difference
    ld = ref_logits.detach().clone().requires_grad_(True)
    (dlogits,) = torch.autograd.grad(train_step.cross_entropy(ld, labels) * LS, ld)
    grads = torch.zeros_like(params)
    rt.backward(dlogits.cuda(), params, params16, grads, accumulate=False)
    if LS != 1.0:
        grads /= LS
    gcpu = grads.cpu()
    for k, o, nn in zip(names, off, num):
        three_way("grad", k, gcpu[o:o + nn].reshape(ref_grads[k].shape), ref_grads[k], grads32[k], GRAD_REL)
    assert not fails, fails
    # accumulate=True doubles every gradient
    if LS != 1.0:
        grads *= LS
    rt.backward(dlogits.cuda(), params, params16, grads, accumulate=True)
    assert rel_err(grads.cpu() / LS, 2 * gcpu) < 1e-5
    return logits, rt, gcpu, (cfg, params, params16, video, dlogits.cuda()), (names, off, num, ref_grads, grads32)


G4_GATE = {"micro": 3.0e-3, "tiny": 2.4e-3}     # 1.5 x measured (1.92e-3, 1.49e-3): see the module docstring


def check_g4(tag, logits, golden_logits):
    e = rel_err(logits, golden_logits)
    report(f"{tag} G4 logits vs fp32 reference golden: rel {e:.3e}")
    assert e < G4_GATE[tag.split()[0]], (tag, e)


def test_micro_vs_emulating_oracle(eng, golden):
    cfgdict = dict(W.MICRO)
    logits, _, _ = run_case(eng, "micro", cfgdict, (1, 2))
    check_g4("micro", logits, golden("micro_vit.npz")["logits"])


def test_tiny_vs_emulating_oracle(eng, golden):
    """BASELINE.json configs[0]: ViT3D tiny (64^3, p16, d192, L4, h3), batch 2."""
    logits, _, _ = run_case(eng, "tiny", dict(W.TINY), (3, 4))
    check_g4("tiny", logits, golden("tiny_vit.npz")["logits"])


def test_micro_train_mode_dropout_same_masks(eng):
    """Train-mode dropout (vit_3d.py:21,23,39,45,100; config default 0.1): the oracle regenerates the product's
    counter-based masks bit for bit, so forward stages, logits and every gradient are gated exactly as without dropout."""
    run_case(eng, "micro+dropout", dict(W.MICRO), (1, 2), dropout=(0.1, 0.2, 123456789))


def test_pool_mean(eng):
    """pool='mean' (vit_3d.py:127: x.mean(dim=1) instead of the cls row): token-mean kernel in front of the head, and a
    head backward that gives every token row dx / n - with and without dropout (the last FF dropout mask then applies to
    every row of the incoming gradient)."""
    run_case(eng, "micro+mean", dict(W.MICRO, pool="mean"), (11, 12))
    run_case(eng, "micro+mean+dropout", dict(W.MICRO, pool="mean"), (11, 12), dropout=(0.1, 0.2, 987654321))
    run_case(eng, "p9+mean", dict(W.MICRO, pool="mean", image_size=27, image_patch_size=9, frames=27, frame_patch_size=9), (13, 14), B=3)


def test_other_head_dims(eng):
    """dim_head 32 and 128 through the whole encoder (forward stages, logits, every gradient, with dropout for one of them): the
    attention of these runs on attention_generic.hip, everything else on the same kernels as dim_head 64."""
    run_case(eng, "dh32", dict(W.MICRO, dim_head=32, heads=4), (21, 22))
    run_case(eng, "dh128+dropout", dict(W.MICRO, dim_head=128, heads=2), (23, 24), dropout=(0.1, 0.1, 5150))


def test_odd_patch_size_like_reference_default(eng):
    """The reference's default geometry is 90^3 / patch 9 (configs/config.yaml:39-40): patch_dim = 729 is not a multiple
    of 8 (scalar gather path, zero-padded GEMM operands, padded weight-gradient scratch) and n = N+1 is odd."""
    cfgdict = dict(W.MICRO, image_size=27, image_patch_size=9, frames=27, frame_patch_size=9)
    run_case(eng, "p9", cfgdict, (5, 6))


def test_inner_dim_differs_from_dim(eng):
    """heads * dim_head != dim, as in the reference's hard-coded encoder (NeuroEncoder.py:187-190: dim 1024, heads 8 ->
    inner 512): to_qkv is [3*inner, dim], to_out.0 is [dim, inner]."""
    run_case(eng, "inner<dim", dict(W.MICRO, dim=192, heads=1, mlp_dim=320), (7, 8))
    run_case(eng, "inner>dim", dict(W.MICRO, dim=64, heads=3, mlp_dim=192), (9, 10))


def test_reference_default_geometry(eng):
    """configs/config.yaml:39-40 + NeuroEncoder.py:187-190 at full width: 90^3 volume, patch 9 (P = 729, n = 1001),
    dim 1024, heads 8 (inner 512), mlp 2048; depth cut from 6 to 2 so the CPU oracles finish in seconds."""
    cfgdict = dict(image_size=90, image_patch_size=9, frames=90, frame_patch_size=9, num_classes=2, dim=1024,
                   depth=2, heads=8, mlp_dim=2048, channels=1, dim_head=64, pool="cls")
    run_case(eng, "ref-default(L2)", cfgdict, (11, 12), B=1)


def test_large_geometry_properties(eng):
    """BASELINE.json configs[4] geometry (128^3, patch 8 -> n = 4097 tokens, dim 1024, heads 16, mlp 4096), depth cut
    to 2: logits vs the bf16-emulating and fp32 oracles on one volume, determinism, and finite gradients of the right size
    (long-sequence attention: 65 key tiles, ragged last tile)."""
    cfgdict = dict(image_size=128, image_patch_size=8, frames=128, frame_patch_size=8, num_classes=2, dim=1024,
                   depth=2, heads=16, mlp_dim=4096, channels=1, dim_head=64, pool="cls")
    sd = W.make_tensors(W.vit_param_spec(**cfgdict), 13)
    cfg, off, num, arena = load_arena(eng, cfgdict, sd)
    params = arena.cuda()
    p16 = params.to(torch.bfloat16)
    rt = eng.VitRuntime(cfg)
    fmri = W.make_volume((2, 128, 128, 128), 14)
    video = ref_cpu.fmri_to_video(fmri.cuda())
    a = rt.forward(video, params, p16, training=True).clone()
    grads = torch.zeros_like(params)
    dlogits = torch.tensor([[1.0, -1.0], [0.5, 0.25]], device="cuda")
    rt.backward(dlogits, params, p16, grads, False)
    g1 = grads.clone()
    b = rt.forward(video, params, p16, training=True).clone()
    grads.zero_()
    rt.backward(dlogits, params, p16, grads, False)
    assert torch.equal(a, b) and torch.equal(g1, grads)          # run-to-run deterministic, forward and backward
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    with torch.no_grad():
        ocfg = ref_cpu.ViTCfg(**cfgdict)
        ref = ref_cpu.vit_forward(sd, ocfg, ref_cpu.fmri_to_video(fmri[:1]), emulate_bf16=True)
        ref32 = ref_cpu.vit_forward(sd, ocfg, ref_cpu.fmri_to_video(fmri[:1]))
    e, e32 = rel_err(a[:1], ref), rel_err(a[:1], ref32)
    report(f"large-geometry(L2) fwd logits vs emulating oracle: rel {e:.3e}; vs fp32 oracle: rel {e32:.3e}")
    assert e <= MAXREL and e32 < 2.6e-3      # 1.5 x the measured 1.68e-3


def test_large_full_depth_batch4_properties(eng):
    """BASELINE.json configs[4] at FULL size (128^3, patch 8 -> n = 4097, dim 1024, depth 24, heads 16, mlp 4096), batch 4,
    bf16 engine: run-to-run determinism of a training forward + backward, batch independence (volume 3 alone == volume 3
    in the batch, bit for bit), finite non-zero gradients, and the logits of one volume against the bf16-emulating oracle
    (three-way against fp32, as everywhere in this file)."""
    cfgdict = dict(image_size=128, image_patch_size=8, frames=128, frame_patch_size=8, num_classes=2, dim=1024,
                   depth=24, heads=16, mlp_dim=4096, channels=1, dim_head=64, pool="cls")
    sd = W.make_tensors(W.vit_param_spec(**cfgdict), 41)
    cfg, off, num, arena = load_arena(eng, cfgdict, sd)
    params = arena.cuda()
    p16 = params.bfloat16()
    rt = eng.VitRuntime(cfg)
    fmri = W.make_volume((4, 128, 128, 128), 42)
    video = ref_cpu.fmri_to_video(fmri.cuda())
    dlogits = torch.tensor([[1.0, -1.0], [0.5, 0.25], [-0.75, 0.5], [0.1, -0.2]], device="cuda")
    a = rt.forward(video, params, p16, training=True).clone()
    grads = torch.zeros_like(params)
    rt.backward(dlogits, params, p16, grads, False)
    g1 = grads.clone()
    b = rt.forward(video, params, p16, training=True).clone()
    grads.zero_()
    rt.backward(dlogits, params, p16, grads, False)
    assert torch.equal(a, b) and torch.equal(g1, grads)
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    nz = [float(g1[o:o + n].abs().max()) > 0 for o, n in zip(off, num)]
    assert all(nz), "a parameter tensor received no gradient"
    single = rt.forward(ref_cpu.fmri_to_video(fmri[3:4].cuda()), params, p16, training=False)
    assert torch.equal(single[0], a[3])
    with torch.no_grad():
        ocfg = ref_cpu.ViTCfg(**cfgdict)
        v0 = ref_cpu.fmri_to_video(fmri[:1])
        emu = ref_cpu.vit_forward(sd, ocfg, v0, emulate_bf16=True)
        ref32 = ref_cpu.vit_forward(sd, ocfg, v0)
    e, e32, ee = rel_err(a[:1], emu), rel_err(a[:1], ref32), rel_err(emu, ref32)
    report(f"large full depth (L24, B4) logits: vs emulating oracle {e:.3e}; vs fp32 oracle {e32:.3e}; emulation vs fp32 {ee:.3e}")
    # two numbers after 24 blocks: HIP, emulation and fp32 differ pairwise by decorrelated bf16 noise of the same size (measured
    # 2.7e-3 / 5.1e-3 / 2.5e-3), so the three-way ratio is a loose 3x here; the per-stage three-way gates run at depth 2 above
    assert e32 <= 3.0 * ee + 1e-3, (e32, ee)
    assert e <= 2e-2           # decorrelation bound after 24 blocks


def test_base_config_forward_and_gradients_one_volume(eng):
    """BASELINE.json configs[1] at full size (128^3, p16, d768, L12, h12), one volume: every forward stage, the logits and EVERY
    parameter gradient through the same three-way gates as the small configurations (the two CPU oracle passes take ~1 min)."""
    run_case(eng, "base(B=1)", dict(W.BASE), (31, 32), B=1)


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


def test_cls_rows_form_under_dropout_applies_the_masks_of_the_dense_tensors(eng):
    """Round 4: the weight-streaming kernels of the cls-rows form carry the nn.Dropout masks (the mask of the dense tensor, hashed at the element
    offset of the strided view), so the form is taken under dropout too - a train-mode forward that records no graph still runs with the block
    dropout on (the frozen encoder of the 4D model under Trainer.train: Trainer.py:59, config4D.yaml TRAINING_DROPOUT 0.2).  Logits of
    rows_form 1 (every row) and 2 (cls rows) agree to fp32 rounding WITH dropout, training workspace or not - one differing mask bit would show
    as a difference of the order of the logits - and differ from the dropout-free logits (the masks really are applied); so do the gradients."""
    cfgdict = dict(W.MICRO)
    sd = W.make_tensors(W.vit_param_spec(**cfgdict), 1)
    cfg, off, num, arena = load_arena(eng, cfgdict, sd)
    params = arena.cuda()
    p16 = params.bfloat16()
    rt = eng.VitRuntime(cfg)
    video = ref_cpu.fmri_to_video(W.make_volume((3, 32, 32, 32), 9).cuda())
    drop = (0.2, 0.2, 424242)
    for training in (False, True):
        a = rt.forward(video, params, p16, training=training, dropout=drop, rows_form=1).clone()
        b = rt.forward(video, params, p16, training=training, dropout=drop, rows_form=2).clone()
        assert rel_l2(b, a) < 1e-5, (training, rel_l2(b, a))
        assert torch.equal(b, rt.forward(video, params, p16, training=training, dropout=drop, rows_form=2))      # deterministic
    clean = rt.forward(video, params, p16, training=False, rows_form=2)
    assert rel_l2(clean, a) > 1e-3
    # gradients of both forms under dropout
    dlogits = torch.tensor([[0.3, -0.3], [-0.2, 0.2], [0.1, -0.1]], device="cuda")
    grads = []
    for form in (1, 2):
        rt.forward(video, params, p16, training=True, dropout=drop, rows_form=form)
        g = torch.zeros_like(params)
        rt.backward(dlogits, params, p16, g, accumulate=False)
        grads.append(g.clone())
    e = rel_l2(grads[1], grads[0])
    report(f"cls-rows form under dropout 0.2 vs every row: gradient arena {e:.2e}")
    assert e < 5e-3, e        # (a differing mask bit would be a difference of order 0.1-1; fp32-rounding differences can reach 1e-3 through bf16 flips, see run_case)
    # and without dropout the two forms agree to fp32 rounding, training or not
    for training in (False, True):
        a = rt.forward(video, params, p16, training=training, rows_form=1).clone()
        b = rt.forward(video, params, p16, training=training, rows_form=2)
        assert rel_l2(b, a) < 1e-5


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
    assert e <= MAXREL          # two numbers: L2 and max-norm coincide
    assert e32 < 2.6e-3         # G4 at base size: 1.5 x the measured 1.70e-3


def test_fp8_inference_forward_vs_fp8_emulating_oracle(eng):
    """BASELINE.json configs[4] asks for fp8 MFMA on the large model.  The fp8 inference path (qkv / out-projection / FC1 / FC2 on OCP e4m3
    operands, per-row weight scales, calibrated per-tensor activation scales) against an oracle with the same cast points
    (oracle/ref_cpu.py: fp8_scales=...), three-way against fp32.  Tolerance, stated: e4m3 carries 3 mantissa bits (2^-4 relative
    per element), so logits sit at a few 1e-2 of the fp32 result - ten times the bf16 path; the gate is that the HIP path is no
    further from fp32 than the emulation of its own arithmetic (x 1.5 + 5e-3), and within 6e-2 of fp32 outright."""
    cfgdict = dict(W.MICRO, depth=3)
    sd = W.make_tensors(W.vit_param_spec(**cfgdict), 51)
    cfg, off, num, arena = load_arena(eng, cfgdict, sd)
    params = arena.cuda()
    p16 = params.bfloat16()
    rt = eng.VitRuntime(cfg)
    fmri = W.make_volume((3, 32, 32, 32), 52)
    video = ref_cpu.fmri_to_video(fmri.cuda())
    scales = rt.calibrate_fp8(video, params, p16)
    f8 = rt.quantize_fp8(params, scales)
    a = rt.forward_fp8(video, params, p16, f8).clone()
    b = rt.forward_fp8(video, params, p16, f8)
    assert torch.equal(a, b)
    bf = rt.forward(video, params, p16, training=False)
    with torch.no_grad():
        ocfg = ref_cpu.ViTCfg(**cfgdict)
        v = ref_cpu.fmri_to_video(fmri)
        ref32 = ref_cpu.vit_forward(sd, ocfg, v)
        emu8 = ref_cpu.vit_forward(sd, ocfg, v, emulate_bf16=True, fp8_scales=scales)
    e_hip, e_emu, e_pair, e_bf = rel_err(a, ref32), rel_err(emu8, ref32), rel_err(a, emu8), rel_err(bf, ref32)
    report(f"fp8 forward (micro, depth 3): HIP vs fp32 {e_hip:.3e}; fp8 emulation vs fp32 {e_emu:.3e}; HIP vs emulation {e_pair:.3e}; bf16 path vs fp32 {e_bf:.3e}")
    assert e_hip <= RATIO * e_emu + 5e-3 and e_hip < 6e-2
    assert e_pair < 6e-2
    # out-projection kept on bf16 operands (scale <= 0 for that linear): the three-linear form of round 2, same gates
    scales3 = rt.calibrate_fp8(video, params, p16, out_proj=False)
    assert all(row[3] == 0.0 for row in scales3) and all(row[3] > 0 for row in scales)
    c = rt.forward_fp8(video, params, p16, rt.quantize_fp8(params, scales3))
    with torch.no_grad():
        emu3 = ref_cpu.vit_forward(sd, ocfg, v, emulate_bf16=True, fp8_scales=scales3)
    e3, e3_emu = rel_err(c, ref32), rel_err(emu3, ref32)
    report(f"fp8 forward (micro, depth 3), out-projection on bf16: HIP vs fp32 {e3:.3e}; emulation vs fp32 {e3_emu:.3e}")
    assert not torch.equal(a, c) and e3 <= RATIO * e3_emu + 5e-3 and e3 < 6e-2


LARGE = dict(image_size=128, image_patch_size=8, frames=128, frame_patch_size=8, num_classes=2, dim=1024,
             heads=16, mlp_dim=4096, channels=1, dim_head=64, pool="cls")


def test_fp8_forward_large_geometry_depth2_vs_fp8_emulating_oracle(eng):
    """BASELINE.json configs[4] ("ViT3D-large ... fp8 MFMA") at ITS geometry - 128^3, patch 8 -> n = 4097 tokens, dim 1024, 16 heads,
    mlp 4096 - depth cut to 2 so the CPU oracles finish: the fp8 inference forward of one volume three-way against the fp8-emulating
    oracle (same e4m3 cast points, per-row weight scales, calibrated activation scales) and the fp32 oracle.  Tolerance as in
    test_fp8_inference_forward_vs_fp8_emulating_oracle: no further from fp32 than the emulation of its own arithmetic x 1.5 + 5e-3."""
    cfgdict = dict(LARGE, depth=2)
    sd = W.make_tensors(W.vit_param_spec(**cfgdict), 13)
    cfg, off, num, arena = load_arena(eng, cfgdict, sd)
    params = arena.cuda()
    p16 = params.bfloat16()
    rt = eng.VitRuntime(cfg)
    fmri = W.make_volume((1, 128, 128, 128), 14)
    video = ref_cpu.fmri_to_video(fmri.cuda())
    scales = rt.calibrate_fp8(video, params, p16)
    f8 = rt.quantize_fp8(params, scales)
    a = rt.forward_fp8(video, params, p16, f8).clone()
    assert torch.equal(a, rt.forward_fp8(video, params, p16, f8))
    with torch.no_grad():
        ocfg = ref_cpu.ViTCfg(**cfgdict)
        v = ref_cpu.fmri_to_video(fmri)
        ref32 = ref_cpu.vit_forward(sd, ocfg, v)
        emu8 = ref_cpu.vit_forward(sd, ocfg, v, emulate_bf16=True, fp8_scales=scales)
    e_hip, e_emu, e_pair = rel_err(a, ref32), rel_err(emu8, ref32), rel_err(a, emu8)
    report(f"fp8 forward (large geometry, depth 2, n = 4097): HIP vs fp32 {e_hip:.3e}; fp8 emulation vs fp32 {e_emu:.3e}; HIP vs emulation {e_pair:.3e}")
    assert e_hip <= RATIO * e_emu + 5e-3 and e_hip < 6e-2
    assert e_pair < 6e-2


def test_fp8_forward_large_full_depth_batch4_properties(eng):
    """configs[4] at FULL size (depth 24, batch 4) in fp8: run-to-run determinism, batch independence bit for bit (volume 2 alone
    == volume 2 in the batch), finite logits, and agreement with the bf16 forward of the same weights at the e4m3 noise level."""
    cfgdict = dict(LARGE, depth=24)
    sd = W.make_tensors(W.vit_param_spec(**cfgdict), 41)
    cfg, off, num, arena = load_arena(eng, cfgdict, sd)
    params = arena.cuda()
    p16 = params.bfloat16()
    rt = eng.VitRuntime(cfg)
    fmri = W.make_volume((4, 128, 128, 128), 42)
    video = ref_cpu.fmri_to_video(fmri.cuda())
    scales = rt.calibrate_fp8(video[:2], params, p16)
    f8 = rt.quantize_fp8(params, scales)
    a = rt.forward_fp8(video, params, p16, f8).clone()
    b = rt.forward_fp8(video, params, p16, f8).clone()
    assert torch.equal(a, b) and torch.isfinite(a).all()
    single = rt.forward_fp8(ref_cpu.fmri_to_video(fmri[2:3].cuda()), params, p16, f8)
    assert torch.equal(single[0], a[2])
    bf = rt.forward(video, params, p16, training=False)
    e = rel_err(a, bf)
    report(f"fp8 forward (large, depth 24, batch 4) vs bf16 forward: rel {e:.3e}")
    assert e < 0.15           # 24 blocks of e4m3 operand noise on two small logits (depth-3 micro: 1.6e-2 vs fp32)


# ---------------------------------------------------------------------------------------------------------------------
# fp8 TRAINING forward (BASELINE.json configs[4] is quoted "fwd / fwd+bwd"): qkv / FC1 / FC2 of every block on
# e4m3 operands in the forward of the train step, bf16 backward over the bf16 activations the same forward kernels write.
def test_fp8_training_forward_equals_the_inference_forward_and_fills_the_workspace(eng):
    """nv_vit_forward_fp8_train against nv_vit_forward_fp8 with the out-projection on bf16 operands: the same arithmetic, hence the
    same logits bit for bit; the bf16 LayerNorm output / statistics it leaves for the backward pass are those of the bf16 forward
    (layer 0: same input), qkv is what the fp8 GEMM stored, and h = gelu(u) to one bf16 rounding."""
    cfgdict = dict(W.MICRO, depth=3)
    sd = W.make_tensors(W.vit_param_spec(**cfgdict), 5)
    cfg, off, num, arena = load_arena(eng, cfgdict, sd)
    params = arena.cuda()
    p16 = params.bfloat16()
    rt = eng.VitRuntime(cfg)
    fmri = W.make_volume((3, 32, 32, 32), 6)
    video = ref_cpu.fmri_to_video(fmri.cuda())
    scales = rt.calibrate_fp8(video, params, p16, out_proj=False)
    f8 = rt.quantize_fp8(params, scales)
    for form in (1, 2):                                             # every row / the last block on its cls rows
        rt.rows_form = form
        inf = rt.forward_fp8(video, params, p16, f8).clone()
        trn = rt.forward_fp8_train(video, params, p16, f8).clone()
        assert torch.equal(inf, trn), form
    rt.rows_form = 1
    trn = rt.forward_fp8_train(video, params, p16, f8).clone()
    M, d, m = 3 * 65, cfgdict["dim"], cfgdict["mlp_dim"]
    taps8 = {k: rt.tap(k, 0, (M, w), torch.bfloat16).clone() for k, w in (("xn1", d), ("xn2", d), ("u", m), ("h", m), ("qkv", 3 * d))}
    st8 = rt.tap("st1", 0, (2, M), torch.float32).clone()
    rt.forward(video, params, p16, training=True, rows_form=1)
    assert torch.equal(taps8["xn1"], rt.tap("xn1", 0, (M, d), torch.bfloat16))        # same x0, same LayerNorm arithmetic
    assert rel_err(st8, rt.tap("st1", 0, (2, M), torch.float32)) < 1e-6                 # mean / rstd: the same formulas in another kernel (fma contraction may differ)
    gelu = torch.nn.functional.gelu(taps8["u"].float())
    assert rel_err(taps8["h"].float(), gelu) < 2 ** -7                                  # h16 = bf16(gelu(u32)), u16 = bf16(u32)
    assert rel_l2(taps8["qkv"].float(), rt.tap("qkv", 0, (M, 3 * d), torch.bfloat16).float()) < 5e-2   # e4m3 operands against bf16 operands
    # the backward pass runs on what the fp8 forward left, deterministically, and agrees with the bf16 step's gradients at the e4m3 noise level
    dlog = torch.tensor([[0.3, -0.3], [-0.2, 0.2], [0.1, -0.1]], device="cuda")
    g_bf = torch.zeros_like(params)
    rt.backward(dlog, params, p16, g_bf, accumulate=False)
    rt.forward_fp8_train(video, params, p16, f8)
    g8 = torch.zeros_like(params)
    rt.backward(dlog, params, p16, g8, accumulate=False)
    rt.forward_fp8_train(video, params, p16, f8)
    g8b = torch.zeros_like(params)
    rt.backward(dlog, params, p16, g8b, accumulate=False)
    assert torch.equal(g8, g8b) and torch.isfinite(g8).all()
    cos = torch.nn.functional.cosine_similarity(g8.flatten(), g_bf.flatten(), dim=0).item()
    report(f"fp8 training forward (micro, depth 3): logits == fp8 inference forward (bitwise); gradient arena vs bf16 step: cosine {cos:.4f}, rel L2 {rel_l2(g8, g_bf):.3e}")
    assert cos > 0.98
    # train-mode dropout: the four block sites carry the bf16 path's masks (attention probabilities, out-projection, GELU output in BOTH
    # its e4m3 and its bf16 copy, FC2 output) - three-way on the logits against the oracle that restates masks and e4m3 cast points
    drop = (0.1, 0.1, 0x5eed)
    rt.rows_form = 1
    d8 = rt.forward_fp8_train(video, params, p16, f8, dropout=drop).clone()
    assert torch.equal(d8, rt.forward_fp8_train(video, params, p16, f8, dropout=drop)) and not torch.equal(d8, trn)
    with torch.no_grad():
        ocfg = ref_cpu.ViTCfg(**cfgdict)
        v = ref_cpu.fmri_to_video(fmri)
        ref32 = ref_cpu.vit_forward(sd, ocfg, v, dropout=drop)
        emu8 = ref_cpu.vit_forward(sd, ocfg, v, emulate_bf16=True, fp8_scales=scales, dropout=drop)
    e_hip, e_emu, e_pair = rel_err(d8, ref32), rel_err(emu8, ref32), rel_err(d8, emu8)
    report(f"fp8 training forward with dropout 0.1 (micro, depth 3): HIP vs fp32 {e_hip:.3e}; fp8 emulation vs fp32 {e_emu:.3e}; HIP vs emulation {e_pair:.3e}")
    assert e_hip <= RATIO * e_emu + 5e-3 and e_pair < 6e-2
    gd = torch.zeros_like(params)
    rt.backward(dlog, params, p16, gd, accumulate=False)          # the backward recomputes the same masks from the forward's arguments
    assert torch.isfinite(gd).all() and gd.abs().sum() > 0


def test_fp8_train_step_large_geometry_depth2_vs_fp8_emulating_oracle(eng):
    """Three train steps (forward on e4m3 operands, bf16 backward, fused AdamW, weights re-quantised in place after every step) at the
    geometry of BASELINE.json configs[4] - 128^3, patch 8, n = 4097, dim 1024, 16 heads, mlp 4096 - depth 2, one volume, against the
    oracle's train step with the same fp8 cast points (quantisation as a straight-through cast) and against the fp32 oracle.
    Stated tolerance: at every step the HIP loss is no further from the fp32 loss than 1.5 x the largest distance the emulation of its
    own arithmetic shows over the three steps + 5e-3, within 2e-2 of the emulation's loss, and the first step's gradient arena points
    the way the emulation's does (cosine > 0.95; measured 1.00).  Measured losses: HIP 0.2126 / 0.1651 / 0.1278, emulation 0.2189 /
    0.1747 / 0.1312, fp32 0.2258 / 0.1770 / 0.1385."""
    from neurovit_amd.NeuroEncoder import NeuroEncoder
    from neurovit_amd.trainer import TrainStep
    from oracle import train_step as ots
    cfgdict = dict(LARGE, depth=2)
    sd = W.make_tensors(W.vit_param_spec(**cfgdict), 13)
    size = dict(TRAINING_VIT_DIM=1024, TRAINING_VIT_DEPTH=2, TRAINING_VIT_HEADS=16, TRAINING_VIT_MLP_DIM=4096)
    LR = 2e-6       # small enough that three AdamW steps on ONE volume move the loss gently (at 1e-4 the first step already drives it to zero)
    config = W.neuro_config(128, 8, DEVICE="cuda", TRAINING_LEARNING_RATE=LR, TRAINING_WEIGHT_DECAY=1e-2, **size)
    model = NeuroEncoder(config)
    model.load_state_dict({"volume_encoder.vit3d." + k: v for k, v in sd.items()}, strict=True)
    model.train()
    vit = model.volume_encoder.vit3d
    fmri = W.make_volume((1, 128, 128, 128), 14)
    y = torch.tensor([1])
    scales = vit.enable_fp8(ref_cpu.fmri_to_video(fmri.cuda()), out_proj=False, training=True)
    p8_ptr = vit._fp8["params8"].data_ptr()
    step = TrainStep(model)
    losses = []
    for i in range(3):
        losses.append(float(step(fmri.cuda(), y.cuda())))
        if i == 0:
            g_hip = vit.flat_gradients().clone().cpu()
    assert not step._native_ok(fmri.cuda(), y.cuda())                                   # fp8 training forwards take the general path
    assert vit._fp8["params8"].data_ptr() == p8_ptr, "the per-step re-quantisation must reuse its buffers"
    ocfg = ref_cpu.ViTCfg(**cfgdict)
    v = ref_cpu.fmri_to_video(fmri)
    curves = {}
    for tag, kw in (("fp32", {}), ("fp8 emulation", dict(emulate_bf16=True, fp8_scales=scales))):
        osd = {k: t.clone() for k, t in sd.items()}
        opt = ots.AdamW(osd, lr=LR, weight_decay=1e-2)
        cur = []
        for i in range(3):
            loss, _, grads = ots.train_step(osd, ocfg, opt, v, y, **kw)
            cur.append(float(loss))
            if i == 0 and tag != "fp32":
                off, num, _ = eng.param_layout(eng.make_config(**cfgdict))
                g_emu = torch.zeros_like(g_hip)
                for (k, _), o, n in zip(sd.items(), off, num):
                    if k in grads:
                        g_emu[o:o + n] = grads[k].reshape(-1)
        curves[tag] = cur
    cos = torch.nn.functional.cosine_similarity(g_hip.flatten(), g_emu.flatten(), dim=0).item()
    g_l2 = rel_l2(g_hip, g_emu)
    report(f"fp8 train step (large geometry, depth 2, 3 steps): losses HIP {losses}, fp8 emulation {curves['fp8 emulation']}, fp32 {curves['fp32']}; "
           f"first-step gradient arena vs emulation: cosine {cos:.4f}, rel-L2 {g_l2:.3e}")
    # the emulation's own distance from fp32 over the three steps is the yardstick (a scalar loss read from two small logits of ONE
    # volume: step by step the emulation may happen to land closer to fp32 than the kernels do, so the spread is taken over the curve)
    spread = max(abs(e - r) for e, r in zip(curves["fp8 emulation"], curves["fp32"]))
    for i in range(3):
        l32, lemu = curves["fp32"][i], curves["fp8 emulation"][i]
        assert abs(losses[i] - l32) <= RATIO * spread + 5e-3 * max(1.0, abs(l32)), (i, losses[i], lemu, l32, spread)
        assert abs(losses[i] - lemu) <= 2e-2, (i, losses[i], lemu)          # measured 3.3e-3 ... 9.6e-3
    # round 5: the oracle's fp8 linears restate the HIP backward itself (bf16 operands the forward kernels wrote, no straight-through estimator, no
    # saturation mask: ref_cpu._LinearF8), so the gradient arena is gated in relative L2 like any other emulation - not only in direction
    assert cos > 0.99 and g_l2 < FP8_GRAD_REL, (cos, g_l2)
    assert all(np.isfinite(losses)) and losses[2] < losses[0]           # it trains


@pytest.mark.parametrize("tag,cfgname,seeds,B", [("tiny", "TINY", (3, 4), 8), ("base", "BASE", (5, 6), 1), ("base", "BASE", (1, 2), 2), ("micro", "MICRO", (1, 2), 2)])
def test_inference_forward_with_folded_layernorms(eng, tag, cfgname, seeds, B):
    """nv_vit_forward_lnfold: the blocks' LayerNorms folded into the GEMMs around them - 21 of ViT3D-base's 24 LayerNorm launches gone.
    Same logits as the plain inference forward up to 16-bit rounding points that moved (bf16(x) W_g instead of bf16(LN(x)) W), as close to the fp32 oracle as
    the plain forward is; the Grad-CAM hook tensor (the last block's attention-LayerNorm output, never folded) still exists; shapes outside the LDS-epilogue
    kernels (micro) fall back to the plain launches bit for bit."""
    from neurovit_amd._cabi import lib
    cfgdict = dict(getattr(W, cfgname))
    sd = W.make_tensors(W.vit_param_spec(**cfgdict), seeds[0])
    cfg, off, num, arena = load_arena(eng, cfgdict, sd)
    params = arena.cuda()
    p16 = params.to(DT16[OPERANDS])
    rt = eng.VitRuntime(cfg)
    rt.operands = OPERANDS
    S = cfgdict["image_size"]
    fmri = W.make_volume((B, S, S, S), seeds[1])
    video = ref_cpu.fmri_to_video(fmri.cuda())
    ocfg = ref_cpu.ViTCfg(**cfgdict)
    n, d, L = ocfg.num_patches + 1, ocfg.dim, ocfg.depth
    plain = rt.forward(video, params, p16, training=False).clone()
    xn_plain = rt.tap("xn1", L - 1, (B, n, d), DT16[OPERANDS]).float().clone()
    fold = rt.lnfold_prepare(params)
    folded = rt.forward_lnfold(video, params, p16, fold).clone()
    xn_fold = rt.tap("xn1", L - 1, (B, n, d), DT16[OPERANDS]).float().clone()
    M, inner, m = B * n, ocfg.heads * ocfg.dim_head, ocfg.mlp_dim
    supported = all(lib.nv_gemm_lnfold_supported(*s) for s in ((M, 3 * inner, d), (M, m, d), (M, d, inner), (M, d, m)))
    if not supported:
        assert torch.equal(folded, plain) and torch.equal(xn_fold, xn_plain)
        return
    with torch.no_grad():
        ref32 = ref_cpu.vit_forward(sd, ocfg, ref_cpu.fmri_to_video(fmri[:1]))
    e_p, e_f, e_pf = rel_err(plain[:1], ref32), rel_err(folded[:1], ref32), rel_err(folded, plain)
    report(f"{tag} B={B} seeds {seeds} ({OPERANDS}) folded-LayerNorm inference forward: logits vs fp32 oracle {e_f:.3e} (plain launches {e_p:.3e}); folded vs plain {e_pf:.3e}; "
           f"hook tensor rel-L2 {rel_l2(xn_fold, xn_plain):.2e}")
    assert not torch.equal(folded, plain)                                  # (it did take the other kernels)
    assert e_f <= 1.5 * e_p + (2e-3 if OPERANDS == "bf16" else 3e-4), (e_f, e_p)
    assert e_pf <= (1e-2 if OPERANDS == "bf16" else 1.5e-3), e_pf          # (base, seeds (1, 2): 6.0e-3 - the pair of logits that sits at 1e-2 against fp32 in either form)
    # two 16-bit roundings of nearly the same tensor sit ~2^-8 (bf16) / 2^-11 (fp16) apart in relative L2: measured 4.9e-3 (B = 2) and 5.0e-3 (B = 1) for bf16
    assert rel_l2(xn_fold, xn_plain) <= (8e-3 if OPERANDS == "bf16" else 1e-3)
    assert torch.equal(rt.forward_lnfold(video, params, p16, fold), folded)     # run-to-run deterministic
