"""Pin the CPU oracle against fixtures produced by the imported reference (G1, G2 of SURVEY §8d).

G1 bit-exact: patch indexing (A1), cls/pos scatter (A5 row layout).
G2 <=1e-5 rel fp32: every stage, grads, params after 1 and 3 AdamW steps, 4D path, Grad-CAM.
"""
import numpy as np
import pytest
import torch

import weights as W
from conftest import rel_err
from oracle import ref_cpu, train_step

TOL = 1e-5   # fp32 restatement vs imported reference (different op order in places)


# ------------------------------------------------------------------ G1: integer-exact indexing
@pytest.mark.parametrize("S,p", [(32, 8), (18, 9), (64, 16), (16, 8)])
def test_patch_index_bit_exact(golden, S, p):
    g = golden("patchify.npz")[f"tok_S{S}_p{p}"]
    vol = torch.arange(S ** 3, dtype=torch.float32).reshape(1, S, S, S)
    tok = ref_cpu.patchify(ref_cpu.fmri_to_video(vol), p, p, p)[0].numpy().astype(np.int32)
    assert np.array_equal(tok, g)
    assert np.array_equal(ref_cpu.patch_index_map(S, p), g.astype(np.int64))


def test_patch_index_hash_full_size(golden):
    g = int(golden("patchify.npz")["hash_S128_p16"][0])
    tok = torch.from_numpy(ref_cpu.patch_index_map(128, 16))
    pos = torch.arange(tok.numel(), dtype=torch.int64).reshape(tok.shape)
    assert int(((tok * 2654435761 + pos * 40503) % 2147483647).sum()) == g


# ------------------------------------------------------------------ G2: ViT-level stages / grads / steps
def _setup(tag):
    vcfg = dict(W.MICRO if tag == "micro" else W.TINY)
    seeds = (1, 2) if tag == "micro" else (3, 4)
    sd = W.make_tensors(W.vit_param_spec(**vcfg), seeds[0])
    S = vcfg["image_size"]
    fmri = W.make_volume((2, S, S, S), seeds[1])
    cfg = ref_cpu.ViTCfg(**vcfg)
    return cfg, sd, fmri


@pytest.mark.parametrize("tag", ["micro", "tiny"])
def test_weights_regenerate(golden, tag):
    g = golden(f"{tag}_vit.npz")
    _, sd, _ = _setup(tag)
    for k, v in W.checksums(sd).items():
        np.testing.assert_allclose(v, g["wsum." + k], rtol=1e-12)


@pytest.mark.parametrize("tag", ["micro", "tiny"])
def test_forward_stages(golden, tag):
    g = golden(f"{tag}_vit.npz")
    cfg, sd, fmri = _setup(tag)
    taps = {}
    with torch.no_grad():
        logits = ref_cpu.vit_forward(sd, cfg, ref_cpu.fmri_to_video(fmri), taps=taps)
    assert np.array_equal(taps["A1"][:, :4].numpy(), g["A1_head"])                  # bit exact gather
    assert rel_err(taps["A2"][:, :8], g["A2_head"]) < TOL
    for k in ("A3", "A4", "A5"):
        assert rel_err(taps[k], g[k]) < TOL, k
    # cls/pos scatter, bit exact: row 0 = cls+pos[0]; rows 1.. = A4 + pos[1..]
    a5 = taps["A5"]
    assert torch.equal(a5[:, 0], (sd["cls_token"] + sd["pos_embedding"][:, :1])[0].expand(2, -1))
    assert torch.equal(a5[:, 1:], taps["A4"] + sd["pos_embedding"][:, 1:])
    for i in range(cfg.depth):
        assert rel_err(taps[f"block{i}"], g[f"block{i}"]) < TOL, i
    pre = "transformer.layers.0.0."
    assert rel_err(taps[pre + "norm.out"], g["l0.norm_out"]) < TOL
    qkv = torch.cat([taps[pre + n].permute(0, 2, 1, 3).reshape(2, -1, cfg.inner) for n in "qkv"], dim=-1)
    assert rel_err(qkv, g["l0.qkv"]) < TOL
    assert rel_err(taps[pre + "attn.rowsum"], g["l0.attn_rowsum"]) < TOL
    assert rel_err(taps[pre + "attn.out"], g["l0.attn_out"]) < TOL
    assert rel_err(taps[f"transformer.layers.{cfg.depth - 1}.0.norm.out"], g["last.norm_out"]) < TOL
    assert rel_err(logits, g["logits"]) < TOL


@pytest.mark.parametrize("tag", ["micro", "tiny"])
def test_train_steps(golden, tag):
    g = golden(f"{tag}_vit.npz")
    cfg, sd, fmri = _setup(tag)
    lr, wd = g["hp"][0], g["hp"][1]
    video = ref_cpu.fmri_to_video(fmri)
    labels = torch.from_numpy(g["labels"]).long()
    params = {k: v.clone() for k, v in sd.items()}
    opt = train_step.AdamW(params, lr=lr, weight_decay=wd)
    loss, logits, grads = train_step.train_step(params, cfg, opt, video, labels)
    assert abs(loss.item() - g["loss"][0]) < 1e-6 * max(1.0, abs(g["loss"][0]))
    for key in g.files:
        if key.startswith("grad.") and "[" not in key:
            assert rel_err(grads[key[5:]], g[key]) < 2e-5, key
    assert rel_err(grads["to_patch_embedding.2.weight"][:, ::16], g["grad.to_patch_embedding.2.weight[:, ::16]"]) < 2e-5
    for k, v in W.checksums(grads).items():
        np.testing.assert_allclose(v[1], g["gradsum." + k][1], rtol=1e-4, err_msg=k)

    def check(step):
        # AdamW's m/(sqrt(v)+eps) is scale invariant, so last-bit differences in near-zero grads are
        # amplified to O(lr) parameter differences: 1e-5 after one step, 1e-4 after three.
        tol = TOL if step == 1 else 1e-4
        for key in g.files:
            if key.startswith(f"step{step}.") :
                assert rel_err(params[key[len(f"step{step}."):]], g[key]) < tol, key
        for k, v in W.checksums(params).items():
            np.testing.assert_allclose(v[1], g[f"step{step}sum." + k][1], rtol=1e-5, err_msg=k)

    check(1)
    losses = [loss.item()]
    for _ in range(2):
        l, _, _ = train_step.train_step(params, cfg, opt, video, labels)
        losses.append(l.item())
    check(3)
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ bf16-emulating mode sanity (G4 report)
def test_emulated_bf16_close_to_fp32():
    cfg, sd, fmri = _setup("micro")
    video = ref_cpu.fmri_to_video(fmri)
    with torch.no_grad():
        a = ref_cpu.vit_forward(sd, cfg, video)
        b = ref_cpu.vit_forward(sd, cfg, video, emulate_bf16=True)
    assert rel_err(b, a) < 3e-2          # bf16 operand rounding, same order as torch CPU bf16 autocast
    # emulated backward runs and is close to fp32 backward
    labels = torch.tensor([0, 1])
    g32 = train_step.train_step({k: v.clone() for k, v in sd.items()}, cfg,
                                train_step.AdamW({}, lr=0), video, labels)[2]
    g16 = train_step.train_step({k: v.clone() for k, v in sd.items()}, cfg,
                                train_step.AdamW({}, lr=0), video, labels, emulate_bf16=True)[2]
    for k in ("transformer.layers.0.0.to_qkv.weight", "to_patch_embedding.2.weight", "pos_embedding"):
        assert rel_err(g16[k], g32[k]) < 5e-2, k


# ------------------------------------------------------------------ NeuroEncoder level: 3D hooks, Grad-CAM, 4D
def _neuro_sd(S, p, seed):
    vc = dict(image_size=S, image_patch_size=p, frames=S, frame_patch_size=p, num_classes=2, dim=1024, depth=6,
              heads=8, mlp_dim=2048, channels=1, dim_head=64)
    return W.make_tensors(W.vit_param_spec(**vc), seed, prefix="volume_encoder.vit3d.")


def test_neuro3d_hooks_and_gradcam(golden):
    g = golden("neuro3d.npz")
    S, p = 32, 8
    config = W.neuro_config(S, p)
    sd = _neuro_sd(S, p, 11)
    assert list(sd.keys()) == list(g["keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["shapes"])
    x = W.make_volume((2, S, S, S), 12)
    leaves = {k: v.requires_grad_(True) for k, v in sd.items()}
    taps = {}
    logits = ref_cpu.neuro_forward(leaves, config, x, taps=taps)
    assert rel_err(logits, g["logits"]) < TOL
    act = taps["transformer.layers.5.0.norm.out"]
    assert rel_err(act, g["activations"]) < TOL
    loss = train_step.cross_entropy(logits, torch.from_numpy(g["labels"]).long())
    (grad,) = torch.autograd.grad(loss, act)
    assert rel_err(grad, g["gradients"]) < 5e-5
    # Grad-CAM tail from the reference's own hooked tensors
    cam = ref_cpu.grad_cam(torch.from_numpy(g["cam_activations"]), torch.from_numpy(g["cam_gradients"]), S, p,
                           config["GRADCAM_THRESHOLD"])
    assert rel_err(cam, g["cam"]) < TOL
    assert rel_err(cam[:, :, config["GRADCAM_SLICE_IDX"]], g["slice_attn"]) < TOL
    # and end to end from the input
    x1 = W.make_volume((1, S, S, S), 13)
    taps = {}
    out = ref_cpu.neuro_forward(leaves, config, x1, taps=taps)
    cls = out.argmax(dim=1)
    assert cls.item() == int(g["cam_class"][0])
    one_hot = torch.zeros_like(out)
    one_hot[0, cls] = 1
    a = taps["transformer.layers.5.0.norm.out"]
    (ga,) = torch.autograd.grad(out, a, grad_outputs=one_hot)
    cam2 = ref_cpu.grad_cam(a.detach(), ga, S, p, config["GRADCAM_THRESHOLD"])
    assert rel_err(cam2, g["cam"]) < 1e-3      # percentile threshold amplifies last-bit differences


def _rect_cfg():
    v = dict(W.RECT)
    (H, Wd), (p1, p2) = v.pop("image_size"), v.pop("image_patch_size")
    return ref_cpu.ViTCfg(image_size=H, image_patch_size=p1, image_width=Wd, patch_width=p2, **v)


def test_rect_vit(golden):
    """vit_3d.py:80-81: (height, width) pairs for the image and the patch, two channels - gather order bit exact, logits / loss /
    gradients of the restatement against the imported reference."""
    g = golden("rect_vit.npz")
    cfg = _rect_cfg()
    sd = W.make_tensors(W.vit_param_spec(**W.RECT), 61)
    H, Wd, p1, p2 = cfg.hw
    idx = torch.arange(cfg.channels * cfg.frames * H * Wd, dtype=torch.float32).reshape(1, cfg.channels, cfg.frames, H, Wd)
    assert np.array_equal(ref_cpu.patchify(idx, p1, p2, cfg.frame_patch_size)[0].numpy().astype(np.int32), g["tok"])
    video = torch.from_numpy(np.random.RandomState(62).standard_normal(size=(3, cfg.channels, cfg.frames, H, Wd)).astype(np.float32))
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    logits = ref_cpu.vit_forward(leaves, cfg, video)
    assert rel_err(logits, g["logits"]) < TOL
    loss = train_step.cross_entropy(logits, torch.from_numpy(g["labels"]).long())
    assert abs(loss.item() - g["loss"][0]) < 1e-6
    names = [k[5:] for k in g.files if k.startswith("grad.")]
    for k, v in zip(names, torch.autograd.grad(loss, [leaves[k] for k in names])):
        assert rel_err(v, g["grad." + k]) < 2e-5, k


def test_noproj_vit(golden):
    """heads == 1 with dim_head == dim: the reference drops to_out (vit_3d.py:32,43-46).  Restatement against the imported reference:
    logits, loss, every gradient."""
    g = golden("noproj_vit.npz")
    cfg = ref_cpu.ViTCfg(**W.NOPROJ)
    sd = W.make_tensors(W.vit_param_spec(**W.NOPROJ), 71)
    assert not any("to_out" in k for k in sd)
    S = W.NOPROJ["image_size"]
    video = ref_cpu.fmri_to_video(W.make_volume((3, S, S, S), 72))
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    logits = ref_cpu.vit_forward(leaves, cfg, video)
    assert rel_err(logits, g["logits"]) < TOL
    loss = train_step.cross_entropy(logits, torch.from_numpy(g["labels"]).long())
    assert abs(loss.item() - g["loss"][0]) < 1e-6
    names = list(sd)
    for k, v in zip(names, torch.autograd.grad(loss, [leaves[k] for k in names])):
        assert rel_err(v, g["grad." + k]) < 2e-5, k


def test_neuro4d(golden):
    g = golden("neuro4d.npz")
    S, p, T = 16, 8, 5
    sd = dict(_neuro_sd(S, p, 21))
    sd.update(W.make_tensors(W.temporal_param_spec(), 22))
    assert set(g["keys"]) == set(sd.keys())
    config = W.neuro_config(S, p, dim=4)
    x = W.make_volume((2, S, S, S, T), 23)
    trainable = set(g["trainable"])
    leaves = {k: (v.requires_grad_(True) if k in trainable else v) for k, v in sd.items()}
    logits = ref_cpu.neuro_forward(leaves, config, x)
    assert rel_err(logits, g["logits"]) < TOL
    loss = train_step.cross_entropy(logits, torch.from_numpy(g["labels"]).long())
    assert abs(loss.item() - g["loss"][0]) < 1e-6
    names = sorted(trainable)
    gr = torch.autograd.grad(loss, [leaves[k] for k in names])
    for k, v in zip(names, gr):
        # LayerNorm over d_model=2 features maps every row to (+-1, -+1): gradients that flow through
        # norm1/norm2 are zero in exact arithmetic and pure rounding noise (<=1e-6, differing at 1e-8) in fp32 -> abs floor.
        ref = torch.from_numpy(g["grad." + k])
        assert (v - ref).abs().max().item() <= 5e-5 * ref.abs().max().item() + 1e-7, k
    with torch.no_grad():
        vols = x.permute(0, 4, 1, 2, 3).reshape(2 * T, S, S, S)
        vl = ref_cpu.neuro_forward(sd, W.neuro_config(S, p, dim=3), vols)
    assert rel_err(vl, g["volume_logits"]) < TOL


def test_zscore_crop_restatement():
    """Row A0: per-sample z-score of the cropped volume: zero mean / unit population std per sample, 90^3 from the MNI grid,
    and the statistic of a 4D sample is taken over all of its timepoints (DatasetADNI_4D.py:87), not per timepoint."""
    g = np.random.default_rng(0)
    raw = g.normal(500.0, 100.0, size=(2, 91, 109, 91)).astype(np.float32)
    out = ref_cpu.zscore_crop(raw)
    assert out.shape == (2, 90, 90, 90) and out.dtype == np.float32
    assert np.allclose(out.reshape(2, -1).mean(1), 0, atol=1e-5) and np.allclose(out.reshape(2, -1).std(1), 1, atol=1e-5)
    c = raw[0, 1:, 10:-9, 1:]
    assert np.allclose(out[0], (c - c.mean()) / (c.std() + 1e-8), atol=1e-6)
    raw4 = g.normal(500.0, 100.0, size=(1, 91, 109, 91, 2)).astype(np.float32)
    raw4[..., 1] += 50.0
    out4 = ref_cpu.zscore_crop(raw4)
    assert out4.shape == (1, 90, 90, 90, 2) and abs(float(out4[..., 1].mean() - out4[..., 0].mean()) - 50.0 / float(raw4[0, 1:, 10:-9, 1:].std())) < 1e-3


def test_dropout_mask_generator_statistics():
    """The counter-based mask restated in oracle/ref_cpu.py::drop_mask (one 64-bit hash -> four 16-bit decisions): keep rate
    1 - p within 4 sigma, the four fields of a hash are uncorrelated, different seeds decorrelate, p = 0 / p = 1 are exact."""
    n = 1 << 20
    for p in (0.1, 0.2, 0.5):
        m = (ref_cpu.drop_mask(1234, p, (n,)) > 0).numpy()
        q = 1 - int(np.float32(p).astype(np.float64) * 65536) / 65536
        assert abs(m.mean() - q) < 4 * np.sqrt(q * (1 - q) / n), (p, m.mean())
        assert np.allclose(ref_cpu.drop_mask(1234, p, (n,)).unique().numpy(), [0, np.float32(1) / (np.float32(1) - np.float32(p))])
        lanes = m.reshape(-1, 4).astype(np.float64)
        c = np.corrcoef(lanes.T)
        assert np.abs(c - np.eye(4)).max() < 0.01
        m2 = (ref_cpu.drop_mask(1235, p, (n,)) > 0).numpy()
        assert abs(np.corrcoef(m, m2)[0, 1]) < 0.01
    assert torch.equal(ref_cpu.drop_mask(7, 0.0, (5, 8)), torch.ones(5, 8))
    assert torch.count_nonzero(ref_cpu.drop_mask(7, 1.0, (5, 8))) == 0
    a = ref_cpu.attn_drop_mask(9, 0.3, 1, 2, 5)               # n = 5 -> rows padded to 8 in the index space
    full = ref_cpu.drop_mask(9, 0.3, (1, 2, 5, 8))
    assert a.shape == (1, 2, 5, 5) and torch.equal(a, full[..., :5])
