#!/usr/bin/env python3
"""Which gradients differ between the cls-rows form and the every-row form of the last block (ViT3D-base, one volume)?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import weights as W  # noqa: E402
from neurovit_amd import engine  # noqa: E402
from oracle import ref_cpu  # noqa: E402

cfgdict = dict(W.BASE)
sd = W.make_tensors(W.vit_param_spec(**cfgdict), 31)
cfg = engine.make_config(**{k: v for k, v in cfgdict.items() if k != "pool"})
off, num, total = engine.param_layout(cfg)
arena = torch.zeros(total)
for (k, v), o, n in zip(sd.items(), off, num):
    arena[o:o + n] = v.reshape(-1)
params = arena.cuda(); p16 = params.bfloat16()
S = cfgdict["image_size"]
video = ref_cpu.fmri_to_video(W.make_volume((1, S, S, S), 32).cuda())
dlogits = torch.tensor([[0.3, -0.3]], device="cuda")
gs, taps = [], []
for form in (1, 2):
    rt = engine.VitRuntime(cfg)
    rt.forward(video, params, p16, training=True, rows_form=form)
    g = torch.zeros_like(params)
    rt.backward(dlogits, params, p16, g, accumulate=False)
    gs.append(g.clone())
    L = cfgdict["depth"] - 1
    n, d, m = 513, cfgdict["dim"], cfgdict["mlp_dim"]
    taps.append({k: rt.tap(k, L, (1, n, w), t)[0, 0].float().clone() for k, w, t in (("u", m, torch.bfloat16), ("h", m, torch.bfloat16), ("x1", d, torch.float32), ("x2", d, torch.float32))})
for k in taps[0]:
    a, b = taps[0][k], taps[1][k]
    print(f"cls row of {k}: max |diff| {float((a - b).abs().max()):.3e}  elements that differ {int((a != b).sum())} of {a.numel()}")
names = list(sd.keys())
tot = float((gs[0] - gs[1]).norm() / gs[0].norm())
print(f"gradient arena rel L2 {tot:.3e}")
rows = []
for k, o, n in zip(names, off, num):
    a, b = gs[0][o:o + n], gs[1][o:o + n]
    rows.append((float((a - b).norm() / a.norm().clamp_min(1e-30)), k, float(a.norm())))
for e, k, nrm in sorted(rows, reverse=True)[:12]:
    print(f"  {k:55s} rel {e:.3e}   |g| {nrm:.3e}")
