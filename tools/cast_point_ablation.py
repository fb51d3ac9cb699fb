#!/usr/bin/env python3
"""Which bf16 cast point costs how much of the logits error?  (VERDICT r1 "Next round" item 1.)

Runs the bf16-emulating oracle (oracle/ref_cpu.py, emulate_bf16=True: same cast points as the HIP path; the GPU parity
tests show the HIP logits sit at the emulation's error to three digits) on the CPU and switches the forward cast points
off one at a time (CAST_OFF), and on one at a time, measuring max|logits - fp32 logits| / max|fp32 logits| - the
north-star's "logits within 1e-3 of CPU reference" figure.  Test infrastructure: not imported by the product.

    python tools/cast_point_ablation.py [--configs micro tiny neuro32 base] [--seeds 3] > profiles/r02_cast_point_ablation.txt
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import weights as W  # noqa: E402
from oracle import ref_cpu  # noqa: E402

POINTS = ["xp", "w", "xn1", "qkv", "p", "ao", "xn2", "h"]
CONFIGS = {
    "micro": (W.MICRO, 2),
    "tiny": (W.TINY, 2),
    "neuro32": (dict(image_size=32, image_patch_size=8, frames=32, frame_patch_size=8, num_classes=2, dim=1024, depth=6, heads=8,
                     mlp_dim=2048, channels=1, dim_head=64, pool="cls"), 2),
    "base": (W.BASE, 1),
}


def logits_err(cfgdict, B, seed, off):
    cfg = ref_cpu.ViTCfg(**cfgdict)
    sd = W.make_tensors(W.vit_param_spec(**cfgdict), seed)
    x = W.make_volume((B, cfg.image_size, cfg.image_size, cfg.image_size), seed + 100)
    video = ref_cpu.fmri_to_video(x)
    with torch.no_grad():
        ref = ref_cpu.vit_forward(sd, cfg, video)
        ref_cpu.CAST_OFF = set(off)
        try:
            emu = ref_cpu.vit_forward(sd, cfg, video, emulate_bf16=True)
        finally:
            ref_cpu.CAST_OFF = set()
    return ((emu - ref).abs().max() / ref.abs().max()).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", nargs="+", default=["micro", "tiny", "neuro32", "base"])
    ap.add_argument("--seeds", type=int, default=3)
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    print("# logits error vs fp32 (max-norm relative), bf16-emulating oracle; mean over seeds [min .. max]")
    print("# 'all casts' = the HIP path's arithmetic; '-X' = cast point X kept in fp32; 'only X' = every other point in fp32")
    for name in a.configs:
        cfgdict, B = CONFIGS[name]
        seeds = range(1, a.seeds + 1) if name != "base" else range(1, min(a.seeds, 2) + 1)
        rows = [("all casts", [])] + [(f"-{p}", [p]) for p in POINTS] + [(f"only {p}", [q for q in POINTS if q != p]) for p in POINTS] + \
               [("-w -xn1 -xn2", ["w", "xn1", "xn2"]), ("-xn1 -xn2 -ao -h (activations fp32, weights bf16)", ["xp", "xn1", "xn2", "ao", "h", "qkv", "p"]),
                ("none (sanity: 0)", POINTS)]
        print(f"\n## {name}: {({k: v for k, v in cfgdict.items() if k in ('image_size', 'image_patch_size', 'dim', 'depth', 'heads', 'mlp_dim')})}, batch {B}")
        for label, off in rows:
            errs = [logits_err(cfgdict, B, s, off) for s in seeds]
            print(f"{label:55s} {np.mean(errs):.3e}  [{min(errs):.2e} .. {max(errs):.2e}]", flush=True)


if __name__ == "__main__":
    main()
