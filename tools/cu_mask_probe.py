#!/usr/bin/env python3
"""CU-masked streams on MI355X (hipExtStreamCreateWithCUMask): (1) census - which CUs / XCDs do the first N and the last 256 - N mask
bits select; (2) experiment - the train step of ViT3D-base (batch 4) with the compute streams confined to N CUs and the AdamW update of
the gradient buckets that become final early running beside the backward pass on the other 256 - N CUs.
usage: cu_mask_probe.py census [N] | step [N] [early_buckets] [steps]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neurovit_amd._cabi import check, lib  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    """torch ExternalStream over a HIP stream restricted to the CUs whose mask bits are listed."""
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask failed: {rc}"
    return torch.cuda.ExternalStream(st.value)


def census(stream, blocks=1024, threads=512, lds=144 * 1024, hold_us=300):
    out = torch.zeros(blocks, 2, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    check(lib.nv_cu_census(out.data_ptr(), blocks, threads, lds, hold_us, stream.cuda_stream), "nv_cu_census")
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype("uint32")
    places = set()
    for hw, xcc in o:
        cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 0x7
        places.add((int(xcc & 0xF), int(se), int(sh), int(cu)))
    return places


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "census"
    n_main = int(sys.argv[2]) if len(sys.argv) > 2 else 240
    if mode == "census":
        full = census(torch.cuda.current_stream())
        a = census(masked_stream(range(n_main)))
        b = census(masked_stream(range(n_main, 256)))
        per_xcc = lambda s: [sum(1 for p in s if p[0] == x) for x in range(8)]
        print(f"unmasked stream: {len(full)} distinct (xcc, se, sh, cu) places, per XCD {per_xcc(full)}")
        print(f"mask bits [0, {n_main}): {len(a)} places, per XCD {per_xcc(a)}")
        print(f"mask bits [{n_main}, 256): {len(b)} places, per XCD {per_xcc(b)}; overlap with the first set: {len(a & b)}")
        return
    early = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    import bench
    bench.torch = torch
    from neurovit_amd import config as nvcfg
    from neurovit_amd.NeuroEncoder import NeuroEncoder
    from neurovit_amd.trainer import TrainStep
    size = nvcfg.preset("base")
    S = size["TRAINING_VIT_INPUT_SIZE"]
    config = dict(DEVICE="cuda:0", TRAINING_DIM=3, TRAINING_DROPOUT=0.0, GRADCAM_CUBE_SIZE=8, DATASET_NAME="adni", TRAINING_LEARNING_RATE=1e-4,
                  TRAINING_WEIGHT_DECAY=1e-2, **size)
    x, y = bench.make_batch(4, S, torch.device("cuda:0"), 42)

    def run(tag, overlap, masked, n_buckets=14):
        torch.manual_seed(42)
        model = NeuroEncoder(config)
        model.train()
        step = TrainStep(model, n_buckets=n_buckets, overlap_optimizer=overlap)
        vit = model.volume_encoder.vit3d
        import contextlib
        ctx = contextlib.nullcontext()
        if masked:
            ctx = torch.cuda.stream(masked_stream(range(n_main)))       # the step (native or staged) runs on a masked current stream
            vit._rt._aux["cuda:0"] = masked_stream(range(n_main))
            if overlap:
                step.sync._comm_stream = masked_stream(range(n_main, 256))
        if overlap:
            # only the first `early` buckets (head + last layers: final early in the backward pass) are updated on the side stream;
            # the others wait for the end of the backward pass and run on the main stream (whole chip)
            late, orig = [], step._bucket_update
            counter = {"i": 0}

            def upd(b, e):
                counter["i"] += 1
                if counter["i"] > early:
                    late.append((b, e))
                else:
                    orig(b, e)
            step.sync.after_bucket = upd
            rest = step.optimizer.step_rest

            def step_rest(grad_scale=1.0):
                for b, e in late:
                    orig(b, e)
                late.clear()
                counter["i"] = 0
                rest(grad_scale=grad_scale)
            step.optimizer.step_rest = step_rest
        with ctx:
            for _ in range(10):
                loss = step(x, y)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = step(x, y)
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        print(f"{tag:70s} {ms:7.3f} ms/step  {4 / ms * 1e3:8.1f} volumes/s  loss {float(loss):.5f}", flush=True)

    for _ in range(2):
        run("native step (reference)", False, False)
        run(f"staged backward + AdamW per bucket on a side stream, no masks, early {early}", True, False)
        run(f"the same, compute on {n_main} CUs, AdamW on {256 - n_main} CUs", True, True)
        run(f"native step, compute confined to {n_main} CUs (cost of the mask alone)", False, True)


if __name__ == "__main__":
    main()
