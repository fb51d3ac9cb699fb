#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/<tag>_pmc_hbm_traffic.csv (one row per
kernel instantiation) and profiles/<tag>_pmc_traffic.json (per nv_prof kind; read by bench.py for roofline.traffic).

    python tools/pmc_traffic_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <tag>

FETCH_SIZE / WRITE_SIZE are derived counters in kilobytes (x 1024 bytes here); MI355X_MICROARCH.md (HBM / rocprofv3 section)
prescribes doubling FETCH_SIZE on gfx950 for wide coalesced reads.  The calibration line printed at the end (adamw_kernel:
16 B/param read, 14 B/param written) checks both the unit and the correction on every run.

Kernel names are keyed by their full template instantiation: a leading `void ` and the trailing argument list are dropped,
`(anonymous namespace)::` is kept out of the key (round 2 split names at the FIRST parenthesis, which collapsed every kernel of
an anonymous namespace - gemm_pp_kernel<...>, gemm_pp_grouped_tn_kernel, the skinny kernels - into the rows "void " and "").
"""
import collections
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


from kernel_names import kernel_key, kind_of      # noqa: E402  (tools/kernel_names.py: demangles the __bf16 / _Float16 template names)


def load(path, counter):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = kernel_key(r["Kernel_Name"])
        d = per.setdefault(k, [0, 0.0])
        d[0] += 1
        d[1] += float(r["Counter_Value"])
    return per


def main():
    fetch_csv, write_csv, tag = sys.argv[1:4]
    fetch, write = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows = []
    for k, (n, v) in fetch.items():
        w = write.get(k, [n, 0.0])
        rows.append((k, n, 2.0 * v * 1024 / n / 1e6, w[1] * 1024 / max(w[0], 1) / 1e6))     # KB -> MB, fetch x2 (gfx950)
    rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
    out_csv = os.path.join(root, "profiles", f"{tag}_pmc_hbm_traffic.csv")
    with open(out_csv, "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras`\n")
        f.write("# fetch corrected x2 as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950; calibration: adamw_kernel = 16 B read + 14 B written per parameter (88.58 M)\n")
        f.write("kernel,nv_prof_kind,launches,fetch_MB_per_launch_corrected,write_MB_per_launch\n")
        for k, n, fm, wm in rows:
            kd = kind_of(k)
            f.write(f"\"{k}\",{'' if kd is None else kd},{n},{fm:.3f},{wm:.3f}\n")
    steps = next((n for k, n, _, _ in rows if k.startswith("patch_ln_fwd_kernel")), None)   # one patch gather per train step (AdamW is 1 or 13 launches by placement)
    by_kind = {}
    for k, n, fm, wm in rows:
        kd = kind_of(k)
        if kd is None:
            continue
        d = by_kind.setdefault(kd, dict(launches=0, fetch_MB=0.0, write_MB=0.0, kernels=[]))
        d["launches"] += n
        d["fetch_MB"] += fm * n
        d["write_MB"] += wm * n
        d["kernels"].append(k)
    for d in by_kind.values():
        d["fetch_MB_per_launch"] = round(d.pop("fetch_MB") / d["launches"], 3)
        d["write_MB_per_launch"] = round(d.pop("write_MB") / d["launches"], 3)
        d["launches_per_step"] = None if not steps else round(d["launches"] / steps, 2)
    gem = [d for kd, d in by_kind.items() if kd in (0, 1, 2, 10, 11, 12, 13, 14, 20, 21, 22)]
    n = sum(d["launches"] for d in gem)
    fm = sum(d["fetch_MB_per_launch"] * d["launches"] for d in gem) / max(n, 1)
    wm = sum(d["write_MB_per_launch"] * d["launches"] for d in gem) / max(n, 1)
    js = {"kernel": "bf16 MFMA GEMM family (nv_prof kinds 0-2, 10-14, 20-22): gemm_ws_kernel<...>, gemm_pp_kernel<...>, gemm_pp_grouped_tn_kernel / gemm_pp_grouped_tn_adamw_kernel, gemm_pq_kernel<...>",
          "steps_profiled": steps, "launches": n, "launches_per_step": None if not steps else round(n / steps, 2),
          "fetch_MB_per_launch": round(fm, 2), "write_MB_per_launch": round(wm, 2), "traffic_MB_per_launch": round(fm + wm, 2),
          "by_kind": {str(k): v for k, v in sorted(by_kind.items())},
          "source": f"profiles/{tag}_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE in separate passes of `bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                    "--no-extras`; fetch x2 gfx950 correction; counts fabric requests incl. Infinity-Cache hits)"}
    json.dump(js, open(os.path.join(root, "profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in js.items() if k != "by_kind"}))
    for k, nn, a, b in rows:
        if k.startswith("adamw"):
            print(f"calibration adamw_kernel: read {a:.1f} MB (expect {16 * 88.58:.1f}), written {b:.1f} MB (expect {14 * 88.58:.1f})")


if __name__ == "__main__":
    main()
