#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/<tag>_pmc_hbm_traffic.csv and
profiles/<tag>_pmc_traffic.json (read by bench.py for roofline.traffic).

    python tools/pmc_traffic_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <tag>

FETCH_SIZE / WRITE_SIZE are derived counters in kilobytes (x 1024 bytes here); MI355X_MICROARCH.md (HBM / rocprofv3 section)
prescribes doubling FETCH_SIZE on gfx950 for wide coalesced reads.  The calibration line printed at the end (adamw_kernel:
16 B/param read, 14 B/param written) checks both the unit and the correction on every run.
"""
import collections
import csv
import json
import os
import sys


def load(path, counter):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0]
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
        f.write("kernel,launches,fetch_MB_per_launch_corrected,write_MB_per_launch\n")
        for k, n, fm, wm in rows:
            f.write(f"\"{k}\",{n},{fm:.3f},{wm:.3f}\n")
    gem = [r for r in rows if "gemm_" in r[0]]
    n = sum(r[1] for r in gem)
    fm = sum(r[2] * r[1] for r in gem) / n
    wm = sum(r[3] * r[1] for r in gem) / n
    js = {"kernel": "gemm_*_kernel (all instantiations)", "launches": n, "fetch_MB_per_launch": round(fm, 2), "write_MB_per_launch": round(wm, 2),
          "traffic_MB_per_launch": round(fm + wm, 2),
          "source": f"profiles/{tag}_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; fetch x2 gfx950 correction; counts fabric requests incl. Infinity-Cache hits)"}
    json.dump(js, open(os.path.join(root, "profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(js))
    for k, nn, a, b in rows:
        if k.startswith("adamw"):
            print(f"calibration adamw_kernel: read {a:.1f} MB (expect {16 * 88.58:.1f}), written {b:.1f} MB (expect {14 * 88.58:.1f})")


if __name__ == "__main__":
    main()
