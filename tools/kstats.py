#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV: per-kernel calls, total / average time per step.
usage: tools/kstats.py <kernel_stats.csv> [steps]   (steps omitted or 0: counted from the patch-gather launches, one per forward)"""
import csv
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import pretty_many      # noqa: E402
rows = list(csv.DictReader(open(sys.argv[1])))
for r, n in zip(rows, pretty_many([r["Name"] for r in rows])):
    r["Name"] = n
steps = float(sys.argv[2]) if len(sys.argv) > 2 and float(sys.argv[2]) > 0 else float(sum(int(r["Calls"]) for r in rows if "patch_ln_fwd" in r["Name"]) or 1)      # 0 / omitted: one patch gather per step
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time per step: {tot / steps / 1e6:.3f} ms; launches per step: {sum(int(r['Calls']) for r in rows) / steps:.1f}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print(f"{float(r['TotalDurationNs']) / steps / 1e3:9.1f} us/step  {int(r['Calls']) / steps:6.1f} x {float(r['AverageNs']) / 1e3:8.2f} us  {r['Name'][:110]}")
