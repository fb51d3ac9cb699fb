#!/usr/bin/env python3
"""Per-kernel GPU durations from a rocprofv3 --kernel-trace CSV, grouped by (kernel, grid): count, mean of the last 3/4, min.
usage: trace_durations.py <kernel_trace.csv> [substring]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
groups = collections.OrderedDict()
for r in rows:
    if sub and sub not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"].split("(")[0][:60], r["Grid_Size_X"], r["Grid_Size_Y"])
    groups.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, gx, gy), d in groups.items():
    tail = d[len(d) // 4:]
    print(f"{name:60s} grid {gx:>7}x{gy:<5} n={len(d):4d}  mean {sum(tail) / len(tail):9.2f} us  min {min(d):9.2f} us")
