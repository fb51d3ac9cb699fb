#!/usr/bin/env python3
"""Timeline of ONE train step from a rocprofv3 --kernel-trace CSV: every dispatch in start order with its stream / queue, duration,
the gap to the previous dispatch's end on the same queue and the idle time of the whole device in front of it.
usage: trace_timeline.py <kernel_trace.csv> [step-index-from-the-end = 2] [--summary]
A step starts at its patch_ln_fwd_kernel dispatch and ends in front of the next one."""
import csv
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import pretty_many      # noqa: E402

path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 2
summary = "--summary" in sys.argv
rows = list(csv.DictReader(open(path)))
for r, n in zip(rows, pretty_many([r["Kernel_Name"] for r in rows])):
    r["Kernel_Name"] = n
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "patch_ln_fwd_kernel" in r["Kernel_Name"]]      # first kernel of a (training or inference) forward
if len(starts) < back + 2:
    sys.exit(f"only {len(starts)} forward passes in the trace")
lo, hi = starts[-back - 1], starts[-back] - 1
step = rows[lo:hi + 1]
t0 = step[0]["s"]
qkey = "Queue_Id" if "Queue_Id" in step[0] else ("Stream_Id" if "Stream_Id" in step[0] else None)
last_end_q, dev_end = {}, rows[lo - 1]["e"] if lo > 0 else rows[lo]["s"]
busy, gaps, n = 0, 0, 0
per = {}
union = 0
for r in step:
    q = r.get(qkey, "0") if qkey else "0"
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:70]
    gap_q = (r["s"] - last_end_q[q]) / 1e3 if q in last_end_q else float("nan")
    idle = max(0.0, (r["s"] - dev_end) / 1e3)
    dur = (r["e"] - r["s"]) / 1e3
    if not summary:
        print(f"{(r['s'] - t0) / 1e3:9.1f} us  q{q:>3}  {dur:8.2f} us  gap(q) {gap_q:7.2f}  idle(dev) {idle:6.2f}  grid {r.get('Grid_Size_X', '?'):>7} wg {r.get('Workgroup_Size_X', '?'):>4}  {name}")
    union += max(0, r["e"] - max(r["s"], dev_end))
    d = per.setdefault(name, [0, 0.0, 0.0])
    d[0] += 1; d[1] += dur; d[2] += idle
    last_end_q[q] = r["e"]
    dev_end = max(dev_end, r["e"])
    gaps += idle; n += 1
span = (max(r["e"] for r in step) - (rows[lo - 1]["e"] if lo > 0 else step[0]["s"])) / 1e3
print(f"# step: {n} dispatches, span {span:.1f} us, device busy (union) {union / 1e3:.1f} us, device idle {gaps:.1f} us, sum of durations {sum(v[1] for v in per.values()):.1f} us")
for name, (c, d, i) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"# {d:9.1f} us  {c:4d} x {d / c:8.2f} us   idle before {i:7.1f} us   {name}")
