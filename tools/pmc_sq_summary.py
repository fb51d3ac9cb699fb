#!/usr/bin/env python3
"""Summarise one rocprofv3 --pmc pass of SQ counters over bench.py (tools/r05_pmc.sh) into profiles/<tag>_pmc_sq.csv (one row per kernel
instantiation) and profiles/<tag>_pmc_sq.json (per nv_prof kind: read by bench.py for roofline.by_kernel[*].mfma_busy).

    python tools/pmc_sq_summary.py <counter_collection.csv> <tag>

Definitions (MI355X_MICROARCH.md "rocprofv3 PMC slots"; profiles/r01_pmc_gemm_sq.txt used the same): GRBM_GUI_ACTIVE is summed over the 8 XCDs
(/ 8 = the kernel's cycles); SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD: mfma_busy = MFMA_BUSY / (GUI / 8 x 1024 SIMDs) = the share of
the chip's matrix-pipe time the launch used, ALONE on the chip (the profiler serialises dispatches under --pmc); SQ_WAVE_CYCLES, SQ_WAIT_*
and SQ_ACTIVE_INST_ANY count quad-cycles: wait_inst = WAIT_INST_ANY / WAVE_CYCLES (issue stalls: MFMA dependencies, pipe busy),
wait_any = WAIT_ANY / WAVE_CYCLES (parked at s_waitcnt / barriers), cu_busy = BUSY_CU_CYCLES / (GUI / 8 x 256 CUs x 4) where it is reported."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import kernel_key, kind_of      # noqa: E402


def main():
    path, tag = sys.argv[1:3]
    acc = collections.OrderedDict()
    launches = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        k = kernel_key(r["Kernel_Name"])
        d = acc.setdefault(k, collections.defaultdict(float))
        d[r["Counter_Name"]] += float(r["Counter_Value"])
        did = (k, r.get("Dispatch_Id"))
        if did not in seen:
            seen.add(did)
            launches[k] += 1
            d["grid"] = float(r.get("Grid_Size", 0) or 0)
    rows = []
    for k, d in acc.items():
        n = launches[k]
        gui = d["GRBM_GUI_ACTIVE"] / 8.0 / n
        if gui <= 0:
            continue
        wave = d["SQ_WAVE_CYCLES"] / n
        rows.append(dict(kernel=k, kind=kind_of(k), launches=n, cycles=gui, mfma_busy=d["SQ_VALU_MFMA_BUSY_CYCLES"] / n / (gui * 1024),
                         cu_busy=d["SQ_BUSY_CU_CYCLES"] / n / (gui * 1024), waves=d["SQ_WAVES"] / n,
                         wait_inst=(d["SQ_WAIT_INST_ANY"] / n / wave) if wave else 0.0, wait_any=(d["SQ_WAIT_ANY"] / n / wave) if wave else 0.0,
                         active=(d["SQ_ACTIVE_INST_ANY"] / n / wave) if wave else 0.0))
    rows.sort(key=lambda r: -r["cycles"] * r["launches"])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", f"{tag}_pmc_sq.csv"), "w") as f:
        f.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras\n")
        f.write("# per-launch averages, every kernel ALONE on the chip (dispatches are serialised under --pmc); definitions: tools/pmc_sq_summary.py\n")
        f.write("kernel,nv_prof_kind,launches,kernel_cycles,mfma_busy,cu_busy,waves,wait_inst_share,wait_any_share,active_inst_share\n")
        for r in rows:
            f.write(f"\"{r['kernel']}\",{'' if r['kind'] is None else r['kind']},{r['launches']},{r['cycles']:.0f},{r['mfma_busy']:.4f},{r['cu_busy']:.4f},{r['waves']:.0f},"
                    f"{r['wait_inst']:.3f},{r['wait_any']:.3f},{r['active']:.3f}\n")
    by_kind = {}
    for r in rows:
        if r["kind"] is None:
            continue
        d = by_kind.setdefault(r["kind"], dict(launches=0, cyc=0.0, busy=0.0, kernels=[]))
        d["launches"] += r["launches"]; d["cyc"] += r["cycles"] * r["launches"]; d["busy"] += r["mfma_busy"] * r["cycles"] * r["launches"]
        d["kernels"].append(r["kernel"])
    out = {str(k): dict(mfma_busy=round(v["busy"] / v["cyc"], 4), launches=v["launches"], kernels=v["kernels"]) for k, v in sorted(by_kind.items())}
    json.dump({"by_kind": out, "source": f"profiles/{tag}_pmc_sq.csv (rocprofv3 --pmc SQ pass of bench.py; kernels alone on the chip; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs))"},
              open(os.path.join(root, "profiles", f"{tag}_pmc_sq.json"), "w"), indent=1)
    for r in rows[:25]:
        print(f"{r['cycles'] * r['launches']:12.0f} cyc  {r['launches']:4d} x  mfma_busy {r['mfma_busy']:.3f}  wait_inst {r['wait_inst']:.2f} wait_any {r['wait_any']:.2f}  {r['kernel'][:90]}")


if __name__ == "__main__":
    main()
