#!/bin/bash
# rocprofv3 PMC passes (counters only, no trace domains) over tools/attn_trace_bench.py: per-kernel SQ shares of the attention kernels.
# usage (through gpurun, from the repo root): bash tools/pmc_attn.sh <outdir under gpurun_out> ; prints the table
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
R=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --output-format csv -d $OUT/p$i -- python3 $R/tools/attn_trace_bench.py 0 4 513 12 4 4097 16 x x > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"].split("(")[0], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
print("kernel,grid,launches,valu_share,mfma_share,wait_any_over_wave_cycles,lds_share,lds_bank_conflict,insts_valu,insts_mfma,insts_lds,insts_salu")
for k, v in acc.items():
    L = max(n[k].values()); c = {x: v[x] / n[k][x] for x in v}
    gui = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if not gui: continue
    print(f"{k[0]},{k[1]},{L},{4 * c.get('SQ_ACTIVE_INST_VALU', 0) / (gui * 1024):.3f},{c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (gui * 1024):.3f},"
          f"{c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.3f},{c.get('SQ_LDS_IDX_ACTIVE', 0) / (gui * 256):.3f},{c.get('SQ_LDS_BANK_CONFLICT', 0):.0f},"
          f"{c.get('SQ_INSTS_VALU', 0):.0f},{c.get('SQ_INSTS_MFMA', 0):.0f},{c.get('SQ_INSTS_LDS', 0):.0f},{c.get('SQ_INSTS_SALU', 0):.0f}")
PY
