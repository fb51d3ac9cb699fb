#!/bin/bash
# rocprofv3 PMC passes over tools/gemm_bench.py (one shape, one tile): usage tools/pmc_gemm.sh <tile> <only> <outdir>
# Counters in their own runs (no trace domains), as the GPU pool requires.
set -e
TILE=$1; ONLY=$2; OUT=$3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAVES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $R/$OUT/$tag -- python3 $R/tools/gemm_bench.py --tile $TILE --only "$ONLY" --iters 3 > $R/$OUT/$tag.log 2>&1 || true
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$R/$OUT/*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        for k, v in acc.items():
            if "gemm" in k:
                print(k, {c: f"{x:.4g}" for c, x in v.items()})
PY
