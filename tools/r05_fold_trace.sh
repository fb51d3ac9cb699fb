# kernel-trace of the inference forward with / without the LayerNorm fold.  Usage: bash tools/r05_fold_trace.sh <out-subdir> [batch]
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
BATCH=${2:-4}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  export NEUROVIT_LN_FOLD=$m
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fold$m -- python3 $R/bench.py --forward-only --batch $BATCH --steps 50 --warmup 10 --no-cpu-baseline > $OUT/fold$m.log 2>&1
  f=$(find $OUT/fold$m -name "*kernel_stats.csv" | head -1)
  python3 $R/tools/kstats.py $f 62 > $OUT/fold$m.kstats.txt
  rm -f $(find $OUT/fold$m -name "*kernel_trace.csv")
done
