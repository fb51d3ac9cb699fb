# where does TRAINING_DROPOUT = 0.1 (the reference's default) cost 10 % of the train step?  kernel tables of both.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_drop
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for d in 0 0.1; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$d -- python3 $R/bench.py --steps 10 --warmup 3 --dropout $d --no-cpu-baseline --no-extras > $OUT/run_$d.log 2>&1
  T=$(find $OUT/stats_$d -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_timeline.py $T 18 --summary > $OUT/timeline_$d.txt
  rm -rf $OUT/stats_$d
done
head -32 $OUT/timeline_0.1.txt
