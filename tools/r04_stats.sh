# kernel-trace + stats pass of the default bench command (the artefact the bench line's per-launch timings are checked against)
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_stats
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/stats_run.log 2>&1
T=$(find $OUT/stats -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_timeline.py $T 18 --summary > $OUT/timeline_summary.txt
head -3 $OUT/timeline_summary.txt
grep -c . $T
