# Round 4, first diagnostic: baseline bench line of the unchanged tree on this box + a full kernel trace of a few steps.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_diag1
rm -rf $OUT; mkdir -p $OUT
python bench.py --steps 30 --warmup 10 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cut -c1-400 $OUT/bench.json; grep "host enqueue" $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-extras > $OUT/trace_run.log 2>&1
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_timeline.py $T 8 > $OUT/timeline_step.txt
tail -45 $OUT/timeline_step.txt
cp $T $OUT/kernel_trace.csv; rm -rf $OUT/trace
echo done
