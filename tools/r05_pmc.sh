# Round-5 utilisation evidence on the GPU box.  Usage: bash tools/r05_pmc.sh <out-subdir>
#   sq / fetch / write : rocprofv3 --pmc passes of the batch-4 train step (the profiler serialises dispatches under --pmc: every kernel ALONE on the chip)
#   trace_*            : rocprofv3 --kernel-trace of the same step in four stream / update placements - per-kernel durations IN the step, which
#                        is where the stretch of the data-gradient GEMMs and of ln_bwd shows and what it can be attributed to
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--steps 3 --warmup 2 --no-cpu-baseline --no-extras"
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $R/bench.py $B > $OUT/sq.log 2>&1
echo "sq pass done"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/write.log 2>&1
echo "traffic passes done"
export NEUROVIT_AUX_STREAM=0
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_single_stream -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $OUT/trace_single.log 2>&1
unset NEUROVIT_AUX_STREAM
export NEUROVIT_FUSE_UPDATE=0
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_two_streams_update_at_end -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $OUT/trace_fuse0.log 2>&1
export NEUROVIT_FUSE_UPDATE=3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_as_timed -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $OUT/trace_fuse3.log 2>&1
unset NEUROVIT_FUSE_UPDATE
echo "traces done"
for d in trace_single_stream trace_two_streams_update_at_end trace_as_timed; do
  f=$(find $OUT/$d -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_timeline.py $f 4 --summary > $OUT/$d.summary.txt 2>&1 || true
  gzip -9 $f || true
done
f=$(find $OUT/trace_as_timed -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_as_timed.csv || true
find $OUT -name "*.csv*" | head -20; du -sh $OUT
