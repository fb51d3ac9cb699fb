# Kernel durations of the train step with the auxiliary (weight-gradient) stream off: every kernel runs alone, so its duration is
# its own in-step cost (cold caches, real operands) rather than a share of two overlapping kernels.  Usage: bash tools/serial_stats.sh <out-subdir>
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
rm -rf $OUT; mkdir -p $OUT
NEUROVIT_AUX_STREAM=0 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras > $OUT/bench_serial.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cut -c1-300 $OUT/bench_serial.json
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras > $OUT/bench_two_streams.json 2>> $OUT/bench.err
cut -c1-300 $OUT/bench_two_streams.json
cd /tmp && export TMPDIR=/tmp
export NEUROVIT_AUX_STREAM=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/stats_run.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -2
echo done
