# Round 4 A/B 1: native one-call train step + merged resident attention backward, against the round-3 forms (same box, alternating).
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_ab1
rm -rf $OUT; mkdir -p $OUT
python -m pytest tests/test_temporal_gpu.py -x -q > $OUT/pytest_attn.log 2>&1 || { tail -30 $OUT/pytest_attn.log; exit 1; }
tail -2 $OUT/pytest_attn.log
python -m pytest tests/test_modules_gpu.py -x -q -k "native or train_step or fused_step or trainer or temporal or 4d or neuro4d" > $OUT/pytest_step.log 2>&1 || { tail -30 $OUT/pytest_step.log; exit 1; }
tail -2 $OUT/pytest_step.log
for i in 1 2; do
  echo "== old forms (python-driven step, two attention-backward launches)"
  NEUROVIT_NATIVE_STEP=0 NEUROVIT_ATTN_MODE=100 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/old$i.err | cut -c95-140; grep "host enqueue" $OUT/old$i.err
  echo "== native step only"
  NEUROVIT_ATTN_MODE=100 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/nat$i.err | cut -c95-140; grep "host enqueue" $OUT/nat$i.err
  echo "== native step + merged attention backward"
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/new$i.err | cut -c95-140; grep "host enqueue" $OUT/new$i.err
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-extras > $OUT/trace_run.log 2>&1
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_timeline.py $T 8 --summary > $OUT/timeline_summary.txt
head -30 $OUT/timeline_summary.txt
cp $T $OUT/kernel_trace.csv; rm -rf $OUT/trace
echo done
