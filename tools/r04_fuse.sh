# Round 4: AdamW inside the weight-gradient GEMM epilogues - parity tests, then a same-box A/B of the train step.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_fuse
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -x -q -k "adamw or grouped" > $OUT/pytest_k.log 2>&1 || { tail -30 $OUT/pytest_k.log; exit 1; }
tail -2 $OUT/pytest_k.log
timeout -k 10 500 python -m pytest tests/test_modules_gpu.py -x -q -k "native or fused_step" > $OUT/pytest_m.log 2>&1 || { tail -30 $OUT/pytest_m.log; exit 1; }
tail -2 $OUT/pytest_m.log
python tools/fuse_probe.py
for i in 1 2 3; do
  for f in 0 1; do
    echo "== fuse_update=$f"
    NEUROVIT_FUSE_UPDATE=$f timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/f${f}_$i.err | cut -c95-140
  done
done
cd /tmp && export TMPDIR=/tmp
NEUROVIT_FUSE_UPDATE=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-extras > $OUT/trace_run.log 2>&1
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_timeline.py $T 8 --summary > $OUT/timeline_summary.txt
head -24 $OUT/timeline_summary.txt
rm -rf $OUT/trace
echo done
