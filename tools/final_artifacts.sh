# Round artefacts on the GPU box: parity report, bench line (+cpu baseline, secondary lines), rocprofv3 kernel stats, PMC traffic.
# Usage (from the repo root, through gpurun): bash tools/final_artifacts.sh [skip-tests | secondary]   (secondary: only the secondary lines + kernel stats - a second call)
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final
if [ "$1" != "secondary" ]; then
rm -rf $OUT; mkdir -p $OUT
rm -f $R/gpurun_out/parity_report.txt
if [ "$1" != "skip-tests" ]; then
  python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 || { tail -30 $OUT/pytest_gpu.log; exit 1; }
  tail -1 $OUT/pytest_gpu.log
  cp $R/gpurun_out/parity_report.txt $OUT/parity_report.txt
fi
python bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
cut -c1-300 $OUT/bench.json
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1 || { tail $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
exit 0
fi
mkdir -p $OUT
# secondary lines: inference forwards (bf16 / fp8 / fp32), batch 20 / 64, the large preset (bf16 and fp8-forward train step), dropout, the reference's shipped config (90^3, patch 9, batch 128)
( python bench.py --forward-only --steps 100 --warmup 10; NEUROVIT_LN_FOLD=0 python bench.py --forward-only --steps 100 --warmup 10; python bench.py --forward-only --operands fp16 --steps 100 --warmup 10;
  python bench.py --operands fp16 --steps 30 --warmup 5 --no-cpu-baseline --no-extras; python bench.py --forward-only --precise --steps 20 --warmup 3;
  python bench.py --forward-only --batch 20 --steps 20 --warmup 3; python bench.py --forward-only --batch 20 --fp8 --steps 20 --warmup 3;
  python bench.py --forward-only --batch 20 --precise --steps 10 --warmup 3;
  python bench.py --forward-only --batch 64 --steps 10 --warmup 3; python bench.py --forward-only --batch 64 --fp8 --steps 10 --warmup 3;
  python bench.py --preset large --steps 8 --warmup 2 --no-cpu-baseline --no-extras; python bench.py --preset large --fp8 --steps 8 --warmup 2 --no-cpu-baseline --no-extras;
  python bench.py --preset large --forward-only --steps 8 --warmup 2;
  python bench.py --preset large --forward-only --fp8 --steps 8 --warmup 2; python bench.py --dropout 0.1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras;
  python bench.py --preset reference --batch 128 --steps 6 --warmup 2 --no-cpu-baseline --no-extras; python bench.py --preset reference --batch 128 --forward-only --steps 6 --warmup 2 --no-cpu-baseline ) > $OUT/bench_secondary.jsonl 2> $OUT/bench_secondary.err || true
python tools/neuro4d_train_bench.py 2> /dev/null | tail -1 > $OUT/neuro4d_train.log || true
cat $OUT/neuro4d_train.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/stats_run.log 2>&1
# (PMC passes: tools/r05_pmc.sh)
T=$(find $OUT/stats -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_timeline.py $T 18 --summary > $OUT/timeline_summary.txt || true
echo done
