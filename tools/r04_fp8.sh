set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_fp8
rm -rf $OUT; mkdir -p $OUT
python -m pytest tests/test_engine_gpu.py -x -q -k "fp8" > $OUT/pytest_fp8.log 2>&1 || { tail -40 $OUT/pytest_fp8.log; exit 1; }
tail -3 $OUT/pytest_fp8.log
python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -x -q -k "fp8 or f8" > $OUT/pytest_fp8b.log 2>&1 || { tail -40 $OUT/pytest_fp8b.log; exit 1; }
tail -2 $OUT/pytest_fp8b.log
grep -i "fp8 train" $R/gpurun_out/parity_report.txt | tail -4
timeout -k 10 200 tools/mfma_shape_bench 2>&1 | tee $OUT/mfma_shape.log
timeout -k 10 400 python bench.py --steps 30 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cut -c1-330 $OUT/bench.json; grep "host enqueue\|cpu baseline" $OUT/bench.err
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['cpu_baseline']['value'], d['cpu_baseline']['all_cores'], d['roofline']['frac'])"
echo done
