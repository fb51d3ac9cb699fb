set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_t2
rm -rf $OUT; mkdir -p $OUT
python -m pytest tests/test_engine_gpu.py -x -q -k "fp8" > $OUT/e.log 2>&1 || { tail -40 $OUT/e.log; exit 1; }
tail -2 $OUT/e.log
python -m pytest tests/test_modules_gpu.py tests/test_kernels_gpu.py -x -q -k "fp8 or f8" > $OUT/m.log 2>&1 || { tail -40 $OUT/m.log; exit 1; }
tail -2 $OUT/m.log
grep -i "fp8 training forward with dropout" $R/gpurun_out/parity_report.txt | tail -1
echo done
