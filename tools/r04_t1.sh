set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_t1
rm -rf $OUT; mkdir -p $OUT
python -m pytest tests/test_kernels_gpu.py -x -q -k "head or skinny or pool" > $OUT/k.log 2>&1 || { tail -40 $OUT/k.log; exit 1; }
tail -2 $OUT/k.log
python -m pytest tests/test_engine_gpu.py -x -q > $OUT/e.log 2>&1 || { tail -40 $OUT/e.log; exit 1; }
tail -2 $OUT/e.log
python -m pytest tests/test_modules_gpu.py -x -q -k "native or train_step or cls_rows or pool" > $OUT/m.log 2>&1 || { tail -40 $OUT/m.log; exit 1; }
tail -2 $OUT/m.log
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/b.err | cut -c95-140
echo done
