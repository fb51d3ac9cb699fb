set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_large
rm -rf $OUT; mkdir -p $OUT
python -m pytest tests/test_modules_gpu.py -x -q -k "fp8" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
for i in 1 2; do
  echo "== large train step bf16"; python bench.py --preset large --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2> $OUT/bf$i.err | cut -c95-140,250-300 | tee -a $OUT/large.log
  echo "== large train step fp8 forward"; python bench.py --preset large --fp8 --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2> $OUT/f8$i.err | cut -c95-140,250-300 | tee -a $OUT/large.log
done
echo done
