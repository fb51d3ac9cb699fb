R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_wt
rm -rf $OUT; mkdir -p $OUT
for i in 1 2; do
  for wt in 0 1 3; do
    echo "== write-through bits $wt"; NEUROVIT_GEMM_WT=$wt python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/w$wt$i.err | cut -c95-140,460-480
  done
done
NEUROVIT_GEMM_WT=3 python -m pytest tests/test_kernels_gpu.py -q -k "gemm" > $OUT/k.log 2>&1; tail -5 $OUT/k.log
echo done
