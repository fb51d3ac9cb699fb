set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_w32
rm -rf $OUT; mkdir -p $OUT
python -m pytest tests/test_kernels_gpu.py -x -q -k "32x32x16 or ping_pong_kernel_forced" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
for i in 1 2; do
  echo "== 16x16x32 (pp forced)"; python tools/gemm_bench.py --tile pp --only "NT" --iters 30 2>&1 | grep -v amdgpu.ids | tee -a $OUT/narrow.log
  echo "== 32x32x16 (pp forced)"; python tools/gemm_bench.py --tile pp --w32 --only "NT" --iters 30 2>&1 | grep -v amdgpu.ids | tee -a $OUT/wide.log
done
echo done
