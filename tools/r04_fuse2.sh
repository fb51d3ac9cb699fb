# Round 4: persistent form of the weight-gradient GEMM with the AdamW epilogue - how many workgroups (CUs) should hold the HBM-bound epilogues?
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_fuse2
rm -rf $OUT; mkdir -p $OUT
NEUROVIT_ADAMW_WGS=108 timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -x -q -k "adamw_update_in_its_epilogue or inside_the_weight_gradient" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
for i in 1 2; do
  for w in 0 160 128 108 72; do
    echo "== wgs=$w"
    NEUROVIT_ADAMW_WGS=$w timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/w${w}_$i.err | cut -c95-140
  done
done
echo done
