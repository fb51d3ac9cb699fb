set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_fuse4
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -x -q -k "adamw or grouped or native or fused_step" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
python tools/fuse_probe.py
for i in 1 2; do
  echo "== large, unfused"
  NEUROVIT_FUSE_UPDATE=0 timeout -k 10 300 python bench.py --preset large --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2> $OUT/lu_$i.err | cut -c95-140
  for w in 0 128; do
    echo "== large, wgs=$w"
    NEUROVIT_ADAMW_WGS=$w timeout -k 10 300 python bench.py --preset large --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2> $OUT/lw${w}_$i.err | cut -c95-140
  done
done
echo done
