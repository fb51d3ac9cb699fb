R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_env
rm -rf $OUT; mkdir -p $OUT
run() { echo "== $1"; env $2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/err.log | cut -c95-140; grep "host enqueue" $OUT/err.log | cut -c1-90; }
for i in 1 2; do
  run "default" "X=1"
  run "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=1"
  run "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=0"
  run "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=8"
  run "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=2"
done
echo done
