set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_fuse3
rm -rf $OUT; mkdir -p $OUT
for i in 1 2; do
  echo "== unfused"
  NEUROVIT_FUSE_UPDATE=0 timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/u_$i.err | cut -c95-140
  for w in 0 96 112 120 128 136 144; do
    echo "== wgs=$w"
    NEUROVIT_ADAMW_WGS=$w timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/w${w}_$i.err | cut -c95-140
  done
done
echo done
