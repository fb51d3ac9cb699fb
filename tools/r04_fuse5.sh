set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_fuse5
rm -rf $OUT; mkdir -p $OUT
for b in 2 8 16 32; do
  for i in 1 2; do
    for f in 0 1; do
      echo "== base batch $b fuse=$f"
      NEUROVIT_FUSE_UPDATE=$f timeout -k 10 300 python bench.py --batch $b --steps 30 --warmup 6 --no-cpu-baseline --no-extras 2> $OUT/b${b}_f${f}_$i.err | cut -c95-140
    done
  done
done
echo done
