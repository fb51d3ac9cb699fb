set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_fuse7
rm -rf $OUT; mkdir -p $OUT
P='import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["loss"], j["roofline"]["frac"])'
for b in 2 8 16 32; do
  for i in 1 2; do
    for f in 0 3; do
      echo "== base batch $b mode=$f"
      NEUROVIT_FUSE_UPDATE=$f timeout -k 10 300 python bench.py --batch $b --steps 30 --warmup 6 --no-cpu-baseline --no-extras 2> $OUT/b${b}_f${f}_$i.err | python -c "$P"
    done
  done
done
for i in 1 2; do
  for f in 0 3; do
    echo "== large mode=$f"
    NEUROVIT_FUSE_UPDATE=$f timeout -k 10 300 python bench.py --preset large --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2> $OUT/l_f${f}_$i.err | python -c "$P"
  done
done
echo done
