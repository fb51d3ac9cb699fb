set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_fuse8
rm -rf $OUT; mkdir -p $OUT
P='import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["loss"], j["roofline"]["frac"])'
for i in 1 2; do
  echo "== unfused"
  NEUROVIT_FUSE_UPDATE=0 timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/u_$i.err | python -c "$P"
  for c in 256 512 1024 2048 0; do
    echo "== per-layer update launch, cap $c"
    NEUROVIT_FUSE_UPDATE=3 NEUROVIT_ADAMW_CAP=$c timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/c${c}_$i.err | python -c "$P"
  done
done
echo done
