set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_fuse6
rm -rf $OUT; mkdir -p $OUT
for i in 1 2; do
  echo "== unfused"
  NEUROVIT_FUSE_UPDATE=0 timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/u_$i.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['loss'], j['roofline']['frac'])"
  echo "== fused persistent 128"
  NEUROVIT_FUSE_UPDATE=1 timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/f_$i.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['loss'], j['roofline']['frac'])"
  for c in 16 32 64 128 0; do
    echo "== per-layer update launch, cap $c"
    NEUROVIT_FUSE_UPDATE=3 NEUROVIT_ADAMW_CAP=$c timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/c${c}_$i.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['loss'], j['roofline']['frac'])"
  done
done
echo done
