set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_t7
rm -rf $OUT; mkdir -p $OUT
P='import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["loss"], j["roofline"]["frac"])'
for i in 1 2 3; do
  for t in 0 1; do
    echo "== mode 3, update last in the batch: $t"
    NEUROVIT_ADAM_LAST=$t timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/t${t}_$i.err | python -c "$P"
  done
done
echo done
