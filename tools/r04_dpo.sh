# per-bucket AdamW on the side stream (TrainStep(overlap_optimizer=True), what a data-parallel run could use) against the update behind the backward pass, python-driven path, 1 GPU
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_dpo
rm -rf $OUT; mkdir -p $OUT
P='import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["loss"], j["roofline"]["frac"])'
for i in 1 2 3; do
  echo "== native step (default)"
  timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/n_$i.err | python -c "$P"
  echo "== python-driven step, AdamW behind the backward pass"
  NEUROVIT_NATIVE_STEP=0 timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/g_$i.err | python -c "$P"
  for b in 7 13; do
    echo "== python-driven step, AdamW per bucket on the side stream, $b buckets"
    NEUROVIT_NATIVE_STEP=0 timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras --overlap-optimizer --buckets $b 2> $OUT/o${b}_$i.err | python -c "$P"
  done
done
echo done
