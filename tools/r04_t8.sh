set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_t8
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_modules_gpu.py -x -q -k "native or fused_step or graph" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
P='import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["loss"], j["roofline"]["frac"])'
for i in 1 2 3; do
  for f in 0 3; do
    echo "== mode $f"
    NEUROVIT_FUSE_UPDATE=$f timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/m${f}_$i.err | python -c "$P"
  done
done
echo done
