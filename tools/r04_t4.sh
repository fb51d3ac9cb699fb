set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_t4
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -x -q -k "adamw or grouped or native or fused_step or graph or trainer" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
P='import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["loss"], j["roofline"]["frac"], j["roofline"].get("frac_update_unfused"), j["config"]["adamw"][:60])'
for i in 1 2 3; do
  for f in 0 1 3; do
    echo "== mode $f"
    NEUROVIT_FUSE_UPDATE=$f timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/m${f}_$i.err | python -c "$P"
  done
done
echo "== default"
timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/d.err | python -c "$P"
echo done
