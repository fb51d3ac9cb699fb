set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_graph
rm -rf $OUT; mkdir -p $OUT
python -m pytest tests/test_modules_gpu.py -x -q -k "graph_replayed or native" > $OUT/m.log 2>&1 || { tail -40 $OUT/m.log; exit 1; }
tail -2 $OUT/m.log
for i in 1 2 3; do
  echo "== eager native step"; NEUROVIT_GRAPH_STEP=0 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/e$i.err | cut -c95-140; grep "host enqueue" $OUT/e$i.err
  echo "== graph-replayed step"; NEUROVIT_GRAPH_STEP=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2> $OUT/g$i.err | cut -c95-140; grep "host enqueue" $OUT/g$i.err
done
echo done
