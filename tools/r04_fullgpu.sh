set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_full
rm -rf $OUT; mkdir -p $OUT
python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1 || { tail $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
python bench.py --steps 30 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cut -c1-330 $OUT/bench.json; grep "host enqueue" $OUT/bench.err
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['cpu_baseline']['value'], d['cpu_baseline']['all_cores'], d['roofline']['frac'], d.get('also'))"
echo done
