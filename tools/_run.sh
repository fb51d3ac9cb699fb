set -e
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -1 gpurun_out/t.log
for i in 1 2 3; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"; done
