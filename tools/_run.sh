set -e
python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -1 gpurun_out/t.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
python - <<'PY'
import torch, time
x = torch.zeros(256*256, device="cuda")
for n in (1,):
    for _ in range(100): x.add_(1.0)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(2000): x.add_(1.0)
    torch.cuda.synchronize(); print("tiny elementwise back-to-back: %.2f us per launch" % ((time.perf_counter()-t)/2000*1e6))
PY
