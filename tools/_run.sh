set -e
python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -1 gpurun_out/t.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/b.json 2>gpurun_out/b.err || { tail gpurun_out/b.err; exit 1; }
python -c "import json; j=json.load(open('gpurun_out/b.json')); print(j['value'], j['ms_per_step'], j['also']); r=j['roofline']; print(r['achieved'], r['achieved_single_stream'], r['by_kernel'])"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --same-device --no-extras 2>gpurun_out/dp.err | cut -c1-300
