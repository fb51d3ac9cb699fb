# Round artefacts on the GPU box: parity report, bench line (+cpu baseline, secondary lines), rocprofv3 kernel stats, PMC traffic.
# Usage (from the repo root, through gpurun): bash tools/final_artifacts.sh r01 v3
set -e
TAG=$1; VER=$2
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
rm -f $R/gpurun_out/parity_report.txt
python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 || { tail -30 $OUT/pytest_gpu.log; exit 1; }
tail -1 $OUT/pytest_gpu.log
cp $R/gpurun_out/parity_report.txt $OUT/parity_report.txt
python bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json | cut -c1-400
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1 || { tail $OUT/smoke.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/stats_run.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1
echo done
