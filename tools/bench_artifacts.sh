# Bench line + rocprofv3 kernel stats + PMC traffic passes on the GPU box (no test suite).  Usage: bash tools/bench_artifacts.sh <out-subdir>
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
rm -rf $OUT; mkdir -p $OUT
python bench.py --steps 30 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cut -c1-600 $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/stats_run.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head; du -sh $OUT
echo done
