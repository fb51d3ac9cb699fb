# rocprofv3 kernel stats of the ViT3D-large train step: bash tools/prof_large.sh <out-subdir>
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --preset large --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-200
python3 $R/tools/kstats.py $(find $OUT/stats -name "*kernel_stats.csv") 6 | head -24
