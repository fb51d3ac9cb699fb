# rocprofv3 kernel stats of inference forwards: bash tools/prof_forward.sh <out-subdir> [bench flags, e.g. --precise]
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; shift
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --forward-only --steps 10 --warmup 3 "$@" > $OUT/run.log 2>&1
tail -2 $OUT/run.log | cut -c1-300
python3 $R/tools/kstats.py $(find $OUT/stats -name "*kernel_stats.csv") 13 | head -25
