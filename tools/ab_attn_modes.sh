# GPU-side durations of the attention kernels under different nv_attn_set_mode values (same build): bash tools/ab_attn_modes.sh "0 5" [B n heads]...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
MODES="$1"; shift
SHAPES="${@:-4 513 12}"
for rep in 1 2; do for m in $MODES; do
  rm -rf /tmp/trm_$m
  rocprofv3 --kernel-trace --output-format csv -d /tmp/trm_$m -- python3 $R/tools/attn_trace_bench.py $m $SHAPES > /dev/null 2>&1
  echo "== mode $m"; python3 $R/tools/trace_durations.py $(find /tmp/trm_$m -name "*kernel_trace.csv") attn
done; done
