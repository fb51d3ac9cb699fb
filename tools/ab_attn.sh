# same-box A/B of two library builds on the attention kernels: GPU-side durations from rocprofv3 traces
# usage: bash tools/ab_attn.sh [B n heads]...   (tools/_ab/lib_base.so = the other build)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SHAPES="${@:-4 513 12 20 513 12 4 4097 16 16 1001 8}"
for tag in base new base new; do
  if [ $tag = base ]; then export NEUROVIT_HIP_LIB=$R/tools/_ab/lib_base.so; else unset NEUROVIT_HIP_LIB; fi
  rm -rf /tmp/tr_$tag
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -- python3 $R/tools/attn_trace_bench.py 0 $SHAPES > /dev/null 2>&1
  echo "== $tag"; python3 $R/tools/trace_durations.py $(find /tmp/tr_$tag -name "*kernel_trace.csv") attn
done
