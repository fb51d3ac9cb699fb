set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc1 $R/gpurun_out/pmc2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/tools/attn_bench.py > $R/gpurun_out/pmc1/run.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/tools/attn_bench.py > $R/gpurun_out/pmc2/run.log 2>&1
tail -2 $R/gpurun_out/pmc2/run.log
