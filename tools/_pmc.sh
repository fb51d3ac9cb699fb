set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc3 -- python3 $R/tools/gemm_bench.py --iters 5 > $R/gpurun_out/pmc3/run.log 2>&1
tail -2 $R/gpurun_out/pmc3/run.log
