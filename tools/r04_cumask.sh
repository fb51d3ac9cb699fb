set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_cumask
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 120 python tools/cu_mask_probe.py census 240 2>&1 | tee $OUT/census240.log
timeout -k 10 120 python tools/cu_mask_probe.py census 224 2>&1 | tee $OUT/census224.log
timeout -k 10 300 python tools/cu_mask_probe.py step 240 4 40 2>&1 | grep -v Warning | tee $OUT/step240.log
timeout -k 10 300 python tools/cu_mask_probe.py step 224 6 40 2>&1 | grep -v Warning | tee $OUT/step224.log
echo done
