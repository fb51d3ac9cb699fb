set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_t3
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests/test_modules_gpu.py tests/test_engine_gpu.py tests/test_dp_gpu.py tests/test_cube_demo_gpu.py -q -m gpu > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
