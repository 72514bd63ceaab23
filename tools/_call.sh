set -u
OUT=gpurun_out/r03f; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_early_stop_gpu.py tests/test_solver_gpu.py -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "rc $?" >> $OUT/pytest.log; tail -25 $OUT/pytest.log
