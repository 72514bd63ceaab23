set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03c
timeout 600 python -m pytest tests/test_early_stop_gpu.py tests/test_solver_gpu.py -m gpu -x -q -p no:cacheprovider -k "relabel or early_stop" > gpurun_out/r03c/pytest_es.log 2>&1; echo "rc $?" >> gpurun_out/r03c/pytest_es.log; tail -15 gpurun_out/r03c/pytest_es.log
