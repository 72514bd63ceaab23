set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03c
timeout 600 python -m pytest tests/test_solver_gpu.py -m gpu -x -q -p no:cacheprovider -k "relabel" > gpurun_out/r03c/pytest_relabel.log 2>&1; echo "rc $?" >> gpurun_out/r03c/pytest_relabel.log; tail -5 gpurun_out/r03c/pytest_relabel.log
timeout 500 python tools/reorder_probe.py > gpurun_out/r03c/reorder_probe2.txt 2>&1; echo "rc $?" >> gpurun_out/r03c/reorder_probe2.txt
tail -9 gpurun_out/r03c/reorder_probe2.txt | cut -c1-330
