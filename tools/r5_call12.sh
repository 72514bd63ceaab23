#!/bin/bash
set -u
OUT=gpurun_out/r5c12
mkdir -p $OUT
timeout 400 python -m pytest tests/test_adjoint_native_gpu.py tests/test_adjoint_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/adj_tests.txt
bash tools/r5_prof_adjoint.sh > $OUT/prof.log 2>&1
tail -8 $OUT/prof.log | cut -c1-120
grep -h "adaptive_control\|adaptive_finish" gpurun_out/r5adj/*_kernel_stats.csv | cut -c1-160
