#!/bin/bash
set -u
OUT=gpurun_out/r5c13
mkdir -p $OUT
timeout 400 python -m pytest tests/test_adjoint_native_gpu.py tests/test_adjoint_gpu.py tests/test_tape_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/adj_tests.txt
grep -q "passed" $OUT/adj_tests.txt && ! grep -q "failed" $OUT/adj_tests.txt || exit 1
bash tools/r5_prof_adjoint.sh > $OUT/prof.log 2>&1
bash tools/r5_final_bench.sh r5c13
