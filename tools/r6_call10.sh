#!/bin/bash
# round 6, GPU call 10: exp-kernel scores on the native VJP stage
OUT=gpurun_out/r6c10
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_tape_gpu.py tests/test_adjoint_native_gpu.py tests/test_adjoint_gpu.py tests/test_autograd_gpu.py -q -m gpu 2>&1 | tail -30 | tee $OUT/tests.txt
