#!/bin/bash
# round 6, GPU call 7: the GAT function on the native VJP stage (adjoint solve + recorded sweep), reference fixtures
OUT=gpurun_out/r6c7
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_tape_gpu.py tests/test_adjoint_native_gpu.py tests/test_adjoint_gpu.py tests/test_autograd_gpu.py -q -m gpu 2>&1 | tail -40 | tee $OUT/tests.txt
