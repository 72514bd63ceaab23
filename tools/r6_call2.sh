#!/bin/bash
# round 6, GPU call 2: recorded fixed-grid training (tape tests), adjoint tests after the r_scale / midpoint changes
OUT=gpurun_out/r6c2
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_tape_gpu.py tests/test_adjoint_native_gpu.py tests/test_adjoint_gpu.py -x -q -m gpu 2>&1 | tail -30 | tee $OUT/tests.txt
