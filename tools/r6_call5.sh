#!/bin/bash
# round 6, GPU call 5: native adjoint stage for cosine / pearson / raw alpha; Pubmed- / CoauthorCS-like adjoint fixtures; projection default
OUT=gpurun_out/r6c5
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_tape_gpu.py tests/test_adjoint_native_gpu.py tests/test_adjoint_gpu.py -q -m gpu 2>&1 | tail -40 | tee $OUT/tests.txt
timeout 300 python tools/linear_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/linear_ab.txt
