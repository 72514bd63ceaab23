#!/bin/bash
# round 6, GPU call 11: `python bench.py --gpus 8` without a launcher, eight ranks on the one device (functional), and --gpus 4 on the R-MAT shape at scale 0.25
OUT=gpurun_out/r6c11
mkdir -p $OUT
T0=$(date +%s)
GNPDE_RANKS_SHARE_DEVICE=1 timeout 900 python bench.py --gpus 8 --steps 4 --warmup 1 > $OUT/gpus8.out 2> $OUT/gpus8.err
echo "gpus 8 rc $? seconds $(( $(date +%s) - T0 ))"
tail -1 $OUT/gpus8.out | cut -c1-1800
tail -3 $OUT/gpus8.err
T0=$(date +%s)
GNPDE_RANKS_SHARE_DEVICE=1 timeout 900 python bench.py --gpus 4 --steps 2 --warmup 1 --graph rmat --scale 0.125 > $OUT/gpus4_rmat.out 2> $OUT/gpus4_rmat.err
echo "gpus 4 rmat rc $? seconds $(( $(date +%s) - T0 ))"
tail -1 $OUT/gpus4_rmat.out | cut -c1-1500
tail -3 $OUT/gpus4_rmat.err
timeout 600 python -m pytest tests/test_tape_gpu.py -q -m gpu -k "budget" 2>&1 | tail -3
