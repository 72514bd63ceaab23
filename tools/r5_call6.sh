#!/bin/bash
set -u
OUT=gpurun_out/r5c6
mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2 3; do timeout 120 python -m pytest tests/test_adjoint_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3; done | tee $OUT/adjoint_tests.log
CNT="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES"
timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/sq_train -o p -- python bench.py --train --steps 4 --warmup 1 --replays 1 --no-cpu-baseline --no-live-pmc > $OUT/sq_train.log 2>&1
python tools/pmc_sq_summary.py $OUT/sq_train "# rocprofv3 --pmc $CNT --kernel-trace -- python bench.py --train --steps 4 --warmup 1 --replays 1 --no-cpu-baseline --no-live-pmc" > $OUT/pmc_sq_train.txt 2>&1
timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/sq_epoch -o p -- python bench.py --config cora-epoch --steps 4 --warmup 3 --no-cpu-baseline > $OUT/sq_epoch.log 2>&1
python tools/pmc_sq_summary.py $OUT/sq_epoch "# rocprofv3 --pmc $CNT --kernel-trace -- python bench.py --config cora-epoch --steps 4 --warmup 3 --no-cpu-baseline" > $OUT/pmc_sq_cora_epoch.txt 2>&1
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete
head -30 $OUT/pmc_sq_train.txt | cut -c1-230
