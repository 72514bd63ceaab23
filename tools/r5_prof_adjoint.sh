#!/bin/bash
set -u
OUT=gpurun_out/r5adj
mkdir -p $OUT
export TMPDIR=/tmp
for cfg in pubmed-adjoint coauthor-adjoint; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$cfg -o p -- python bench.py --config $cfg > $OUT/$cfg.log 2>&1
  python tools/prof_summary.py "$(find $OUT/st_$cfg -name '*kernel_stats.csv' | head -1)" $OUT/${cfg}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --config $cfg" > /dev/null 2>> $OUT/$cfg.log
  python tools/trace_sequence.py "$(find $OUT/st_$cfg -name '*kernel_trace.csv' | head -1)" adaptive_control_kernel -1 20 > $OUT/${cfg}_trial_sequence.txt 2>&1
  find $OUT/st_$cfg -name '*kernel_trace.csv' -delete
done
head -16 $OUT/pubmed-adjoint_kernel_stats.csv | cut -c1-150
cat $OUT/pubmed-adjoint_trial_sequence.txt | cut -c1-120
