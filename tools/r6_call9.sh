#!/bin/bash
# round 6, GPU call 9: q and k as two tables where a key row is shorter than a cache line (A = 16): bitwise test, headline A/B with live counters
OUT=gpurun_out/r6c9
mkdir -p $OUT
timeout 900 python -m pytest tests/test_solver_gpu.py -x -q -m gpu -k "key_table" 2>&1 | tail -8 | tee $OUT/tests.txt
for rep in 1 2; do for k in 0 1; do
  GNPDE_TUNE=14=$k timeout 400 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-hbm-probe --no-live-pmc 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('key_table knob $k:', d['value'], d['ms_per_step'], [ (s['kernel'][:14], s['avg_us']) for s in d['roofline']['secondary']], d['roofline']['avg_launch_us'])" | tee -a $OUT/bench_knobs.txt
done; done
cd /tmp && export TMPDIR=/tmp
for k in 0 1; do
GNPDE_TUNE=14=$k timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/st_$k -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 0 --no-cpu-baseline --no-roofline-probe --replays 1 --no-configs > $GRAFT_REPO_ROOT/$OUT/prof_$k.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py "$(find $GRAFT_REPO_ROOT/$OUT/st_$k -name '*kernel_stats.csv' | head -1)" $GRAFT_REPO_ROOT/$OUT/stats_$k.csv "knob $k" > /dev/null 2>&1
head -8 $GRAFT_REPO_ROOT/$OUT/stats_$k.csv | cut -c1-120
rm -rf $GRAFT_REPO_ROOT/$OUT/st_$k
done
