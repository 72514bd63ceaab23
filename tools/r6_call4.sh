#!/bin/bash
# round 6, GPU call 4: projection kernel A/B (stand-alone and inside the headline solve)
OUT=gpurun_out/r6c4
mkdir -p $OUT
timeout 300 python tools/linear_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/linear_ab.txt
for k in 0 6 8 9; do
  GNPDE_TUNE=8=$k timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-live-pmc --no-hbm-probe 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('knob $k', d['value'], d['ms_per_step'], [ (s['kernel'][:20], s['avg_us']) for s in d['roofline']['secondary']])" | tee -a $OUT/bench_knobs.txt
done
