#!/bin/bash
# round 5, GPU call 5: suite after the cosine clamp change; default line without configs (live PMC child in the solver's order);
# projection variants gnpde_tune(8, 3 | 4 | 5)
set -u
OUT=gpurun_out/r5c5
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $OUT/bench_main.json 2> $OUT/bench_main.err
for v in 3 4 5; do
  GNPDE_TUNE=8=$v timeout 200 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-live-pmc --no-hbm-probe > $OUT/bench_tune8_$v.json 2> $OUT/bench_tune8_$v.err
done
python - $OUT <<'PY'
import json, sys, os
for f in ['bench_main', 'bench_tune8_3', 'bench_tune8_4', 'bench_tune8_5']:
  try:
    d = json.loads(open(os.path.join(sys.argv[1], f + '.json')).read().strip().splitlines()[-1])
    r = d['roofline']
    sec = {s_['kernel'][:14]: (s_.get('avg_us'), s_.get('traffic'), s_.get('frac_traffic')) for s_ in r.get('secondary', []) if 'kernel' in s_}
    print(f, d['value'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'traffic', r.get('traffic'), 'agg us', r.get('avg_launch_us'), sec)
  except Exception as exc:
    print(f, 'unreadable', exc)
PY
