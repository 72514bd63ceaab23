#!/bin/bash
# round 5, GPU call 2: the driver's command (default bench line with the configs block), raw PMC kept; tape + GNN tests after the fold / encoder change
set -u
OUT=gpurun_out/r5c2
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_tape_gpu.py tests/test_golden_gpu.py tests/test_early_stop_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest_subset.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_subset.log
tail -5 $OUT/pytest_subset.log
T0=$(date +%s)
timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 --keep-pmc $OUT/pmc > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc $? seconds $(( $(date +%s) - T0 ))" | tee $OUT/bench_default.time
tail -5 $OUT/bench_default.err
python - $OUT/bench_default.json <<'PY'
import json, sys
try:
  d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  r = d['roofline']
  print('C3', d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'alg', r['frac_algorithmic'], 'traffic', r.get('traffic'), 'hbm_probe', (r.get('hbm_bound_probe') or {}).get('frac'))
  for s_ in r.get('secondary', []):
    print('  secondary', s_.get('kernel', '')[:40], s_.get('avg_us'), s_.get('bytes'), s_.get('traffic'), s_.get('frac_traffic'))
  for k, v in (d.get('configs') or {}).items():
    if isinstance(v, dict):
      rr = v.get('roofline') or {}
      par = {a: v[a] for a in v if a.startswith('parity')}
      print(k, '|', v.get('value'), v.get('unit'), '| ms/step', v.get('ms_per_step'), '| frac', rr.get('frac'), 'alg', rr.get('frac_algorithmic'), '|', json.dumps(par)[:300], '|', v.get('error') or v.get('skipped') or '', '| s', v.get('seconds'))
    else:
      print(k, v)
except Exception as exc:
  print('unreadable', exc)
PY
