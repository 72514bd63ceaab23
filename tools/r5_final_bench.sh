#!/bin/bash
# the driver's command + the T = 100 line at HEAD (profiles/r05_bench_default_steps20.json, r05_bench_steps100.json)
set -u
OUT=gpurun_out/$1
mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 --keep-pmc $OUT/live_pmc > $OUT/bench_default_steps20.json 2> $OUT/bench_default.err
echo "default bench rc $? seconds $(( $(date +%s) - T0 ))" | tee $OUT/bench_default.time
timeout 400 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs > $OUT/bench_steps100.json 2>> $OUT/bench_default.err
python - $OUT <<'PY'
import json, sys, os
out = sys.argv[1]
d = json.loads(open(os.path.join(out, 'bench_default_steps20.json')).read().strip().splitlines()[-1])
r = d['roofline']
print('C3', d['value'], 'frac', r['frac'], 'alg', r['frac_algorithmic'], 'hbm_probe', (r.get('hbm_bound_probe') or {}).get('frac'))
for k, v in (d.get('configs') or {}).items():
  if isinstance(v, dict):
    print(k, '|', v.get('value'), v.get('unit'), '| ms/step', v.get('ms_per_step'), '|', v.get('error') or v.get('skipped') or '', v.get('backward_speedup_vs_flat_host_loop') or '')
  else:
    print(k, v)
e = json.loads(open(os.path.join(out, 'bench_steps100.json')).read().strip().splitlines()[-1])
print('steps100', e['value'], e['config'].get('node_relabelling') and e['config']['node_relabelling'].get('order'))
PY
