set -u
OUT=gpurun_out/r03g; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "rc $?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log
timeout 400 python bench.py --graph rmat --steps 8 --warmup 1 > $OUT/bench_rmat.json 2> $OUT/bench_rmat.err; echo "rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03g/bench_rmat.json').read().strip().split('\n')[-1])
r=d['roofline']; print(d['value'], d['ms_per_step']); print({k:r[k] for k in ('bound','achieved','peak','frac','row_gather_gbs','frac_of_row_gather_ceiling','avg_launch_us')}); print([(x['avg_us'],x['gbs']) for x in r['secondary']])
PY
