set -u
OUT=gpurun_out/r03e; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_golden_gpu.py tests/test_solver_gpu.py tests/test_rmat_gpu.py tests/test_autograd_gpu.py -x -q -p no:cacheprovider --durations=5 > $OUT/pytest.log 2>&1; echo "rc $?" >> $OUT/pytest.log; tail -14 $OUT/pytest.log
timeout 200 python bench.py --steps 100 --warmup 10 > $OUT/bench_arxiv.json 2> $OUT/bench_arxiv.err; echo "rc $?"; tail -3 $OUT/bench_arxiv.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03e/bench_arxiv.json').read().strip().split('\n')[-1])
r=d['roofline']; print(d['value'], d['ms_per_step'], d.get('parity_vs_oracle_one_eval')); print({k:r[k] for k in ('bound','achieved','peak','frac','gather_model_gbs','avg_launch_us')}); print([ (x['avg_us'],x['gbs']) for x in r['secondary']])
PY
