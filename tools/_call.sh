set -u
OUT=gpurun_out/r03c; mkdir -p $OUT
timeout 200 python bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/bench_arxiv.json 2> $OUT/bench_arxiv.err; echo "rc $?"; tail -3 $OUT/bench_arxiv.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03c/bench_arxiv.json').read().strip().split('\n')[-1])
r=d['roofline']; print(d['value'], d['ms_per_step']); print({k:r[k] for k in ('bound','achieved','peak','frac','avg_launch_us','traffic_gbs')}); print(r['ceiling']); print(r['secondary']); print(r.get('traffic_source'))
PY
timeout 500 python -m pytest tests/test_solver_gpu.py tests/test_golden_gpu.py tests/test_rewiring_gpu.py tests/test_rewire_prims_gpu.py tests/test_autograd_gpu.py -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "rc $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 400 python bench.py --graph rmat --steps 4 --warmup 1 > $OUT/bench_rmat.json 2> $OUT/bench_rmat.err; echo "rc $?"; tail -3 $OUT/bench_rmat.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03c/bench_rmat.json').read().strip().split('\n')[-1])
r=d['roofline']; print(d['value'], d['ms_per_step']); print({k:r[k] for k in ('bound','achieved','peak','frac','frac_of_hbm_peak','avg_launch_us','traffic_gbs')}); print(r['ceiling']); print(r['secondary'])
PY
