set -u
OUT=gpurun_out/r03p; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_golden_gpu.py tests/test_rmat_gpu.py tests/test_autograd_gpu.py -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
for T in "" "12=1"; do
GNPDE_TUNE=$T timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('tune [$T]', d['value'], d['ms_per_step'], [x['avg_us'] for x in r['secondary']], r['frac'], r['ceiling']['avg_launch_us'], r['ceiling_uniform_random_ids']['avg_launch_us'])"
done
GNPDE_TUNE="" timeout 400 python bench.py --graph rmat --steps 8 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], [x['avg_us'] for x in r['secondary']], r['frac'], r['frac_of_row_gather_ceiling'])"
