set -u
OUT=gpurun_out/r03l; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_golden_gpu.py -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
for T in "" "8=2"; do
GNPDE_TUNE=$T timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('tune [$T]', d['value'], d['ms_per_step'], [x['avg_us'] for x in r['secondary']])"
done
