set -u
OUT=gpurun_out/r03m; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_golden_gpu.py tests/test_rmat_gpu.py -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
BENCH="python bench.py --graph arxiv --steps 20 --warmup 0 --no-cpu-baseline --no-roofline-probe --replays 1"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- $BENCH > "$OUT/stats.log" 2>&1
python tools/prof_summary.py "$(find $OUT/stats -name '*kernel_stats.csv' | head -1)" "$OUT/arxiv_kernel_stats.csv" "rocprofv3 --kernel-trace --stats -- $BENCH" > /dev/null 2>> "$OUT/stats.log"
find "$OUT" -name '*kernel_trace.csv' -delete
head -8 "$OUT/arxiv_kernel_stats.csv"
timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], [x['avg_us'] for x in r['secondary']])"
timeout 400 python bench.py --graph rmat --steps 8 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], [x['avg_us'] for x in r['secondary']])"
