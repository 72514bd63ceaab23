set -u
OUT=gpurun_out/r03d; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sharded_gpu.py tests/test_solver_gpu.py tests/test_kernels_gpu.py -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "rc $?" >> $OUT/pytest.log; tail -12 $OUT/pytest.log
for W in 2 4; do
  GNPDE_RANKS_SHARE_DEVICE=1 MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=4 timeout 300 python -m torch.distributed.run --nnodes=1 \
    --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((29700 + W)) bench.py --gpus $W --steps 10 --warmup 2 \
    > "$OUT/bench_${W}ranks_one_gpu.log" 2>&1
  echo "rc $?" >> "$OUT/bench_${W}ranks_one_gpu.log"
  tail -2 "$OUT/bench_${W}ranks_one_gpu.log" | cut -c1-3000
done
timeout 200 python bench.py --steps 100 --warmup 10 > $OUT/bench_arxiv.json 2> $OUT/bench_arxiv.err; echo "rc $?"; tail -3 $OUT/bench_arxiv.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03d/bench_arxiv.json').read().strip().split('\n')[-1])
r=d['roofline']; print(d['value'], d['ms_per_step'], d.get('parity_vs_oracle_one_eval')); print({k:r[k] for k in ('bound','achieved','peak','frac','gather_model_gbs','avg_launch_us','traffic_gbs')}); print(r['ceiling']); print(r['secondary'])
PY
BENCH="python bench.py --graph arxiv --steps 20 --warmup 0 --no-cpu-baseline --no-roofline-probe --replays 1"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- $BENCH > "$OUT/stats.log" 2>&1
python tools/prof_summary.py "$(find $OUT/stats -name '*kernel_stats.csv' | head -1)" "$OUT/arxiv_kernel_stats.csv" "rocprofv3 --kernel-trace --stats -- $BENCH" > /dev/null 2>> "$OUT/stats.log"
find "$OUT" -name '*kernel_trace.csv' -delete
head -9 "$OUT/arxiv_kernel_stats.csv"
