set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03y; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=6 > $OUT/pytest.log 2>&1; echo "rc $?" >> $OUT/pytest.log; tail -12 $OUT/pytest.log
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"; cut -c1-200 $OUT/bench_default.json
timeout 400 python bench.py --graph rmat --steps 8 --warmup 1 > $OUT/bench_rmat.json 2> $OUT/bench_rmat.err; echo "bench rmat rc $?"; cut -c1-200 $OUT/bench_rmat.json
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for W in 2 4; do
  GNPDE_RANKS_SHARE_DEVICE=1 MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=4 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((29700 + W)) bench.py --gpus $W --steps 10 --warmup 2 > "$OUT/bench_${W}ranks_one_gpu.log" 2>&1
  echo "rc $?" >> "$OUT/bench_${W}ranks_one_gpu.log"
  tail -2 "$OUT/bench_${W}ranks_one_gpu.log" | cut -c1-200
done
