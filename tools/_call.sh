set -u
export TMPDIR=/tmp
bash tools/final_evidence_r3.sh r03z 961ec88
OUT=gpurun_out/r03z
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=8 > $OUT/pytest.log 2>&1; echo "rc $?" >> $OUT/pytest.log; tail -14 $OUT/pytest.log
for W in 2 4; do
  GNPDE_RANKS_SHARE_DEVICE=1 MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=4 timeout 300 python -m torch.distributed.run --nnodes=1     --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((29700 + W)) bench.py --gpus $W --steps 10 --warmup 2     > "$OUT/bench_${W}ranks_one_gpu.log" 2>&1
  echo "rc $?" >> "$OUT/bench_${W}ranks_one_gpu.log"
  tail -2 "$OUT/bench_${W}ranks_one_gpu.log" | cut -c1-400
done
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
