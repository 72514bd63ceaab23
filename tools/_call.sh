set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03c
timeout 600 python -m pytest tests/test_sharded_gpu.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r03c/pytest_sharded.log 2>&1; echo "rc $?" >> gpurun_out/r03c/pytest_sharded.log; tail -5 gpurun_out/r03c/pytest_sharded.log
for W in 2 4; do
  GNPDE_RANKS_SHARE_DEVICE=1 MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=4 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((29700 + W)) bench.py --gpus $W --steps 10 --warmup 2 > "gpurun_out/r03c/bench_${W}ranks_one_gpu.log" 2>&1
  echo "rc $?" >> "gpurun_out/r03c/bench_${W}ranks_one_gpu.log"
  tail -2 "gpurun_out/r03c/bench_${W}ranks_one_gpu.log" | cut -c1-300
done
