set -u
OUT=gpurun_out/r03q; mkdir -p $OUT
for W in 2; do
MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=4 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 2967$W tools/_dbg_general.py 2>&1 | grep "^world\|Error" | head
done
timeout 900 python -m pytest tests/test_sharded_gpu.py -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "rc $?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log | cut -c1-300
