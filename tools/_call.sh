for W in 1 2; do
MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=4 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 2967$W tools/_dbg_general.py 2>&1 | grep "^world\|Error" | head
done
