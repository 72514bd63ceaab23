#!/bin/bash
# round 6, GPU call 13: small-graph lines after caching the time grid per block.t (no device -> host read per solve)
OUT=gpurun_out/r6c13
mkdir -p $OUT
run() { timeout 300 python bench.py "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d.get('value'), d.get('unit'), d.get('ms_per_step'), d.get('ms_train_step'), d.get('ms_test_step'))"; }
echo C1 euler; run --graph cora --function laplacian --method euler --steps 4 --warmup 4 --no-live-pmc --no-hbm-probe --replays 21 --no-cpu-baseline --no-configs
echo C1 rk4; run --graph cora --function laplacian --steps 100 --warmup 10 --no-live-pmc --no-hbm-probe --no-cpu-baseline --no-configs
echo C2; run --graph cora --steps 100 --warmup 10 --no-live-pmc --no-hbm-probe --no-cpu-baseline --no-configs
echo C2 T=18; run --graph cora --steps 18 --warmup 4 --no-live-pmc --no-hbm-probe --no-cpu-baseline --no-configs --replays 21
echo cora epoch; run --config cora-epoch --steps 40 --warmup 5 --no-cpu-baseline
timeout 900 python -m pytest tests/test_solver_gpu.py tests/test_golden_gpu.py tests/test_early_stop_gpu.py -x -q -m gpu 2>&1 | tail -3
