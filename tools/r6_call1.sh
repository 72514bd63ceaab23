#!/bin/bash
# round 6, GPU call 1: the driver's command at HEAD (last line must be the compact headline), the self-launching --gpus 2 test, tape tests
OUT=gpurun_out/r6c1
mkdir -p $OUT
T0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.out 2> $OUT/bench_default.err
echo "bench rc $? seconds $(( $(date +%s) - T0 ))" | tee $OUT/bench_default.time
tail -c 8192 $OUT/bench_default.out | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('parsed last line:', len(json.dumps(d)), d['value'], d['roofline']['frac'], d['cpu_baseline'])"
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_tape_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee $OUT/tests.txt
