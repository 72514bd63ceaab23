#!/bin/bash
# round 6, GPU call 3: training with the adjoint off at the C3 shape (recorded solve + reverse sweep vs host loop); C5 with live PMC passes
OUT=gpurun_out/r6c3
mkdir -p $OUT
T0=$(date +%s)
timeout 600 python bench.py --train --no-adjoint --steps 10 --warmup 2 > $OUT/train_no_adjoint.json 2> $OUT/train_no_adjoint.err
echo "train rc $? seconds $(( $(date +%s) - T0 ))"; tail -3 $OUT/train_no_adjoint.err
python -c "
import json; d=json.loads(open('$OUT/train_no_adjoint.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','forward_ms','backward_ms','vjp_stage_ms','host_loop','speedup_vs_host_loop','parity_vs_host_loop','train_solve_path')}, d['roofline']['frac'])"
T0=$(date +%s)
timeout 1200 python bench.py --graph rmat --steps 4 --warmup 1 --no-hbm-probe --keep-pmc $OUT/rmat_pmc > $OUT/rmat.out 2> $OUT/rmat.err
echo "rmat rc $? seconds $(( $(date +%s) - T0 ))"; tail -3 $OUT/rmat.err
python -c "
import json; d=json.loads(open('$OUT/rmat.out').read().strip().splitlines()[-1])
print(json.dumps(d)[:3000])"
