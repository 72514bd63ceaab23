#!/bin/bash
# round 6, GPU call 6: tape records q||k and weights; permute kernel; training benches (adjoint off / on) + kernel stats
OUT=gpurun_out/r6c6
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_tape_gpu.py tests/test_adjoint_native_gpu.py tests/test_adjoint_gpu.py tests/test_solver_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee $OUT/tests.txt
timeout 600 python bench.py --train --no-adjoint --steps 10 --warmup 2 > $OUT/train_no_adjoint.json 2> $OUT/train_no_adjoint.err
python -c "
import json; d=json.loads(open('$OUT/train_no_adjoint.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','forward_ms','backward_ms','vjp_stage_ms','host_loop','speedup_vs_host_loop','parity_vs_host_loop','train_solve_path')}, d['roofline']['frac'])"
timeout 600 python bench.py --train --steps 10 --warmup 2 --no-live-pmc --no-cpu-baseline > $OUT/train_adjoint.json 2> $OUT/train_adjoint.err
python -c "
import json; d=json.loads(open('$OUT/train_adjoint.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value','forward_ms','backward_ms','f_plus_vjp_ms')})"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/st_train -o p -- python $GRAFT_REPO_ROOT/bench.py --train --no-adjoint --steps 10 --warmup 2 --replays 3 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py "$(find $OUT/st_train -name '*kernel_stats.csv' | head -1)" $OUT/train_no_adjoint_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --train --no-adjoint --steps 10 --warmup 2 --replays 3" 2>&1 | head -30
rm -rf $OUT/st_train
