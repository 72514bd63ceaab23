#!/bin/bash
# round 6, GPU call 8: adaptive adjoint trial step with two tail launches (error partials; controller inside the finish kernel), two trial steps per graph
OUT=gpurun_out/r6c8
mkdir -p $OUT
timeout 900 python -m pytest tests/test_adjoint_native_gpu.py tests/test_adjoint_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee $OUT/tests.txt
for cfg in pubmed-adjoint coauthor-adjoint; do timeout 300 python bench.py --config $cfg 2>/dev/null | tail -1 > $OUT/$cfg.json; python - <<PY
import json
d = json.loads(open('$OUT/$cfg.json').read())
print('$cfg', {k: d.get(k) for k in ('value', 'unit', 'forward_ms', 'backward_ms', 'evals_forward', 'augmented_evals_backward', 'backward_speedup_vs_flat_host_loop', 'parity_vs_flat_host_loop')})
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/tr_pub -o p -- python $GRAFT_REPO_ROOT/bench.py --config pubmed-adjoint > $GRAFT_REPO_ROOT/$OUT/tr_pub.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_sequence.py "$(find $OUT/tr_pub -name '*kernel_trace.csv' | head -1)" adaptive_finish2 -1 20 > $OUT/pubmed_adjoint_trial_sequence.txt 2>&1
tail -12 $OUT/pubmed_adjoint_trial_sequence.txt
rm -rf $OUT/tr_pub
