#!/bin/bash
set -u
OUT=gpurun_out/r5c9
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_adjoint_native_gpu.py tests/test_adjoint_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -25 | tee $OUT/adjoint_tests.log
timeout 200 python bench.py --config pubmed-adjoint > $OUT/pubmed_adjoint.json 2> $OUT/pubmed_adjoint.err
tail -3 $OUT/pubmed_adjoint.err
python - $OUT/pubmed_adjoint.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ['value', 'forward_ms', 'backward_ms', 'evals_forward', 'augmented_evals_backward', 'flat_host_loop', 'backward_speedup_vs_flat_host_loop']:
  print(k, d[k])
print({a: b for a, b in d['parity_vs_flat_host_loop'].items() if a != 'what'})
PY
