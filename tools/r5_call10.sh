#!/bin/bash
set -u
OUT=gpurun_out/r5c10
mkdir -p $OUT
timeout 400 python -m pytest tests/test_adjoint_native_gpu.py tests/test_adjoint_gpu.py -m gpu -x -q 2>&1 | tail -4
bash tools/r5_prof_adjoint.sh > $OUT/prof.log 2>&1
tail -12 $OUT/prof.log | cut -c1-120
for cfg in pubmed-adjoint coauthor-adjoint; do timeout 200 python bench.py --config $cfg 2>/dev/null | tail -1 > $OUT/$cfg.json; python - <<PY
import json; d=json.load(open('$OUT/$cfg.json')); print('$cfg', {k:d[k] for k in d if k in ('value','ms_per_step')}, json.dumps(d.get('config',{}))[:600])
PY
done
