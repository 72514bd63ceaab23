#!/bin/bash
# The HBM-bound shape (R-MAT 2^21 nodes, 77 M entries, d = 256) in one GPU call (GPU box only; run from the repo root):
#   1. bench line with the row deal the graph builder picks (hashed blocks for this graph)   -> bench_rmat.json
#   2. the same with contiguous eighths forced (GNPDE_TUNE=10=1): the A/B of csrc/spmm.hip's rows -> XCD deal
#   3. kernel stats + PMC traffic of the shipped configuration (tools/profile_round.sh)
#   tools/round_evidence_rmat.sh TAG COMMIT
# Every bench run spends ~70 s generating the graph on the host; budget ~8 minutes for the whole script.
set -u
TAG=$1; COMMIT=$2
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 400 python bench.py --graph rmat --steps 8 --warmup 1 > "$OUT/bench_rmat.json" 2> "$OUT/bench_rmat.err"
GNPDE_TUNE=10=1 timeout 400 python bench.py --graph rmat --steps 8 --warmup 1 > "$OUT/bench_rmat_contiguous_eighths.json" 2> "$OUT/bench_rmat_contiguous_eighths.err"
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
for name in ('bench_rmat', 'bench_rmat_contiguous_eighths'):
  try:
    d = json.loads(open('%s/%s.json' % (out, name)).read().strip().split('\n')[-1])
    print(name, d['value'], 'steps/s', d['ms_per_step'], 'ms/step; aggregation', d['roofline']['avg_launch_us'], 'us',
          d['roofline']['achieved'], 'GB/s;', d['config'].get('xcd_row_deal'), d['config'].get('xcd_contiguous_imbalance'))
  except Exception as exc:
    print(name, 'unreadable:', exc)
PY
bash tools/profile_round.sh "$TAG/profile" "$COMMIT" rmat 2
