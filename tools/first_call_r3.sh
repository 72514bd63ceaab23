#!/bin/bash
# Round 3, first GPU call: time what round 2 shipped without a measurement.  No PMC passes here (the kernels change this
# round; counters are collected at the round's final HEAD by tools/round_evidence*.sh).
#   tools/first_call_r3.sh TAG COMMIT
set -u
TAG=$1; COMMIT=$2
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 420 python -m pytest tests -m gpu -x -q -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest.log"
tail -3 "$OUT/pytest.log"
BENCH="python bench.py --graph arxiv --steps 20 --warmup 0 --no-cpu-baseline --no-roofline-probe --replays 1"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- $BENCH > "$OUT/stats.log" 2>&1
python tools/prof_summary.py "$(find $OUT/stats -name '*kernel_stats.csv' | head -1)" "$OUT/arxiv_kernel_stats.csv" \
  "rocprofv3 --kernel-trace --stats -- $BENCH   (commit $COMMIT)" > /dev/null 2>> "$OUT/stats.log"
find "$OUT" -name '*kernel_trace.csv' -delete
head -10 "$OUT/arxiv_kernel_stats.csv"
timeout 200 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; cut -c1-400 "$OUT/bench_default.json"
timeout 60 python tools/xcd_check.py > "$OUT/xcd_check.log" 2>&1; echo "rc $?" >> "$OUT/xcd_check.log"; tail -4 "$OUT/xcd_check.log"
timeout 60 python tools/hub_fold_ab.py arxiv > "$OUT/hub_fold_ab.log" 2>&1; echo "rc $?" >> "$OUT/hub_fold_ab.log"; tail -3 "$OUT/hub_fold_ab.log"
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
BENCH="python bench.py --graph rmat --steps 2 --warmup 0 --no-cpu-baseline --no-roofline-probe --replays 1"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_rmat" -o p -- $BENCH > "$OUT/stats_rmat.log" 2>&1
python tools/prof_summary.py "$(find $OUT/stats_rmat -name '*kernel_stats.csv' | head -1)" "$OUT/rmat_kernel_stats.csv" \
  "rocprofv3 --kernel-trace --stats -- $BENCH   (commit $COMMIT)" > /dev/null 2>> "$OUT/stats_rmat.log"
find "$OUT" -name '*kernel_trace.csv' -delete
head -10 "$OUT/rmat_kernel_stats.csv"
timeout 200 python tools/hub_fold_ab.py rmat > "$OUT/hub_fold_ab_rmat.log" 2>&1; echo "rc $?" >> "$OUT/hub_fold_ab_rmat.log"; tail -3 "$OUT/hub_fold_ab_rmat.log"
