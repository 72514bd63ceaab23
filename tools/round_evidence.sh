#!/bin/bash
# Evidence of the SHIPPED kernels in one short GPU call (GPU box only; run from the repo root), most valuable first so that
# a call cut off by the GPU budget still leaves the earlier items under gpurun_out/$TAG:
#   1. rocprofv3 --kernel-trace --stats of the bench command     -> kernel_stats.csv
#   2. the default bench line                                   -> bench_default.json
#   3. two PMC passes (FETCH_SIZE | WRITE_SIZE TCC_HIT TCC_MISS) -> hbm_traffic.json (record of the dominant kernel)
#   4. the GPU test suite, as far as the time goes               -> pytest.log
#   tools/round_evidence.sh TAG COMMIT [PYTEST_SECONDS]
set -u
TAG=$1; COMMIT=$2; TSEC=${3:-300}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python bench.py --graph arxiv --steps 20 --warmup 0 --no-cpu-baseline --no-roofline-probe --replays 1"
date +%s > "$OUT/t0"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- $BENCH > "$OUT/stats.log" 2>&1
python tools/prof_summary.py "$(find $OUT/stats -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv" \
  "rocprofv3 --kernel-trace --stats -- $BENCH   (commit $COMMIT)" > /dev/null 2>> "$OUT/stats.log"
find "$OUT" -name '*kernel_trace.csv' -delete
date +%s > "$OUT/t1"
timeout 200 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
date +%s > "$OUT/t2"
cp profiles/hbm_traffic.json "$OUT/hbm_traffic.json"
DIRS=""
i=0
for CNT in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/pmc_$i" -o p -- $BENCH --no-graph > "$OUT/pmc_$i.log" 2>&1
  DIRS="$DIRS $OUT/pmc_$i"
done
python tools/pmc_traffic.py "$OUT/hbm_traffic.json" "arxiv_d128_spmm" "$COMMIT" "$BENCH --no-graph" $DIRS > "$OUT/pmc_summary.log" 2>&1
find "$OUT" -name '*kernel_trace.csv' -delete
find "$OUT" -name '*counter_collection.csv' -delete
date +%s > "$OUT/t3"
timeout "$TSEC" python -m pytest tests -m gpu -x -q -p no:cacheprovider > "$OUT/pytest.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest.log"
date +%s > "$OUT/t4"
head -8 "$OUT/kernel_stats.csv"
tail -3 "$OUT/pytest.log"
