#!/bin/bash
# Kernel-time summary + PMC traffic of one bench configuration (GPU box only; run from the repo root).
#   tools/profile_round.sh TAG COMMIT GRAPH STEPS [extra bench args]
# Writes gpurun_out/$TAG/{stats,pmc_*}/ and gpurun_out/$TAG/hbm_traffic.json (copy the summaries to profiles/).
# Counters are collected in their own passes with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
TAG=$1; COMMIT=$2; GRAPH=$3; STEPS=$4; shift 4
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python bench.py --graph $GRAPH --steps $STEPS --warmup 0 --no-cpu-baseline --no-roofline-probe --replays 1 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- $BENCH > "$OUT/stats.log" 2>&1
python tools/prof_summary.py "$(find $OUT/stats -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv" \
  "rocprofv3 --kernel-trace --stats -- $BENCH   (commit $COMMIT)" > /dev/null
D=$(python -c "import sys; sys.path.insert(0,'.'); import gnpde_amd as G; print(G.synthetic.CONFIGS['$GRAPH']['d'])")
DIRS=""
i=0
for CNT in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/pmc_$i" -o p -- $BENCH --no-graph > "$OUT/pmc_$i.log" 2>&1
  DIRS="$DIRS $OUT/pmc_$i"
done
python tools/pmc_traffic.py "$OUT/hbm_traffic.json" "${GRAPH}_d${D}_spmm" "$COMMIT" "$BENCH --no-graph" $DIRS > "$OUT/pmc_summary.log" 2>&1
# raw traces are large: keep the summaries only
find "$OUT" -name '*kernel_trace.csv' -delete
find "$OUT" -name '*counter_collection.csv' -size +20M -delete
head -12 "$OUT/kernel_stats.csv"
cat "$OUT/pmc_summary.log" | head -40
