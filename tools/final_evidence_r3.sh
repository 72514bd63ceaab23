#!/bin/bash
# Round-3 evidence at HEAD (GPU box only; run from the repo root):   tools/final_evidence_r3.sh TAG COMMIT
#   arxiv: rocprofv3 kernel stats of the solver's own launches, the default bench line (with the CPU baseline), two PMC passes
#          (with the roofline probes, so that the gather-ceiling kernel is counted too) -> hbm_traffic.json
#   rmat : kernel stats, two PMC passes -> hbm_traffic.json, bench line
#   third argument: the graphs to cover (default "arxiv rmat")
set -u
TAG=$1; COMMIT=$2; GRAPHS=${3:-"arxiv rmat"}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
cp profiles/hbm_traffic.json "$OUT/hbm_traffic.json"
for G in $GRAPHS; do
  if [ $G = arxiv ]; then STEPS=20; PROBE=""; else STEPS=2; PROBE="--no-roofline-probe"; fi
  BENCH="python bench.py --graph $G --steps $STEPS --warmup 0 --no-cpu-baseline --no-roofline-probe --replays 1"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$G" -o p -- $BENCH > "$OUT/stats_$G.log" 2>&1
  python tools/prof_summary.py "$(find $OUT/stats_$G -name '*kernel_stats.csv' | head -1)" "$OUT/${G}_kernel_stats.csv" \
    "rocprofv3 --kernel-trace --stats -- $BENCH   (commit $COMMIT)" > /dev/null 2>> "$OUT/stats_$G.log"
  head -9 "$OUT/${G}_kernel_stats.csv"
  PB="python bench.py --graph $G --steps $STEPS --warmup 0 --no-cpu-baseline $PROBE --replays 1 --no-graph"
  DIRS=""; i=0
  for CNT in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 500 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/pmc_${G}_$i" -o p -- $PB > "$OUT/pmc_${G}_$i.log" 2>&1
    DIRS="$DIRS $OUT/pmc_${G}_$i"
  done
  D=$(python -c "import sys; sys.path.insert(0,'.'); import gnpde_amd as G; print(G.synthetic.CONFIGS['$G']['d'])")
  python tools/pmc_traffic.py "$OUT/hbm_traffic.json" "${G}_d${D}_spmm" "$COMMIT" "$PB" $DIRS > "$OUT/pmc_summary_$G.log" 2>&1
  tail -25 "$OUT/pmc_summary_$G.log" | head -40
  find "$OUT" -name '*kernel_trace.csv' -delete
  find "$OUT" -name '*counter_collection.csv' -delete
done
cp "$OUT/hbm_traffic.json" profiles/hbm_traffic.json     # (so that the bench lines below carry the fresh record)
case "$GRAPHS" in *arxiv*) timeout 300 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc $?"; cut -c1-300 "$OUT/bench_default.json";; esac
case "$GRAPHS" in *rmat*) timeout 400 python bench.py --graph rmat --steps 8 --warmup 1 > "$OUT/bench_rmat.json" 2> "$OUT/bench_rmat.err"; echo "bench rmat rc $?"; cut -c1-300 "$OUT/bench_rmat.json";; esac
