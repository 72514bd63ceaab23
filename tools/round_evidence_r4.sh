#!/bin/bash
# Round-4 evidence of the SHIPPED code in one GPU call (GPU box only; run from the repo root), most valuable first:
#   tools/round_evidence_r4.sh TAG COMMIT
# Writes gpurun_out/$TAG/*: kernel stats + bench lines of the headline, R-MAT, training, C4, Cora and normaliser variants, the PMC
# record of the aggregation (hbm_traffic.json), then the GPU test suite.
set -u
TAG=$1; COMMIT=$2
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
prof() {  # name, command...
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/st_$name" -o p -- "$@" > "$OUT/$name.stats.log" 2>&1
  python tools/prof_summary.py "$(find $OUT/st_$name -name '*kernel_stats.csv' | head -1)" "$OUT/${name}_kernel_stats.csv" \
    "rocprofv3 --kernel-trace --stats -- $*   (commit $COMMIT)" > /dev/null 2>> "$OUT/$name.stats.log"
  find "$OUT/st_$name" -name '*kernel_trace.csv' -delete
}
B="python bench.py"
prof arxiv_steps20 $B --graph arxiv --steps 20 --warmup 0 --no-cpu-baseline --no-roofline-probe --replays 1
timeout 400 $B --steps 20 --warmup 5 > "$OUT/bench_default_steps20.json" 2> "$OUT/bench_default.err"
timeout 400 $B --steps 100 --warmup 10 --no-cpu-baseline > "$OUT/bench_steps100.json" 2>> "$OUT/bench_default.err"
cp profiles/hbm_traffic.json "$OUT/hbm_traffic.json"
DIRS=""
i=0
PB="$B --graph arxiv --steps 20 --warmup 0 --no-cpu-baseline --no-roofline-probe --replays 1 --no-graph"
for CNT in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/pmc_$i" -o p -- $PB > "$OUT/pmc_$i.log" 2>&1
  DIRS="$DIRS $OUT/pmc_$i"
done
python tools/pmc_traffic.py "$OUT/hbm_traffic.json" "arxiv_d128_spmm" "$COMMIT" "$PB" $DIRS > "$OUT/pmc_summary.log" 2>&1
find "$OUT" -name '*kernel_trace.csv' -delete; find "$OUT" -name '*counter_collection.csv' -delete
prof train $B --train --steps 10 --warmup 2 --replays 3
timeout 300 $B --train --steps 10 --warmup 2 > "$OUT/bench_train.json" 2> "$OUT/bench_train.err"
prof c4 $B --config c4 --warmup 1 --replays 3 --no-cpu-baseline
timeout 400 $B --config c4 --warmup 2 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"
for f in laplacian transformer; do
  timeout 100 $B --graph cora --function $f --steps 100 --warmup 10 --no-live-pmc --no-hbm-probe > "$OUT/bench_cora_$f.json" 2> "$OUT/bench_cora.err"
done
timeout 100 $B --graph cora --function transformer --square-plus --norm-idx 1 --steps 100 --warmup 10 --no-cpu-baseline --no-live-pmc --no-hbm-probe > "$OUT/bench_cora_transformer_as_run.json" 2>> "$OUT/bench_cora.err"
prof cora_transformer $B --graph cora --function transformer --steps 100 --warmup 0 --no-cpu-baseline --no-roofline-probe --replays 1
for v in "--norm-idx 1" "--square-plus" "--norm-idx 1 --square-plus"; do
  timeout 200 $B --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc --no-hbm-probe $v > "$OUT/bench_arxiv${v// /_}.json" 2>> "$OUT/bench_default.err"
done
timeout 500 $B --graph rmat --steps 8 --warmup 1 --no-live-pmc --no-hbm-probe > "$OUT/bench_rmat_steps8.json" 2> "$OUT/bench_rmat.err"
prof rmat_steps2 $B --graph rmat --steps 2 --warmup 0 --no-cpu-baseline --no-roofline-probe --replays 1
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_gpu.log"
tail -3 "$OUT/pytest_gpu.log"
for f in bench_default_steps20 bench_steps100 bench_train bench_c4 bench_cora_laplacian bench_cora_transformer bench_cora_transformer_as_run bench_rmat_steps8 "bench_arxiv--norm-idx_1" "bench_arxiv--square-plus" "bench_arxiv--norm-idx_1_--square-plus"; do
  python - "$OUT/$f.json" <<'PY'
import json, sys
try:
  d = json.loads(open(sys.argv[1]).read())
  r = d.get('roofline') or {}
  print(sys.argv[1].split('/')[-1], d.get('value'), d.get('unit'), 'ms/step', d.get('ms_per_step'), 'frac', r.get('frac'), 'alg', r.get('frac_algorithmic'), 'hbm_probe', (r.get('hbm_bound_probe') or {}).get('frac'))
except Exception as exc:
  print(sys.argv[1], 'unreadable', exc)
PY
done
