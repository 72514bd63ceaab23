#!/bin/bash
# Round-6 evidence of the SHIPPED code in one GPU call (GPU box only; run from the repo root), most valuable first:
#   tools/round_evidence_r6.sh TAG COMMIT
# Writes gpurun_out/$TAG/*: the driver's command (three stdout lines: bench_detail, bench_configs, the compact headline LAST; raw live-PMC
# CSVs kept), kernel stats of the headline / training (adjoint on and off) / C4 / R-MAT runs, the T = 100 line, smoke, the GPU suite.
set -u
TAG=$1; COMMIT=$2
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
prof() {  # name, command...
  local name=$1; shift
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/st_$name" -o p -- "$@" > "$OUT/$name.stats.log" 2>&1
  python tools/prof_summary.py "$(find $OUT/st_$name -name '*kernel_stats.csv' | head -1)" "$OUT/${name}_kernel_stats.csv" \
    "rocprofv3 --kernel-trace --stats -- $*   (commit $COMMIT)" > /dev/null 2>> "$OUT/$name.stats.log"
  rm -rf "$OUT/st_$name"
}
B="python bench.py"
T0=$(date +%s)
timeout 1700 $B --gpus 1 --steps 20 --warmup 5 --keep-pmc "$OUT/live_pmc" > "$OUT/bench_default_steps20.out" 2> "$OUT/bench_default.err"
echo "default bench rc $? seconds $(( $(date +%s) - T0 ))" | tee "$OUT/bench_default.time"
tail -c 8192 "$OUT/bench_default_steps20.out" | tail -1 > "$OUT/bench_default_headline.json"
python -c "
import json; d=json.loads(open('$OUT/bench_default_headline.json').read()); print('headline bytes', len(json.dumps(d)), d['value'], d['roofline']['frac'], d['cpu_baseline']['value']); print(json.dumps(d.get('configs_summary')))"
timeout 400 $B --steps 100 --warmup 10 --no-cpu-baseline --no-configs > "$OUT/bench_steps100.out" 2>> "$OUT/bench_default.err"
tail -1 "$OUT/bench_steps100.out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('T=100', d['value'], d['ms_per_step'])"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1; tail -1 "$OUT/smoke.txt"
prof arxiv_steps20 $B --graph arxiv --steps 20 --warmup 0 --no-cpu-baseline --no-roofline-probe --replays 1
prof train $B --train --steps 10 --warmup 2 --replays 3 --no-live-pmc --no-cpu-baseline
prof train_no_adjoint $B --train --no-adjoint --steps 10 --warmup 2 --replays 3
prof c4 $B --config c4 --warmup 1 --replays 3 --no-cpu-baseline --no-live-pmc
prof rmat $B --graph rmat --steps 4 --warmup 0 --no-cpu-baseline --no-roofline-probe --replays 1
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc $?" >> "$OUT/pytest_gpu.log"
tail -3 "$OUT/pytest_gpu.log"
