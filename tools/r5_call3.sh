#!/bin/bash
# round 5, GPU call 3: full suite after the row-order / padding-cache / refresh changes; cora-epoch + c4 lines; launch sequences of the
# Cora GRAND-nl evaluation as run_GNN.py runs it and of one recorded training epoch
set -u
OUT=gpurun_out/r5c3
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout 300 python bench.py --config cora-epoch --steps 20 --warmup 3 > $OUT/cora_epoch.json 2> $OUT/cora_epoch.err
timeout 400 python bench.py --config c4 --warmup 2 > $OUT/c4.json 2> $OUT/c4.err
timeout 120 python bench.py --graph cora --steps 100 --warmup 10 --square-plus --norm-idx 1 --no-live-pmc --no-hbm-probe > $OUT/cora_as_run.json 2> $OUT/cora_as_run.err
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_c2 -o p -- python bench.py --graph cora --steps 20 --warmup 0 --square-plus --norm-idx 1 --no-cpu-baseline --no-roofline-probe --replays 1 > $OUT/tr_c2.log 2>&1
python tools/trace_sequence.py "$(find $OUT/tr_c2 -name '*kernel_trace.csv' | head -1)" linear_kernel -1 24 > $OUT/c2_as_run_sequence.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_ep -o p -- python bench.py --config cora-epoch --steps 4 --warmup 3 --no-cpu-baseline > $OUT/tr_ep.log 2>&1
python tools/trace_sequence.py "$(find $OUT/tr_ep -name '*kernel_trace.csv' | head -1)" tape_dots_fold -1 60 > $OUT/cora_epoch_sequence_after_backward.txt 2>&1
python tools/trace_sequence.py "$(find $OUT/tr_ep -name '*kernel_trace.csv' | head -1)" tape_store -1 80 > $OUT/cora_epoch_sequence_forward.txt 2>&1
find $OUT -name '*kernel_trace.csv' -delete
cat $OUT/c2_as_run_sequence.txt
for f in cora_epoch c4 cora_as_run; do
  python - $OUT/$f.json <<'PY'
import json, sys
try:
  d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  r = d.get('roofline') or {}
  print(sys.argv[1].split('/')[-1], d.get('value'), d.get('unit'), 'ms/step', d.get('ms_per_step'), 'frac', r.get('frac'), 'alg', r.get('frac_algorithmic'),
        {k: d[k] for k in d if k.startswith('parity') or k.startswith('ms_') or k.startswith('nfe')})
except Exception as exc:
  print(sys.argv[1], 'unreadable', exc)
  print(open(sys.argv[1].replace('.json', '.err')).read()[-1200:])
PY
done
