#!/bin/bash
# round 5, GPU call 1: the whole GPU suite, then the Cora best_params epoch on the recorded solve and on the host loop
set -u
OUT=gpurun_out/r5c1
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -30 $OUT/pytest_gpu.log
timeout 300 python bench.py --config cora-epoch --steps 20 --warmup 3 > $OUT/cora_epoch.json 2> $OUT/cora_epoch.err
GNPDE_HOST_DOPRI5_TRAINING=1 timeout 300 python bench.py --config cora-epoch --steps 10 --warmup 3 --no-cpu-baseline > $OUT/cora_epoch_host_loop.json 2> $OUT/cora_epoch_host_loop.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_cora -o p -- python bench.py --config cora-epoch --steps 10 --warmup 3 --no-cpu-baseline > $OUT/cora_epoch_prof.log 2>&1
python tools/prof_summary.py "$(find $OUT/st_cora -name '*kernel_stats.csv' | head -1)" $OUT/cora_epoch_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --config cora-epoch --steps 10 --warmup 3" > /dev/null 2>> $OUT/cora_epoch_prof.log
find $OUT -name '*kernel_trace.csv' -delete
for f in cora_epoch cora_epoch_host_loop; do
  python - $OUT/$f.json <<'PY'
import json, sys
try:
  d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  print(sys.argv[1].split('/')[-1], d.get('value'), d.get('unit'), 'train ms', d.get('ms_train_step'), 'test ms', d.get('ms_test_step'), d.get('ms_train_phases_synchronised'),
        'nfe', d.get('nfe_forward_per_epoch'), d.get('nfe_test_per_epoch'), d.get('train_solve_path'), d.get('parity_vs_restated_torchdiffeq'))
except Exception as exc:
  print(sys.argv[1], 'unreadable', exc)
  print(open(sys.argv[1].replace('.json', '.err')).read()[-1500:])
PY
done
