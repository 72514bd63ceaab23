#!/bin/bash
# First GPU call of a round (GPU box only; run from the repo root): everything that was committed without a measurement, most
# valuable first, every part under its own timeout and with its output under gpurun_out/$TAG so that a cut-off call keeps
# what it had.   tools/first_call.sh TAG COMMIT          (~14 minutes)
#   1. GPU suite                                   -> pytest.log
#   2. arxiv: kernel stats, default bench line, PMC traffic of the shipped kernels (tools/round_evidence.sh without the suite)
#   3. A/B checks: rows -> XCD deal (bit identity + timing), LDS-staged hub fold (bit identity + timing)
#   4. 2- and 4-rank functional runs of the partitioned bench on this one GPU (transport ladder, two-step and timed-solve parity)
#   5. R-MAT: bench with the picked deal and with contiguous eighths forced, kernel stats, PMC (tools/round_evidence_rmat.sh)
set -u
TAG=$1; COMMIT=$2
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 420 python -m pytest tests -m gpu -x -q -p no:cacheprovider > "$OUT/pytest.log" 2>&1; echo "pytest rc $?" >> "$OUT/pytest.log"
tail -3 "$OUT/pytest.log"
bash tools/round_evidence.sh "$TAG/arxiv" "$COMMIT" 1 > "$OUT/arxiv.log" 2>&1
head -8 "$OUT/arxiv/kernel_stats.csv"
timeout 60 python tools/xcd_check.py > "$OUT/xcd_check.log" 2>&1; echo "rc $?" >> "$OUT/xcd_check.log"; tail -4 "$OUT/xcd_check.log"
timeout 60 python tools/hub_fold_ab.py arxiv > "$OUT/hub_fold_ab.log" 2>&1; echo "rc $?" >> "$OUT/hub_fold_ab.log"; tail -2 "$OUT/hub_fold_ab.log"
for W in 2 4; do
  GNPDE_RANKS_SHARE_DEVICE=1 MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=4 timeout 120 python -m torch.distributed.run --nnodes=1 \
    --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((29700 + W)) bench.py --gpus $W --steps 10 --warmup 2 \
    > "$OUT/bench_${W}ranks_one_gpu.log" 2>&1
  echo "rc $?" >> "$OUT/bench_${W}ranks_one_gpu.log"
  tail -2 "$OUT/bench_${W}ranks_one_gpu.log" | cut -c1-600
done
bash tools/round_evidence_rmat.sh "$TAG/rmat" "$COMMIT" > "$OUT/rmat.log" 2>&1
tail -20 "$OUT/rmat.log"
