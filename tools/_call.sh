set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03c
timeout 600 python tools/reorder_probe_rmat.py > gpurun_out/r03c/reorder_probe_rmat.txt 2>&1; echo "rc $?" >> gpurun_out/r03c/reorder_probe_rmat.txt
tail -6 gpurun_out/r03c/reorder_probe_rmat.txt | cut -c1-400
