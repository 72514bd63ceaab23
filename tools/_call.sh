set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03c
timeout 300 python tools/_probe_att.py > gpurun_out/r03c/probe_att.txt 2>&1; echo "rc $?" >> gpurun_out/r03c/probe_att.txt
tail -14 gpurun_out/r03c/probe_att.txt | cut -c1-200
