set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03x
timeout 600 python -m pytest tests/test_solver_gpu.py -m gpu -x -q -p no:cacheprovider -k "relabel" > gpurun_out/r03x/pytest_relabel.log 2>&1; echo "rc $?" >> gpurun_out/r03x/pytest_relabel.log; tail -5 gpurun_out/r03x/pytest_relabel.log
bash tools/final_evidence_r3.sh r03x d11b9fc rmat
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r03x/bench_default_nocpu.json 2> gpurun_out/r03x/bench_default_nocpu.err; echo "bench rc $?"; cut -c1-200 gpurun_out/r03x/bench_default_nocpu.json
