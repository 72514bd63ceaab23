timeout 200 python tools/agg_streams.py arxiv 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03o_agg_streams.log
