timeout 200 python tools/gather_locality.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03i_gather_locality.log
