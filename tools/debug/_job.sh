mkdir -p gpurun_out/r04n; python tools/debug/copy_sites.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids" > gpurun_out/r04n/copy_sites.txt; head -70 gpurun_out/r04n/copy_sites.txt
