mkdir -p gpurun_out/r04q; python tools/debug/host_bwd_profile.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids" | tee gpurun_out/r04q/bwd_host.txt | head -60
