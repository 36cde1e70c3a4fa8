F='^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids\|socket.cpp\|ProcessGroupNCCL'
for m in never before after never before after; do python tools/debug/hwq_env_probe.py $m 2>&1 | grep "ms/step"; done
