F='^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids\|socket.cpp\|ProcessGroupNCCL'
python tools/debug/overlap_timeline.py first 2>&1 | grep -v "$F" | cut -c1-330
python tools/debug/overlap_timeline.py none 2>&1 | grep -v "$F" | cut -c1-330
