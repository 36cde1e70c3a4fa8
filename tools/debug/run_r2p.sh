F='^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids\|socket.cpp\|ProcessGroupNCCL'
python tools/debug/pg_cost.py first eager 2>&1 | grep -v "$F"
python tools/debug/pg_cost.py first lazy 2>&1 | grep -v "$F"
GPU_MAX_HW_QUEUES=8 python tools/debug/pg_cost.py first eager 2>&1 | grep -v "$F"
GPU_MAX_HW_QUEUES=2 python tools/debug/pg_cost.py first eager 2>&1 | grep -v "$F"
python tools/debug/pg_cost.py after eager 2>&1 | grep -v "$F"
