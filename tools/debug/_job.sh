mkdir -p gpurun_out/r04y
python tools/debug/center_stage_profile.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids\|Warning\|warn" | tee gpurun_out/r04y/center.txt | head -60
