mkdir -p gpurun_out/r04m
python tools/debug/split_batch_probe.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids" | tee gpurun_out/r04m/split.txt
