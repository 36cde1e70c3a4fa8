mkdir -p gpurun_out/r04z
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "linear" 2>&1 | grep -E "passed|failed|Error|error|assert|mismatch" | tail -8 | tee gpurun_out/r04z/t_k.txt
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|mismatch|no tests" | tail -8 | tee gpurun_out/r04z/t_model.txt
python tools/debug/center_stage_profile.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids\|Warning\|warn" | tee gpurun_out/r04z/center3.txt | head -24
python bench.py --steps 30 --warmup 8 2>&1 | grep '"metric"' | tee gpurun_out/r04z/bench3.json | cut -c1-200
