mkdir -p gpurun_out/r04z
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "layernorm or affine or layer_norm" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 | tee gpurun_out/r04z/t_ln.txt
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_api_gpu.py tests/test_vitl14_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert|no tests" | tail -8 | tee gpurun_out/r04z/t_model.txt
python tools/debug/center_stage_profile.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids\|Warning\|warn" | tee gpurun_out/r04z/center.txt | head -50
python bench.py --steps 30 --warmup 8 2>&1 | grep '"metric"' | tee gpurun_out/r04z/bench.json | cut -c1-200
