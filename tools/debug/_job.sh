mkdir -p gpurun_out/r04v
for i in 1 2; do for v in 0 1; do SEGCLIP_WGRAD_SIDE=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | grep '"metric"' | grep -o '"ms_per_step": [0-9.]*\|"loss": [0-9.]*' | tr '\n' ' ' | sed "s/^/wgrad_side=$v /" | tee -a gpurun_out/r04v/ab_side.txt; echo | tee -a gpurun_out/r04v/ab_side.txt; done; done
SEGCLIP_WGRAD_SIDE=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "grouped or res_stack" 2>&1 | grep -E "passed|failed|Error|error|assert|mismatch" | tail -4
SEGCLIP_WGRAD_SIDE=1 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_bench_size_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|mismatch" | tail -4
