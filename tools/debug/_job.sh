mkdir -p gpurun_out/r04w
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "grouped or res_stack or deferred or bench_gemm" 2>&1 | grep -E "passed|failed|Error|error|assert|mismatch" | tail -8 | tee gpurun_out/r04w/t_k.txt
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|mismatch|no tests" | tail -8 | tee gpurun_out/r04w/t_model.txt
for g in 1 12 1 12; do SEGCLIP_WGRAD_GROUP=$g python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | grep '"metric"' | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/group=$g /" | tee -a gpurun_out/r04w/ab.txt; done
