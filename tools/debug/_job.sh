mkdir -p gpurun_out/r04u
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "residual_table or patch_embedding or im2col or linear or bench_gemm" 2>&1 | grep -E "passed|failed|Error|error|assert|mismatch" | tail -6
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_bench_size_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|mismatch" | tail -4
SEGCLIP_BENCH_PROFILE_DIR=gpurun_out/r04u/rl python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>&1 | grep '"metric"' | cut -c1-220
grep -E "p8_kernel<false, false|pq_kernel<false, false, 5" gpurun_out/r04u/rl/kernel_stats.txt | cut -c1-150
rm -rf gpurun_out/r04u/rl/trace
