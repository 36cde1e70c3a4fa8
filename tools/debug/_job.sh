mkdir -p gpurun_out/r04u
for v in 0 1 0 1; do SEGCLIP_ATTN_BWD_SMALLWG=$v python tools/bench_attn.py 2>&1 | grep "T77" | grep -v SDPA | sed "s/^/smallwg=$v /" | tee -a gpurun_out/r04u/attn.txt; done
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention or attn or residual_block or causal" 2>&1 | grep -E "passed|failed|Error|error|assert|mismatch" | tail -4
for i in 1 2; do for v in 0 1; do SEGCLIP_ATTN_BWD_SMALLWG=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | grep '"metric"' | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/smallwg=$v /" | tee -a gpurun_out/r04u/ab.txt; done; done
