mkdir -p gpurun_out/r04a; export TMPDIR=/tmp
F='^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids\|socket.cpp\|ProcessGroupNCCL'
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04a/pytest.txt
SEGCLIP_BENCH_PROFILE_DIR=gpurun_out/r04a/rl timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04a/bench_line.json 2> gpurun_out/r04a/bench_line.err
cp gpurun_out/r04a/rl/kernel_stats.txt gpurun_out/r04a/kernel_stats.txt
python tools/stream_gaps.py $(ls gpurun_out/r04a/rl/trace/*.db gpurun_out/r04a/rl/trace/*/*.db 2>/dev/null | head -1) 120 > gpurun_out/r04a/stream_gaps.txt 2>&1
rm -rf gpurun_out/r04a/rl/trace gpurun_out/r04a/rl/pmc_*
timeout 300 python tools/bench_gemm.py 2>&1 | grep -v "$F" > gpurun_out/r04a/gemm_shapes.txt
timeout 300 python tools/bench_pq.py 2>&1 | grep -v "$F" > gpurun_out/r04a/bench_pq.txt
timeout 200 python tools/bench_attn.py 2>&1 | grep -v "$F" > gpurun_out/r04a/attn.txt
timeout 300 python tools/bench_hbm.py 2>&1 | grep -v "$F" > gpurun_out/r04a/hbm_kernels.txt
cat gpurun_out/r04a/pytest.txt; head -40 gpurun_out/r04a/kernel_stats.txt | cut -c1-160; tail -c 1500 gpurun_out/r04a/bench_line.json
