mkdir -p gpurun_out/r04c; export TMPDIR=/tmp; O=gpurun_out/r04c
F='^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids\|socket.cpp\|ProcessGroupNCCL'
timeout 300 python tools/bench_pq.py 2>&1 | grep -v "$F" > $O/bench_pq.txt
timeout 300 python tools/bench_pq.py 19712 2>&1 | grep -v "$F" > $O/bench_pq_text.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bench_size_gpu.py -x -q 2>&1 | tail -5 > $O/pytest.txt
for e in "SEGCLIP_GEMM_PQ_WGRAD=0 SEGCLIP_GEMM_PQ_RES32=0" "SEGCLIP_GEMM_PQ_WGRAD=1 SEGCLIP_GEMM_PQ_RES32=0" "SEGCLIP_GEMM_PQ_WGRAD=0 SEGCLIP_GEMM_PQ_RES32=1" "SEGCLIP_GEMM_PQ_WGRAD=1 SEGCLIP_GEMM_PQ_RES32=1" "SEGCLIP_GEMM_PQ_WGRAD=0 SEGCLIP_GEMM_PQ_RES32=0" "SEGCLIP_GEMM_PQ_WGRAD=1 SEGCLIP_GEMM_PQ_RES32=1"; do
  env $e timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$e', d['ms_per_step'], d['value'], d['config']['loss'])" >> $O/bench_ab.txt
done
grep -i "wgr\|res\|MISMATCH\|OK" $O/bench_pq.txt; grep -i "wgr\|res\|MISMATCH\|OK\|torch" $O/bench_pq_text.txt; cat $O/pytest.txt $O/bench_ab.txt
