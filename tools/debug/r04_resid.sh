mkdir -p gpurun_out/r04b; export TMPDIR=/tmp; O=gpurun_out/r04b
F='^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids\|socket.cpp\|ProcessGroupNCCL'
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "residual_stream or res_stack or linear" -s 2>&1 | tail -15 > $O/pytest.txt
timeout 300 python tools/bench_pq.py 2>&1 | grep -v "$F" > $O/bench_pq.txt
for r in fp32 bf16 fp32 bf16; do
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --resid $r --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$r', d['ms_per_step'], d['value'], d['config']['loss'])" >> $O/bench_ab.txt
done
timeout 900 python tools/accuracy_b256.py 2>&1 | grep -v "$F" > $O/accuracy_b256.txt
cat $O/pytest.txt; grep -i "res\|MISMATCH\|OK" $O/bench_pq.txt; cat $O/bench_ab.txt; grep "==\|matrices\|vectors" $O/accuracy_b256.txt
