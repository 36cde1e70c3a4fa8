mkdir -p gpurun_out/r04l; export TMPDIR=/tmp; O=gpurun_out/r04l
F='^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids\|socket.cpp\|ProcessGroupNCCL'
timeout 300 python tools/bench_pq.py 2>&1 | grep -v "$F" | grep -i "act8\|MISMATCH\|ALL CHECKS\|False" > $O/bench_pq.txt
timeout 300 python tools/bench_pq.py 19712 2>&1 | grep -v "$F" | grep -i "act8\|MISMATCH\|ALL CHECKS\|False" >> $O/bench_pq.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_dist_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|FAILED|^E " | head -20 > $O/pytest.txt
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['config']['loss'])" >> $O/ab.txt; done
timeout 200 python tools/bench_attn.py 2>&1 | grep -v "$F" > $O/attn.txt
cat $O/bench_pq.txt $O/pytest.txt $O/ab.txt $O/attn.txt
