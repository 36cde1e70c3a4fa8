mkdir -p gpurun_out/r04j; export TMPDIR=/tmp; O=gpurun_out/r04j
F='^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids\|socket.cpp\|ProcessGroupNCCL'
for t in 1 0 3 2; do echo "== SEGCLIP_PQ_TAIL=$t" >> $O/bench_pq.txt; SEGCLIP_PQ_TAIL=$t timeout 300 python tools/bench_pq.py 2>&1 | grep -v "$F" | grep -i "MISMATCH\|ALL CHECKS\|False\|K3072\|K2304\|N768 K768\|torch" >> $O/bench_pq.txt; done
run() { env $1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['value'], d['config']['loss'])" >> $O/ab.txt; }
for rep in 1 2; do for e in "SEGCLIP_PQ_TAIL=1" "SEGCLIP_PQ_TAIL=0" "SEGCLIP_PQ_TAIL=2" "SEGCLIP_PQ_TAIL=1 SEGCLIP_PQ_TAIL_MINK=12" "SEGCLIP_PQ_TAIL=1 SEGCLIP_MAIN_HIGH=1"; do run "$e"; done; done
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_bench_size_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|FAILED|^E " | head -20 > $O/pytest.txt
cat $O/bench_pq.txt $O/ab.txt $O/pytest.txt
