mkdir -p gpurun_out/r04d; export TMPDIR=/tmp; O=gpurun_out/r04d
F='^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids\|socket.cpp\|ProcessGroupNCCL'
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -5 > $O/pytest.txt
timeout 200 python tools/bench_attn.py 2>&1 | grep -v "$F" > $O/attn.txt
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['config']['loss'])" >> $O/bench_ab.txt
done
cat $O/pytest.txt $O/attn.txt $O/bench_ab.txt
