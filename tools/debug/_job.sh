mkdir -p gpurun_out/r04p
for B in 64 256; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 --batch $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('B=$B', d['ms_per_step'], d['value'], d['config']['loss'])"; done | tee gpurun_out/r04p/b.txt
B=64 python tools/debug/host_lead.py 2>&1 | grep "host fwd" | tail -3 | tee -a gpurun_out/r04p/b.txt
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_streams_gpu.py tests/test_dist_gpu.py tests/test_train_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|FAILED|^E " | head -10 | tee -a gpurun_out/r04p/b.txt
