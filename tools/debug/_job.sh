mkdir -p gpurun_out/r04w
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert|mismatch" | tail -8 | tee gpurun_out/r04w/t_all.txt
python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-roofline --force-dist 2>&1 | grep '"metric"' | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/force-dist /" | tee -a gpurun_out/r04w/ab2.txt
python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-roofline 2>&1 | grep '"metric"' | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/plain /" | tee -a gpurun_out/r04w/ab2.txt
