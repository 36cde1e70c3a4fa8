mkdir -p gpurun_out/r04v
for i in 1 2 3; do for v in 0 1; do SEGCLIP_SHARED_RQ=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | grep '"metric"' | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/shared_rq=$v /" | tee -a gpurun_out/r04v/ab2.txt; done; done
