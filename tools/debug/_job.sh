mkdir -p gpurun_out/r04v
for i in 1 2; do for v in 1 0; do SEGCLIP_OVERLAP_TOWERS=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | grep '"metric"' | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/overlap_towers=$v /" | tee -a gpurun_out/r04v/ab_towers.txt; done; done
