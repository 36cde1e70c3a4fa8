mkdir -p gpurun_out/r04t
for i in 1 2; do for v in 0 1; do SEGCLIP_CROSS_FUSED=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --batch 64 2>&1 | grep '"metric"' | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/B64 cross_fused=$v /" | tee -a gpurun_out/r04t/ab64.txt; done; done
