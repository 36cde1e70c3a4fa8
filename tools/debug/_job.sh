mkdir -p gpurun_out/r04v
for i in 1 2; do for cfg in "0 0" "0 1" "1 1"; do set -- $cfg
SEGCLIP_WGRAD_SIDE=$1 SEGCLIP_MAIN_HIGH=$2 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | grep '"metric"' | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/wgrad_side=$1 main_high=$2 /" | tee -a gpurun_out/r04v/ab_side2.txt; done; done
