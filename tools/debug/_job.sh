mkdir -p gpurun_out/r04t
b() { python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | grep '"metric"' | grep -o '"ms_per_step": [0-9.]*'; }
for i in 1 2 3; do
echo "default $(b)" | tee -a gpurun_out/r04t/ab_scratch.txt
echo "HSA_NO_SCRATCH_RECLAIM=1 $(HSA_NO_SCRATCH_RECLAIM=1 b)" | tee -a gpurun_out/r04t/ab_scratch.txt
done
