mkdir -p gpurun_out/r04u
b() { python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | grep '"metric"' | grep -o '"ms_per_step": [0-9.]*'; }
for i in 1 2; do
echo "default $(b)" | tee -a gpurun_out/r04u/ab_env.txt
echo "HSA_ENABLE_INTERRUPT=0 $(HSA_ENABLE_INTERRUPT=0 b)" | tee -a gpurun_out/r04u/ab_env.txt
echo "GPU_MAX_HW_QUEUES=4 $(GPU_MAX_HW_QUEUES=4 b)" | tee -a gpurun_out/r04u/ab_env.txt
echo "GPU_MAX_HW_QUEUES=2 $(GPU_MAX_HW_QUEUES=2 b)" | tee -a gpurun_out/r04u/ab_env.txt
echo "HSA_ENABLE_SDMA=0 $(HSA_ENABLE_SDMA=0 b)" | tee -a gpurun_out/r04u/ab_env.txt
done
