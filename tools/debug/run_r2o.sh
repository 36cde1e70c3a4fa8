mkdir -p gpurun_out/r2o
F='^RCCL\|^HIP\|^ROCm\|Hostname\|Librccl\|amdgpu.ids\|socket.cpp'
python tools/debug/stream_probe.py 2>&1 | grep -v "$F"
python tools/debug/stream_probe.py pg 2>&1 | grep -v "$F"
GPU_MAX_HW_QUEUES=8 python tools/debug/stream_probe.py pg 2>&1 | grep -v "$F"
timeout 300 python tools/debug/gradsync_cost.py 2>&1 | grep "ms/step"
timeout 300 python bench.py --no-cpu-baseline --force-dist > gpurun_out/r2o/bench_dist.json 2> gpurun_out/r2o/bench_dist.err; cut -c1-330 gpurun_out/r2o/bench_dist.json
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2o/bench1.json 2> gpurun_out/r2o/bench1.err; cut -c1-330 gpurun_out/r2o/bench1.json
timeout 300 python tools/bench_hbm.py > gpurun_out/r2o/hbm.txt 2>&1; cat gpurun_out/r2o/hbm.txt | grep -v "$F"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2o/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2o/pytest.log
bash tools/pmc_attn.sh gpurun_out/r2o/pmc_attn 2>&1 | tail -70
