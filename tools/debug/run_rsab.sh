mkdir -p gpurun_out/r02d
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r02d/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r02d/pytest_gpu.log | tail -3
timeout 300 python bench.py --no-cpu-baseline --force-dist 2>/dev/null | cut -c1-600
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-330
timeout 300 python bench.py --no-cpu-baseline --full-loss 2>/dev/null | cut -c1-330
