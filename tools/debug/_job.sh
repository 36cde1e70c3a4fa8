mkdir -p gpurun_out/r04u
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "cross" 2>&1 | grep -E "passed|failed|Error|error|assert|mismatch" | tail -4
python tools/debug/center_stage_profile.py 2>&1 | grep -E "total kernel|smallq" | tee gpurun_out/r04u/center.txt
