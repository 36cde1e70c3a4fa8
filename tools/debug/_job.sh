mkdir -p gpurun_out/r04u
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "group_linear" 2>&1 | grep -E "passed|failed|Error|error|assert|mismatch" | tail -3
for gs in 1 2 3 4 6; do echo "gsplit=$gs"; SEGCLIP_GL64_GSPLIT=$gs python tools/debug/center_stage_profile.py 2>&1 | grep -E "group_linear" ; done | tee gpurun_out/r04u/gl.txt
