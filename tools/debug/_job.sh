mkdir -p gpurun_out/r04u
for i in 1 2; do
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | grep '"metric"' | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/package default (1) /" | tee -a gpurun_out/r04u/ab_kernarg2.txt
HIP_FORCE_DEV_KERNARG=0 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>&1 | grep '"metric"' | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/dev_kernarg=0 /" | tee -a gpurun_out/r04u/ab_kernarg2.txt
done
python - <<'PY' 2>&1 | tail -2
import os, subprocess, sys
env = dict(os.environ); env.pop("HIP_FORCE_DEV_KERNARG", None)
code = "import os, torch; torch.zeros(1, device='cuda'); import segclip_amd; print('env after late import:', os.environ.get('HIP_FORCE_DEV_KERNARG'))"
print(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip())
PY
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_streams_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -3
