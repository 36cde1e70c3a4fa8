mkdir -p gpurun_out/r4e
for cfg in "0 0" "1 0" "1 1" "0 0" "1 0" "1 1"; do set -- $cfg
  SEGCLIP_GEMM_PQ=$1 SEGCLIP_PQ_PERSIST=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PQ=$1 PERSIST=$2', d['ms_per_step'], d['value'], d['config']['loss'])" | tee -a gpurun_out/r4e/instep.log
done
