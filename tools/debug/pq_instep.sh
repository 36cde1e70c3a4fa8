# in-step A/B of gemm_bf16_pq.hip (SEGCLIP_GEMM_PQ=0/1), two rounds each
mkdir -p gpurun_out/instep
for cfg in 0 1 0 1; do
  SEGCLIP_GEMM_PQ=$cfg timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PQ=$cfg', d['ms_per_step'], d['value'], d['config']['loss'])" | tee -a gpurun_out/instep/instep.log
done
