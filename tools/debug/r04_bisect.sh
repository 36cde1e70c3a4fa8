mkdir -p gpurun_out/r04g; export TMPDIR=/tmp; O=gpurun_out/r04g
for e in "X=1" "SEGCLIP_ATTN_FWD_LEAN=0" "SEGCLIP_GL64=0" "SEGCLIP_GEMM_PQ_WGRAD=0" "SEGCLIP_GEMM_PQ_RES32=0" "SEGCLIP_ATTN_FWD_LEAN=0 SEGCLIP_GL64=0 SEGCLIP_GEMM_PQ_WGRAD=0 SEGCLIP_GEMM_PQ_RES32=0"; do
  echo "== $e" >> $O/bisect.txt
  env $e timeout 600 python -m pytest tests/test_vitl14_gpu.py -x -q -s 2>&1 | grep -E "vitl14_336 B|passed|failed" >> $O/bisect.txt
done
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_api_gpu.py tests/test_dist_gpu.py tests/test_train_gpu.py tests/test_textmae_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|FAILED|assert|^E " | head -30 > $O/pytest.txt
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/trc -o trc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 > /tmp/trc.log 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY' > gpurun_out/r04g/kstats.txt
import csv,glob
f=glob.glob('/tmp/trc/**/*kernel_stats.csv', recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:45]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6/8:9.3f} ms/step avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['config']['loss'])" >> $O/bench_ab.txt
done
cat $O/bisect.txt $O/pytest.txt $O/bench_ab.txt; head -45 $O/kstats.txt
