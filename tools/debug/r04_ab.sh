mkdir -p gpurun_out/r04i; export TMPDIR=/tmp; O=gpurun_out/r04i
run() { env $1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['value'])" >> $O/ab.txt; }
for rep in 1 2; do
  for e in "ALL=on" "SEGCLIP_FUSED_HEAD=0" "SEGCLIP_GL64=0" "SEGCLIP_ATTN_FWD_LEAN=0" "SEGCLIP_GEMM_PQ_RES32=0" "SEGCLIP_GEMM_PQ_WGRAD=0"; do run "$e"; done
done
run "ALL=on"
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/trc -o trc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 > /tmp/trc.log 2>&1; cd $GRAFT_REPO_ROOT
ls -R /tmp/trc | head -20 > $O/trc_ls.txt
DB=$(ls /tmp/trc/*.db /tmp/trc/*/*.db 2>/dev/null | head -1)
python tools/small_kernel_sections.py $DB 45 30 > $O/sections.txt 2>&1
python - "$DB" <<'PY' > gpurun_out/r04i/kstats.txt
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels"))
agg = collections.defaultdict(lambda: [0, 0])
for n, s, e in rows:
    a = agg[n]; a[0] += 1; a[1] += e - s
for n, (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:50]:
    print(f"{n[:110]:110s} {cnt:6d} {t/1e6/8:9.3f} ms/step avg {t/cnt/1e3:8.1f} us")
PY
cat $O/ab.txt; head -12 $O/sections.txt; head -50 $O/kstats.txt
