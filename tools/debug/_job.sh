mkdir -p gpurun_out/r04w
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>&1 | grep '"metric"' > gpurun_out/r04w/bench_union.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04w/bench_union.json"))
r = d["roofline"]
print(d["value"], d["ms_per_step"], "frac", r["frac"], "achieved", r["achieved"], "time", r["time_per_step_ms"], "union", r["union_ms_per_step"], r["achieved_union"], r["frac_union"], "step_frac", r["step_frac"])
for k, v in r["classes"].items():
    print(k, v)
PY
