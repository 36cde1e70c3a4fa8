#!/bin/bash
# tools/profile_round.sh <tag>   (run on the GPU box; results under gpurun_out/<tag>/)
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> kernel_stats.txt
# 2. PMC passes (separate runs, counters only) of ONE bench step     -> gemm traffic json (FETCH_SIZE / WRITE_SIZE)
# 3. PMC passes of single GEMM shapes (MFMA busy, LDS conflicts, L2 hit rate) for the three operand layouts
TAG=${1:-r02}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/bench_prof.err
python tools/prof_summary.py $(ls $OUT/prof/*.db | head -1) 60 > $OUT/kernel_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc_$c.log 2>&1
done
python tools/pmc_traffic.py $(ls $OUT/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls $OUT/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1) $OUT/gemm_traffic.json > $OUT/gemm_traffic.log 2>&1
for spec in "50176 768 3072 nt" "50176 3072 768 dgrad" "50176 2304 768 wgrad"; do
  set -- $spec
  timeout 600 bash tools/pmc_gemm.sh $1 $2 $3 $4 $OUT/pmc_gemm_$4 > $OUT/pmc_gemm_$4.txt 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
for mode in ("nt", "dgrad", "wgrad"):
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(f"{out}/pmc_gemm_{mode}/*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(gemm_bf16\w*|splitk\w*)", r["Kernel_Name"])
            if not m: continue
            k = m.group(1)
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "TCC_HIT_sum", "FETCH_SIZE"):
                tot[k]["_ns_" + r["Counter_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); cnt[k]["_ns_" + r["Counter_Name"]] += 1
    with open(f"{out}/pmc_gemm_{mode}_summary.txt", "w") as fo:
        for k in tot:
            fo.write(f"== {mode}: {k}\n")
            for c in sorted(tot[k]): fo.write("  %-34s per launch %16.0f (%d launches)\n" % (c, tot[k][c] / cnt[k][c], cnt[k][c]))
PY
cat $OUT/kernel_stats.txt | head -30 | cut -c1-140; cat $OUT/gemm_traffic.log | tail -2; cat $OUT/pmc_gemm_*_summary.txt | head -80
