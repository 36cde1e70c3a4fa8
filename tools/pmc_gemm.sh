#!/bin/bash
# tools/pmc_gemm.sh M N K mode [outdir]   (run on the GPU box)
# rocprofv3 PMC passes (separate runs; counters only, no tracing) on ONE GEMM shape of tools/one_gemm.py and a
# per-kernel summary: MFMA-pipe busy, LDS bank conflicts, wait breakdown, L2 hit rate, HBM bytes.
M=$1; N=$2; K=$3; MODE=${4:-nt}; OUT=${5:-gpurun_out/pmc_${M}_${N}_${K}_${MODE}}
export TMPDIR=/tmp
mkdir -p $OUT
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python tools/one_gemm.py $M $N $K $MODE > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        import re
        m = re.search(r"(gemm_bf16\w*|splitk\w*|reduce_multi\w*)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1)
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "TCC_HIT_sum", "FETCH_SIZE"):
            tot[k]["_ns_" + r["Counter_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); cnt[k]["_ns_" + r["Counter_Name"]] += 1
for k in tot:
    print("==", k)
    for c in sorted(tot[k]):
        print("  %-34s %16.0f  (per launch %14.0f, %d launches)" % (c, tot[k][c], tot[k][c] / cnt[k][c], cnt[k][c]))
    t = tot[k]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in t and "SQ_BUSY_CYCLES" in t:
        print("  -> MFMA pipe busy (MFMA_BUSY / (4 * SQ_BUSY_CYCLES per-SE sum)) : see raw; LDS conflict share %.3f" % (t["SQ_LDS_BANK_CONFLICT"] / max(t["SQ_LDS_IDX_ACTIVE"], 1)))
    if "TCC_HIT_sum" in t:
        print("  -> L2 hit rate %.3f" % (t["TCC_HIT_sum"] / max(t["TCC_HIT_sum"] + t["TCC_MISS_sum"], 1)))
PY
