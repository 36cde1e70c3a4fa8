#!/bin/bash
# tools/pmc_attn.sh [outdir] [B T H causal]   (run on the GPU box): rocprofv3 PMC passes (counters only) on the attention
# kernels and a per-kernel summary.
OUT=${1:-gpurun_out/pmc_attn}; shift
export TMPDIR=/tmp
mkdir -p $OUT
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python tools/one_attn.py $ARGS > $OUT/$name.log 2>&1; }
ARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run sq3 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
python - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(attn_\w+)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1)
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES",):
            tot[k]["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); cnt[k]["_ns"] += 1
with open(out + "/summary.txt", "w") as fo:
    for k in tot:
        fo.write(f"== {k}\n")
        for c in sorted(tot[k]): fo.write("  %-34s per launch %16.0f (%d launches)\n" % (c, tot[k][c] / cnt[k][c], cnt[k][c]))
print(open(out + "/summary.txt").read())
PY
