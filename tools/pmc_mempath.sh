#!/bin/bash
# tools/pmc_mempath.sh M N K mode [outdir]   (run on the GPU box)
# Vector-memory path of ONE GEMM shape (tools/one_gemm.py) in rocprofv3 counters-only passes: L2 read latency seen by the
# L1 (TCP), L1 stalls, TLB, TA back-pressure, SQ fifo-full, L2 tag stalls and fabric read level.
M=$1; N=$2; K=$3; MODE=${4:-nt}; OUT=${5:-gpurun_out/mem_${M}_${N}_${K}_${MODE}}
export TMPDIR=/tmp
mkdir -p $OUT
run() { local name=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python tools/one_gemm.py $M $N $K $MODE > $OUT/$name.log 2>&1 || tail -3 $OUT/$name.log; }
if [ -z "$SKIP_TCP" ]; then
run tcp1 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tcp2 TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
run tcp3 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
run tcp4 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
# (TA_* counters abort rocprofv3 on this image: not collected)
fi
run sqa SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run sqb SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
run sqc SQ_INST_LEVEL_VMEM
run tcc1 TCC_TAG_STALL_sum TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum TCC_BUSY_sum TCC_CYCLE_sum
run tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum TCC_MISS_sum
python - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(gemm_bf16\w*)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1)
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
        tot[k]["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); cnt[k]["_ns"] += 1
for k in tot:
    print("==", k, " avg launch %.1f us" % (tot[k]["_ns"] / cnt[k]["_ns"] / 1e3))
    t = {c: tot[k][c] / cnt[k][c] for c in tot[k]}
    for c in sorted(t):
        if c != "_ns": print("  %-40s per launch %16.0f" % (c, t[c]))
    def ratio(a, b, label):
        if a in t and b in t and t[b] > 0: print("  -> %-50s %.1f" % (label, t[a] / t[b]))
    ratio("TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum", "avg TCP->TCC read latency (cycles)")
    ratio("TCP_TCP_LATENCY_sum", "TCP_TA_TCP_STATE_READ_sum", "avg TCP wave latency (cycles)")
    ratio("TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RDREQ_sum", "avg L2->fabric read latency (cycles)")
    ratio("SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM_RD", "avg VMEM instruction latency (SQ level/insts, cycles?)")
PY
