"""tools/profiles_from_round.py <tag> [step_ms_note]: turn gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU
box) into the tracked files profiles/<tag>_bench_kernel_stats.txt, profiles/<tag>_gemm_traffic.json and
profiles/<tag>_gemm_pmc.txt, with the command lines and the derived figures (MFMA-pipe busy, L2 hit rate, HBM bytes)."""
import json, os, re, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "gpurun_out", tag) + "/"
P = os.path.join(ROOT, "profiles", tag)
bench = json.loads(open(R + "bench_prof.json").read().strip().splitlines()[-1])

hdr = f"""# rocprofv3 --kernel-trace --stats of the default bench command, as run by tools/profile_round.sh {tag}:
#   cd /tmp && export TMPDIR=/tmp
#   rocprofv3 --kernel-trace --stats -d gpurun_out/{tag}/prof -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline
# ViT-B/16 224^2 + 77-token text, B=256, bf16, fwd+bwd.  11 model passes are inside the trace (2 warm-up + 5 timed +
# 4 of the roofline section), so per-step totals = total_ms / 11.  Summary made by tools/prof_summary.py from the
# rocprofv3 database.  Step time printed by this (profiled) run: {bench['ms_per_step']} ms.
# gemm_bf16_p8_kernel<A_KS,B_KS,0>: <0,0> = forward NT, <0,1> = data gradient, <1,1> = weight gradient (split-K).
"""
open(P + "_bench_kernel_stats.txt", "w").write(hdr + open(R + "kernel_stats.txt").read())

t = json.load(open(R + "gemm_traffic.json"))
t["source"] = (f"tools/profile_round.sh {tag}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, two separate runs of "
               "`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline`")
json.dump(t, open(P + "_gemm_traffic.json", "w"), indent=1)


def parse(f):
    out, k = {}, None
    for l in open(f):
        if l.startswith("=="):
            k = l.split(":")[1].strip(); out[k] = {}
        else:
            m = re.match(r"\s+(\S+)\s+per launch\s+(\d+)", l)
            if m:
                out[k][m.group(1)] = float(m.group(2))
    return out


txt = """# rocprofv3 --pmc passes (counters only, no tracing; one counter group per run) on single GEMM shapes of the SHIPPED
# kernel gemm_bf16_p8_kernel (256x256x64 tile, LDS-DMA, 4-phase schedule): tools/pmc_gemm.sh via tools/profile_round.sh.
# Values are per launch (mean of the 3 launches of tools/one_gemm.py).  FETCH_SIZE / WRITE_SIZE are the raw counters in
# KiB; "HBM bytes" below applies the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE x2).  _ns_* = kernel duration
# (ns) in that counter's pass.
"""
shapes = {"nt": ("forward NT  C[50176,768] = A[50176,3072] W[768,3072]^T", 50176, 768, 3072),
          "dgrad": ("data grad   dX[50176,3072] = dY[50176,768] W[768,3072]", 50176, 3072, 768),
          "wgrad": ("weight grad dW[2304,768] = dY[50176,2304]^T X[50176,768] (split-K + splitk_reduce_vec_kernel)", 2304, 768, 50176)}
for mode in ("nt", "dgrad", "wgrad"):
    f = R + f"pmc_gemm_{mode}_summary.txt"
    if not os.path.exists(f):
        continue
    d = parse(f)
    desc, M, N, K = shapes[mode]
    txt += f"\n## {mode}: {desc}\n" + open(f).read()
    g = d["gemm_bf16_p8_kernel"]
    ns = g["_ns_SQ_WAVE_CYCLES"]; cyc = g["GRBM_GUI_ACTIVE"] / 8
    fl = 2.0 * M * N * K
    alg = (M * K + N * K + M * N) * 2 if mode != "wgrad" else (K * M + K * N) * 2 + g["WRITE_SIZE"] * 1024
    hbm = g["FETCH_SIZE"] * 1024 * 2 + g["WRITE_SIZE"] * 1024
    txt += (f"# derived ({mode}): duration {ns / 1e3:.1f} us -> {fl / ns / 1e3:.0f} TFLOP/s; shader clock = GRBM_GUI_ACTIVE/8/duration = {cyc / ns:.2f} GHz;\n"
            f"#   MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x {cyc / 1e3:.0f}k cycles) = {g['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc) * 100:.1f} %;\n"
            f"#   wave wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES = {g['SQ_WAIT_ANY'] / g['SQ_WAVE_CYCLES'] * 100:.0f} %; LDS bank-conflict cycles / LDS active = {g['SQ_LDS_BANK_CONFLICT'] / g['SQ_LDS_IDX_ACTIVE'] * 100:.1f} %;\n"
            f"#   L2 hit = TCC_HIT/(HIT+MISS) = {g['TCC_HIT_sum'] / (g['TCC_HIT_sum'] + g['TCC_MISS_sum']) * 100:.0f} %;\n"
            f"#   HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE = {hbm / 1e6:.0f} MB ({hbm / ns / 1e3:.2f} TB/s) vs operand bytes "
            f"{'A+B+C' if mode != 'wgrad' else 'A+B + the fp32 split-K slabs written'} = {alg / 1e6:.0f} MB\n")
open(P + "_gemm_pmc.txt", "w").write(txt)
print("wrote", P + "_{bench_kernel_stats.txt,gemm_traffic.json,gemm_pmc.txt}")
