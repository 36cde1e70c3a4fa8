"""tools/profiles_from_round.py <tag>: turn gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) into the
tracked files profiles/<tag>_*: the bench line and the kernel table of the SAME run (bench.py's own rocprofv3 child), the
other bench configurations, the PMC passes of single GEMM shapes with derived figures (MFMA-pipe busy, L2 hit rate, HBM
bytes), the attention PMC passes and the micro-benchmark tables."""
import json, os, re, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "gpurun_out", tag) + "/"
P = os.path.join(ROOT, "profiles", tag) + "_"


def last_json(path):
    for l in reversed(open(path).read().strip().splitlines()):
        l = l.strip()
        if l.startswith("{"):
            return json.loads(l)


line = last_json(R + "bench_line.json")
json.dump(line, open(P + "bench_line.json", "w"), indent=1)
hdr = ("# Kernel table of the rocprofv3 child that bench.py's roofline block is computed from (tools/profile_round.sh %s:\n"
       "#   SEGCLIP_BENCH_PROFILE_DIR=... python bench.py --steps 20 --warmup 5  ->  profiles/%s_bench_line.json, same run).\n"
       "# 6 model passes of the child are in the trace; ms/step = total_ms / 6.  at::cuda::spin_kernel = the stream-overlap probe\n"
       "# of segclip_amd/streams.py (once per process, not step work).  Step time of the (unprofiled) parent: %.3f ms.\n"
       "# gemm_bf16_pq_kernel<A_KS,B_KS,MODE>: <0,0,*> forward (0 plain, 2 QuickGELU + one-byte derivative, 5 + fp32 residual),\n"
       "#   <0,1,*> data gradient (0 plain, 3 x saved derivative), <1,1,4> weight gradient (split-K slabs);\n"
       "# gemm_bf16_pq_group_kernel: the weight gradients of several consecutive blocks as one launch (DESIGN 4.5).\n") % (tag, tag, line["ms_per_step"])
open(P + "bench_kernel_stats.txt", "w").write(hdr + open(R + "kernel_stats.txt").read())
cfg = {}
for n, desc in (("gb2048", "--global-batch 2048 (SURVEY 8d strong-scaling base on one GPU)"), ("full_loss", "--full-loss (BASELINE configs[3])"),
                ("resid_bf16", "--resid bf16: bf16 residual stream between the blocks of a tower (config.bf16_resid, opt-in)"),
                ("dist", "--force-dist: N>1 code path (RCCL group, GradSync fp32 wire) on one rank"), ("dist_bf16wire", "--force-dist --wire bf16"),
                ("vitl14", "--spec vitl14_336 --batch 128 --attn-fp8 off (BASELINE configs[4], bf16 attention)"),
                ("vitl14_fp8", "--spec vitl14_336 --batch 128 --attn-fp8 on"), ("plain_ref", "default configuration right before the `dist` / `dist_bf16wire` lines (same box)"),
                ("plain_ref2", "the same, after the dist lines"),
                ("b64", "--batch 64"), ("b96", "--batch 96 (the reference recipe's per-GPU batch; config.pad_rows pads the text tower's 7392 rows to 7424; r05: b96 / b96_pad_off / b192 were added after the round's pass, on another box)"),
                ("b96_pad_off", "--batch 96 SEGCLIP_PAD_ROWS=0 (the text tower on the ragged-edge kernels), same box"),
                ("b128", "--batch 128"), ("b192", "--batch 192"), ("b512", "--batch 512"),
                ("text_trim", "--text-trim: config.text_trim in the timed region (opt-in; causal text tower up to the batch's last EOT)"),
                ("attn_fwd_old_a", "SEGCLIP_ATTN_FWD_PF=0 (one workgroup per item attention forward), same box, --steps 30"),
                ("attn_fwd_pf_a", "default (persistent attention forward), same box, --steps 30"),
                ("attn_fwd_old_b", "SEGCLIP_ATTN_FWD_PF=0, second pass"), ("attn_fwd_pf_b", "default, second pass"),
                ("half_off_a", "SEGCLIP_PQ_HALF=0 (full 256 x 256 tiles only), same box, --steps 30"),
                ("half_on_a", "default (half-tile tail of gemm_bf16_pq.hip), same box, --steps 30"),
                ("half_off_b", "SEGCLIP_PQ_HALF=0, second pass"), ("half_on_b", "default, second pass"),
                ("b128_half_off_a", "--batch 128 SEGCLIP_PQ_HALF=0 (M = 256 q + 128 leaves gemm_bf16_pq.hip), same box, --steps 30"),
                ("b128_half_on_a", "--batch 128 default (last 128 rows as a row of half-tiles), same box, --steps 30"),
                ("b128_half_off_b", "--batch 128 SEGCLIP_PQ_HALF=0, second pass"), ("b128_half_on_b", "--batch 128 default, second pass"),
                ("full_fold_off_a", "--full-loss SEGCLIP_FOLD_GRADS=0 (the autograd engine adds the two passes' parameter gradients), same box"),
                ("full_fold_on_a", "--full-loss default (config.fold_param_grads), same box"),
                ("full_fold_off_b", "--full-loss SEGCLIP_FOLD_GRADS=0, second pass"), ("full_fold_on_b", "--full-loss default, second pass"),
                ("wgrad_single_a", "SEGCLIP_WGRAD_GROUP=1 (one launch per weight gradient), same box, --steps 30"),
                ("wgrad_grouped_a", "default (grouped weight gradients), same box, --steps 30"),
                ("wgrad_single_b", "SEGCLIP_WGRAD_GROUP=1, second pass"), ("wgrad_grouped_b", "default, second pass")):
    if not os.path.exists(R + f"bench_{n}.json"):
        continue
    d = last_json(R + f"bench_{n}.json")
    cfg[n] = {"command": "python bench.py --no-cpu-baseline " + desc, "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
              "config": d["config"], "roofline": d.get("roofline"), "clocks": d.get("clocks")}
json.dump(cfg, open(P + "bench_configs.json", "w"), indent=1)
txt = ("# rocprofv3 --pmc passes (counters only; one counter group per run) on single GEMM shapes of the shipped 8-phase kernel\n"
       "# (tools/pmc_gemm.sh via tools/profile_round.sh); values per launch = mean of 3 launches of tools/one_gemm.py.\n"
       "# FETCH_SIZE / WRITE_SIZE raw in KiB; HBM bytes apply the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE x2).\n")
shapes = {"nt": ("forward NT  C[50176,768] = A[50176,3072] W[768,3072]^T", 50176, 768, 3072),
          "dgrad": ("data grad   dX[50176,3072] = dY[50176,768] W[768,3072]", 50176, 3072, 768),
          "wgrad": ("weight grad dW[2304,768] = dY[50176,2304]^T X[50176,768] (split-K + combine)", 2304, 768, 50176)}
for mode, (desc, M, N, K) in shapes.items():
    if not os.path.exists(R + f"pmc_gemm_{mode}.txt"):
        continue
    body = open(R + f"pmc_gemm_{mode}.txt").read()
    txt += f"\n## {mode}: {desc}\n" + body
    g, cur = {}, None
    for l in body.splitlines():
        if l.startswith("=="):
            cur = l.split()[1]; g[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+\d+\s+\(per launch\s+(\d+)", l)
            if m and cur:
                g[cur][m.group(1)] = float(m.group(2))
    k = g.get("gemm_bf16_pq_kernel") or g.get("gemm_bf16_p8_kernel")
    if k:
        ns = k["_ns_SQ_WAVE_CYCLES"]; cyc = k["GRBM_GUI_ACTIVE"] / 8
        fl = 2.0 * M * N * K
        hbm = k["FETCH_SIZE"] * 1024 * 2 + k["WRITE_SIZE"] * 1024
        txt += (f"  => kernel {ns / 1e3:.1f} us ({fl / ns / 1e3:.0f} TF/s in the counter pass), MFMA pipe busy "
                f"{k['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f} (busy cycles / (GUI_ACTIVE/8 x 1024 SIMDs)), waves waiting "
                f"{k['SQ_WAIT_ANY'] / k['SQ_WAVE_CYCLES']:.3f} + issue-stalled {k['SQ_WAIT_INST_ANY'] / k['SQ_WAVE_CYCLES']:.3f} of their cycles, "
                f"LDS conflicts {k['SQ_LDS_BANK_CONFLICT'] / k['SQ_LDS_IDX_ACTIVE']:.3f} of LDS-active cycles, "
                f"L2 hit {k['TCC_HIT_sum'] / (k['TCC_HIT_sum'] + k['TCC_MISS_sum']):.3f}, HBM {hbm / 1e6:.0f} MB\n")
open(P + "gemm_pmc.txt", "w").write(txt)
open(P + "attn_pmc.txt", "w").write("# rocprofv3 --pmc passes on the attention kernels, B=256 T=196 H=12 hd=64 (tools/pmc_attn.sh); per launch.\n"
                                   "# attn_bwd_dqw_bf16_kernel = the streaming backward with a dQ wave (attention_dqw.inc, round 6)\n" + open(R + "pmc_attn.txt").read())
for a, h in (("gemm_shapes.txt", "# GEMM rates per shape (tools/bench_gemm.py, HIP events, 10 launches each) next to torch.matmul (hipBLASLt) on the same MI355X.\n"),
             ("gemm_epilogues.txt", "# Fused-epilogue variants of one residual block's GEMMs against the plain kernel (tools/bench_epi.py, M = 50176, D = 768)\n"),
             ("gemm_pq.txt", "# gemm_bf16_pq.hip against the 8-phase kernel gemm_bf16_p8.hip per shape and mode, interleaved rounds (tools/bench_pq.py), M = 50176 (vision) and 19712 (text)\n"),
             ("hbm_kernels.txt", "# HBM-bound kernels against 8 TB/s (tools/bench_hbm.py)\n"),
             ("attn.txt", "# attention kernels in isolation (tools/bench_attn.py): T=196 vision (single-pass backward), T=77 causal text\n"),
             ("center_stage.txt", "# Kernel time of the learnable-center stage alone (SemanticLearnerModule forward + backward, B = 256, bf16, t18 mode;\n# tools/debug/center_stage_profile.py, torch profiler, mean of 3 passes).  Round-3 build: 4.71 ms over 361 launches, round 5: 3.18 ms over 241 (docs/experiment_log.md 4.5)\n"),
             ("stream_gaps.txt", "# tools/stream_gaps.py on the kernel trace of the bench child (profiled run: the host is slower than in the timed run)\n"),
             ("stream_breakdown.txt", "# tools/stream_breakdown.py on the same trace: kernel time per stream and kernel (the window holds more than the 3 passes it divides by:\n# read the rows relative to each other - stream 0 is the vision tower + everything serial, i.e. the critical path)\n"),
             ("gemm_half_tile.txt", "# tools/bench_pq_half.py: gemm_bf16_pq.hip with full tiles only against the half-tile tail (SEGCLIP_PQ_HALF=2, per-call switch), interleaved rounds, M = 50432; outputs bit-identical\n")):
    if os.path.exists(R + a):
        open(P + a, "w").write(h + open(R + a).read())
if os.path.exists(R + "eager_ab.json"):
    e = last_json(R + "eager_ab.json")
    e["segclip_amd_same_run"] = {"pairs_per_s": line["value"], "ms_per_step": line["ms_per_step"]}
    e["speedup_vs_eager"] = round(line["value"] / e["pairs_per_s"], 2)
    json.dump(e, open(P + "eager_ab.json", "w"), indent=1)
import shutil
for a, b in (("bench_intervals.json", "bench_intervals.json"), ("bucket_timeline.txt", "bucket_timeline.txt")):
    if os.path.exists(R + a):
        shutil.copy(R + a, P + b)
if os.path.exists(R + "accuracy_b256.txt"):
    body = "".join(l for l in open(R + "accuracy_b256.txt") if not re.match(r"^(RCCL|HIP|ROCm|Hostname|Librccl)|amdgpu.ids", l))
    open(P + "accuracy_b256.txt", "w").write(body)
print("wrote", P + "*")
