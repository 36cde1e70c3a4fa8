"""bench.py - image-text pairs/s, forward+backward, of the SegCLIP contrastive hot path on MI355X.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N>1 launched through
torch.distributed.run, one rank per GPU over RCCL (a bare `python bench.py --gpus N` starts the N ranks
itself).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1] at N=1, configs[2] at N=8): ViT-B/16 224^2 + 77-token text,
contrastive loss only, per-GPU batch 256 (weak scaling: global batch 256*N, 2048 at N=8), bf16 MFMA
kernels, fp32 master weights, synthetic images/captions, closed-form random weights.  A step is one
forward + backward (loss -> every parameter gradient, incl. the embedding all-gather and, for N>1, the
bucketed gradient all-reduce of segclip_amd.dist.GradSync); the optimizer step is excluded, as in
BASELINE.json's metric definition.  --global-batch 2048 switches to SURVEY 8(d)'s strong-scaling form
(per-GPU batch 2048/N, incl. the single-GPU B=2048 point).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # see segclip_amd/__init__.py: keeps the side streams on their own queues

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import segclip_amd  # noqa: E402
from segclip_amd import ops, synth  # noqa: E402

# algorithmic GFLOP per image-text pair, forward+backward, contractions only
GF_PER_PAIR = {("vitb16", False): 109.675, ("vitb16", True): 144.074, ("vitl14_336", False): 535.273}
MODEL_NAME = {"vitb16": "ViT-B/16 224^2", "vitl14_336": "ViT-L/14 336^2 (12 blocks of width 1024, 576 patches)", "tiny": "tiny"}
PEAK_BF16_TF = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling: global = batch * N)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: fixed global batch (SURVEY 8d: 2048), per-GPU batch = global / N")
    ap.add_argument("--spec", default="vitb16")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--full-loss", action="store_true", help="configs[3]: + superpixel-KL + MAE")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-worker", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity-leg", action="store_true",
                    help="skip the legs behind the timed region: exact-f32 parity mode timed for 3 steps + bf16-vs-f32 agreement on "
                         "injected noise (`parity_mode`, `bf16_vs_f32`), and the single-GPU global-batch-2048 point")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child runs (HBM bytes of the GEMM)")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the N>1 code path (RCCL group, GradSync, barriers) even with one rank: single-GPU check of it")
    ap.add_argument("--grad-sync", default="segclip", choices=["segclip", "ddp"],
                    help="N>1 gradient exchange: segclip_amd.dist.GradSync (default) or torch DDP (comparison only)")
    ap.add_argument("--attn-fp8", default="auto", choices=["auto", "on", "off"],
                    help="e4m3 MFMA for QK^T / PV in the self-attention forward.  auto = off: at head_dim 64 the non-scaled e4m3 "
                         "MFMA runs at the bf16 rate and the quantisation passes make the step slower (1281 vs 1308 pairs/s on "
                         "configs[4], profiles/r03_bench_configs.json); `on` keeps the configs[4] switch reachable")
    ap.add_argument("--resid", default="auto", choices=["auto", "bf16", "fp32"],
                    help="residual stream between the blocks of a tower in bf16 mode (config.bf16_resid); auto = the package default")
    ap.add_argument("--f32-split", action="store_true", help="with --dtype f32: config.f32_split (Linear layers as bf16 x 3 products)")
    ap.add_argument("--text-trim", action="store_true",
                    help="config.text_trim for the TIMED region (opt-in, off by default: the headline times the reference's full "
                         "77-token context): the causal text tower on the positions up to the batch's last EOT only - identical loss "
                         "and gradients (tests/test_model_gpu.py); the default run reports it as the secondary field `text_trim`")
    ap.add_argument("--text-after-blocks", type=int, default=-1, help="config.text_after_blocks override (launch order of the towers)")
    ap.add_argument("--wire", default="auto", choices=["auto", "bf16", "fp32"], help="gradient all-reduce wire format (auto = fp32, what the reference DDP exchanges; bf16 is an opt-in)")
    ap.add_argument("--rccl-channels", type=int, default=0,
                    help="N>1: cap (and floor) the RCCL channel count = the CUs the collective kernels take from the 256-CU GEMMs "
                         "(sets NCCL_MIN_NCHANNELS = NCCL_MAX_NCHANNELS before the communicator is created; 0 = RCCL's own choice)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend of the N>1 path.  gloo + --share-gpu exist for the single-GPU test of this script's "
                         "own N>1 control flow (RCCL refuses two ranks on one device); numbers from it are not RCCL numbers")
    ap.add_argument("--share-gpu", action="store_true", help="map rank r to device r %% visible devices (tests only)")
    return ap.parse_args()


def usable_cores():
    """CPU cores this process can really use: affinity mask and cgroup CPU quota, not os.cpu_count() (a 256-core host
    with a container quota ran the 256-thread oracle step 400x slower than the 32-thread one: oversubscribed OpenMP)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(float(quota) / period + 0.5)))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline_worker(threads):
    """One thread count of the CPU baseline, in its own process (prints one JSON object)."""
    from oracle import segclip_oracle as so
    from tests.helpers import FULL_FLAGS, model_param_shapes, oracle_params
    spec = synth.SPECS["vitb16"]
    torch.set_num_threads(threads)
    B, out = 4, {}
    for tag, flags in (("contrastive", {}), ("full_loss", FULL_FLAGS)):
        P = oracle_params(spec, model_param_shapes(spec, flags))
        batch = synth.synthetic_batch(spec, B, seed=2, with_seg=bool(flags))
        noise = synth.synthetic_noise(spec, B, seed=2)
        best, loss = None, None
        for it in range(4):  # 1 warm-up + 3 timed
            for p in P.values():
                p.grad = None
            t0 = time.perf_counter()
            loss, _ = so.segclip_forward(batch, P, spec, noise, flags)
            loss.backward()
            dt = time.perf_counter() - t0
            if it and (best is None or dt < best):
                best = dt
            if it and dt > 5.0:
                break          # a slow host: one timed step instead of three keeps the run bounded
        out[tag] = [B / best, best, float(loss.detach())]
        print(json.dumps({"partial": out}), flush=True)
    print(json.dumps({"done": out}), flush=True)


def cpu_baseline():
    """Reference CPU path (the oracle = validated CPU restatement of the reference; the reference's Python cannot travel
    to the GPU box) on this host's cores, SURVEY.md 8(d): BASELINE.json configs[0] - ViT-B/16 + 77-token text, batch 4,
    1 step fwd+bwd after 1 warm-up, best of 3 - contrastive-only (the metric's loss) and the full SegCLIP loss, with
    all usable cores (affinity / cgroup quota) and, because torch-CPU eager stops scaling long before 100+ threads, with
    32 threads as well.  Every thread count runs in its own process under a 90-second limit, so the bench stays
    bounded on any host; `value` is the best contrastive-only rate, `cores` the thread count that produced it."""
    import subprocess
    host, usable = os.cpu_count() or 1, usable_cores()
    try:
        model_name = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        model_name = "unknown"
    results, notes = {}, []
    for threads in sorted({usable, min(usable, 32)}):
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads))
        got = None
        try:
            pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(threads)], env=env,
                                stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=90, text=True)
            lines = pr.stdout
        except subprocess.TimeoutExpired as e:
            lines = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
            notes.append(f"{threads} threads: stopped at the 90-s limit")
        for l in lines.splitlines():
            try:
                o = json.loads(l)
                got = o.get("done") or o.get("partial") or got
            except Exception:
                pass
        for tag, v in (got or {}).items():
            results[(tag, threads)] = tuple(v)
    con = {t: v for (tag, t), v in results.items() if tag == "contrastive"}
    full = {t: v for (tag, t), v in results.items() if tag == "full_loss"}
    if not con:
        return {"value": None, "unit": "pairs/s", "cores": usable, "host_cores": host, "cpu": model_name, "kind": "port",
                "sample": "no CPU baseline finished within the limit: " + "; ".join(notes)}
    bt = max(con, key=lambda t: con[t][0])
    out = {"value": round(con[bt][0], 3), "unit": "pairs/s", "cores": bt, "usable_cores": usable, "host_cores": host,
           "cpu": model_name, "kind": "port",
           "sample": f"oracle (CPU restatement of the reference, torch {torch.__version__} fp32), BASELINE configs[0]: "
                     f"ViT-B/16 B=4 fwd+bwd, 1 warm-up + best of 3, one process per thread count; contrastive-only: "
                     + ", ".join(f"{t} threads {v[1]:.2f} s/step" for t, v in sorted(con.items()))
                     + ("; full loss: " + ", ".join(f"{t} threads {v[1]:.2f} s/step" for t, v in sorted(full.items())) if full else "")
                     + ("; " + "; ".join(notes) if notes else ""),
           "loss": round(con[bt][2], 6)}
    if full:
        ft = max(full, key=lambda t: full[t][0])
        out["full_loss_value"] = round(full[ft][0], 3)
        out["full_loss"] = round(full[ft][2], 6)
    return out


def respawn(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU over RCCL)
    through torch.distributed.run and relay rank 0's JSON line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def child_argv(a, steps, warmup):
    """bench.py command line of the same per-GPU workload on ONE GPU, without the roofline / CPU-baseline legs."""
    argv = [os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup), "--batch", str(a.batch),
            "--spec", a.spec, "--dtype", a.dtype, "--attn-fp8", a.attn_fp8, "--resid", a.resid, "--no-roofline", "--no-cpu-baseline",
            "--no-parity-leg"]
    if a.full_loss:
        argv.append("--full-loss")
    if a.text_trim:
        argv.append("--text-trim")
    return argv


PEAK_HBM_GBS = 8000.0


def count_step_work(step):
    """{kernel class: [launches, flops, bytes]} of one step, counted by the op layer.  Runs one step: at N>1 EVERY rank
    must call it (the step's collectives)."""
    ops._OpCount.start()
    step()
    torch.cuda.synchronize()
    work = {}
    for kind, fl, nb in ops._OpCount.stop():
        w = work.setdefault(kind, [0, 0.0, 0.0])
        w[0] += 1; w[1] += fl; w[2] += nb
    return work


def roofline_block(a, step, pairs_per_gpu, world, work, collective_free=True):
    """Roofline of the kernel classes that carry the step, from KERNEL time: this command's per-GPU workload is run again
    for a few steps as a child under `rocprofv3 --kernel-trace --stats` (and under two counters-only --pmc passes for the
    HBM bytes of the dominant kernel); the algorithmic flops / bytes of each class (`work`) are counted by the op layer
    during one pass of the parent.  The towers run on two streams, so class times may overlap each other (their sum
    exceeds the step time); a class time is the sum of its kernels' own durations, exactly what `rocprofv3 --stats` prints.
    collective_free=False (N>1: only rank 0 is here): nothing in this function may run a model step."""
    from tools import rocprof_roofline as rr
    step_frac = (round(pairs_per_gpu * GF_PER_PAIR[(a.spec, a.full_loss)] / 1e3 / PEAK_BF16_TF, 4)
                 if (a.spec, a.full_loss) in GF_PER_PAIR else None)
    ksteps, kwarm = 4, 2
    keep = os.environ.get("SEGCLIP_BENCH_PROFILE_DIR")   # tools/profile_round.sh keeps the database and the summary
    try:
        m = rr.measure(child_argv(a, ksteps, kwarm), ksteps + kwarm,
                       pmc_argv=None if a.no_traffic else child_argv(a, 1, 1), keep_dir=keep, timeout=600)
    except Exception as e:
        return roofline_fallback(a, step if collective_free else None, step_frac, work,
                                 f"rocprofv3 child not usable here ({type(e).__name__}: {str(e)[:200]})")
    cl = m["classes"]
    if keep:
        with open(os.path.join(keep, "kernel_stats.txt"), "w") as f:
            f.write(rr.format_table(m["table"], ksteps + kwarm,
                                    header=f"# rocprofv3 --kernel-trace --stats -- python {' '.join(child_argv(a, ksteps, kwarm)[1:])}\n"
                                           f"# (the child run bench.py's roofline block is computed from)"))

    def mfma(kind, cls):
        if kind not in work or cls not in cl or cl[cls]["time_per_step_ms"] <= 0:
            return None
        n, fl, nb = work[kind]
        t = cl[cls]["time_per_step_ms"] * 1e-3
        d = {"time_per_step_ms": cl[cls]["time_per_step_ms"], "time_base": "sum_of_durations (two concurrent streams: may exceed ms_per_step)",
             "launches_per_step": cl[cls]["launches_per_step"],
             "avg_launch_us": cl[cls]["avg_launch_us"], "bound": "mfma", "achieved_per_launch": round(fl / t / 1e12, 1), "unit": "TFLOP/s",
             "frac_per_launch": round(fl / t / 1e12 / PEAK_BF16_TF, 4), "algorithmic_gbytes_per_s": round(nb / t / 1e9, 1)}
        u = cl[cls].get("union_ms_per_step")
        if u:   # time during which at least one kernel of the class runs (two concurrent streams share the chip)
            d.update(union_ms_per_step=u, achieved=round(fl / (u * 1e-3) / 1e12, 1),
                     frac=round(fl / (u * 1e-3) / 1e12 / PEAK_BF16_TF, 4))
        else:
            d.update(achieved=d["achieved_per_launch"], frac=d["frac_per_launch"])
        return d

    def hbm(kind, cls):
        if kind not in work or cls not in cl or cl[cls]["time_per_step_ms"] <= 0:
            return None
        n, fl, nb = work[kind]
        t = cl[cls]["time_per_step_ms"] * 1e-3
        return {"time_per_step_ms": cl[cls]["time_per_step_ms"], "launches_per_step": cl[cls]["launches_per_step"],
                "avg_launch_us": cl[cls]["avg_launch_us"], "bound": "hbm", "achieved": round(nb / t / 1e9, 1), "unit": "GB/s",
                "frac": round(nb / t / 1e9 / PEAK_HBM_GBS, 4)}

    g = mfma("gemm_bf16", "gemm_bf16")
    if g is None:
        return roofline_fallback(a, step if collective_free else None, step_frac, work, "no bf16 GEMM dispatch in the rocprofv3 trace")
    n, fl, nb = work["gemm_bf16"]
    classes = {"gemm_bf16": g, "attention_fwd": mfma("attn_fwd", "attn_fwd"), "attention_bwd": mfma("attn_bwd", "attn_bwd"),
               "layernorm_bwd": hbm("ln_bwd", "ln_bwd"), "layernorm_fwd": hbm("ln_fwd", "ln_fwd")}
    for extra in ("splitk_reduce", "row_reductions", "gemm_f32", "other"):   # ("startup_probe" is not step work)
        if extra in cl:
            classes[extra] = {"time_per_step_ms": cl[extra]["time_per_step_ms"], "launches_per_step": cl[extra]["launches_per_step"]}
    tr = m["traffic"]
    return {"bound": "mfma", "kernel": "gemm_bf16_pq_kernel / gemm_bf16_pq_group_kernel / gemm_bf16_p8_kernel / gemm_bf16_dma_kernel (+gemm_bf16_kernel fallback), all launches of a step",
            "achieved": g["achieved"], "peak": PEAK_BF16_TF, "unit": "TFLOP/s", "frac": g["frac"],
            "time_base": ("union: the time of a step during which at least one kernel of the class is running (both streams), "
                          "the figure that stays below ms_per_step" if g.get("union_ms_per_step") else "sum_of_durations"),
            "union_ms_per_step": g.get("union_ms_per_step"),
            "avg_launch_us": g["avg_launch_us"], "launches_per_step": g["launches_per_step"],
            "sum_of_durations_ms_per_step": g["time_per_step_ms"],
            "achieved_per_launch": g["achieved_per_launch"], "frac_per_launch": g["frac_per_launch"],
            "gpu_busy_ms_per_step": m.get("union_all_ms_per_step"),
            "algorithmic_flops_per_launch": round(fl / n), "algorithmic_bytes_per_launch": round(nb / n),
            "traffic": tr["hbm_bytes_per_launch"] if tr else None, "traffic_detail": tr,
            "source": f"kernel durations of a rocprofv3 --kernel-trace --stats child run of this workload on one GPU ({ksteps + kwarm} model "
                      "passes, text tower concurrent on its second stream as in the timed region); flops / bytes counted by the op layer",
            "two_stream_note": "achieved / frac = the class's flops over union_ms_per_step (merged kernel intervals of the class over both "
                               "streams; <= ms_per_step by construction).  achieved_per_launch / frac_per_launch = flops per launch over the "
                               "average launch duration = what `rocprofv3 --stats` prints (sum_of_durations_ms_per_step: two GEMMs sharing "
                               "the chip each take longer, so this sum may exceed ms_per_step and counts the machine twice)",
            "all_kernels_ms_per_step": round(sum(v["time_per_step_ms"] for k, v in cl.items() if k != "startup_probe"), 3),
            "classes": classes, "step_frac": step_frac, "notes": m["notes"]}


def roofline_fallback(a, step, step_frac, work, why):
    """HIP-event timing of every bf16 GEMM launch with the towers SERIALISED (one stream: an event interval is then the
    kernel's own time plus its dispatch gap) - used only where rocprofv3 cannot be started.  step=None (N>1, where a step
    on rank 0 alone would hang in its collectives): no kernel-level number, the step-level fraction only."""
    if step is None:
        n, fl, nb = work.get("gemm_bf16", [0, 0.0, 0.0])
        return {"bound": "mfma", "kernel": "bf16 GEMM kernels, all launches of a step", "achieved": None, "peak": PEAK_BF16_TF,
                "unit": "TFLOP/s", "frac": None, "traffic": None, "launches_per_step": n,
                "algorithmic_flops_per_launch": round(fl / n) if n else None,
                "algorithmic_bytes_per_launch": round(nb / n) if n else None,
                "source": "FALLBACK at N>1: no kernel-time measurement (" + why + "); step_frac is the whole-step fraction",
                "step_frac": step_frac}
    segclip_amd.config.overlap_towers = False
    step()
    ops._GemmProfile.start()
    step()
    rec = [r for r in ops._GemmProfile.stop() if r[2]]
    segclip_amd.config.overlap_towers = True
    tsum, fsum = sum(r[0] for r in rec), sum(r[1] for r in rec)
    return {"bound": "mfma", "kernel": "bf16 GEMM kernels, all launches of a step", "achieved": round(fsum / tsum / 1e12, 1),
            "peak": PEAK_BF16_TF, "unit": "TFLOP/s", "frac": round(fsum / tsum / 1e12 / PEAK_BF16_TF, 4), "traffic": None,
            "avg_launch_us": round(tsum / len(rec) * 1e6, 1), "launches_per_step": len(rec),
            "algorithmic_flops_per_launch": round(fsum / len(rec)),
            "algorithmic_bytes_per_launch": round(sum(r[3] for r in rec) / len(rec)),
            "source": "FALLBACK: HIP events around serialised launches (towers on one stream); " + why, "step_frac": step_frac}


def main():
    a = parse()
    if a.cpu_baseline_worker:
        cpu_baseline_worker(a.cpu_baseline_worker)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        n_dev = torch.cuda.device_count()
        if n_dev < a.gpus and not a.share_gpu:
            raise SystemExit(f"--gpus {a.gpus} but only {n_dev} GPU(s) visible")
        respawn(a)
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.global_batch:
        if a.global_batch % world:
            raise SystemExit(f"--global-batch {a.global_batch} is not divisible by {world} ranks")
        a.batch = a.global_batch // world
    if a.share_gpu:
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or a.force_dist
    if multi:
        import datetime
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if a.rccl_channels > 0:   # before the communicator exists: RCCL reads these when it builds its rings
            os.environ["NCCL_MIN_NCHANNELS"] = os.environ["NCCL_MAX_NCHANNELS"] = str(a.rccl_channels)
        # the other ranks wait at the final barrier while rank 0 runs its rocprofv3 children (minutes): a long timeout
        kw = dict(device_id=dev) if a.backend == "nccl" else {}
        dist.init_process_group(a.backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=60), **kw)
    spec = synth.SPECS[a.spec]
    flags = dict(use_seglabel=True, use_vision_mae_recon=True) if a.full_loss else {}
    segclip_amd.set_compute_dtype(torch.bfloat16 if a.dtype == "bf16" else torch.float32)
    if a.f32_split:
        segclip_amd.config.f32_split = True
    attn_fp8 = a.dtype == "bf16" and a.attn_fp8 == "on"
    segclip_amd.config.attn_fp8 = attn_fp8
    if a.resid != "auto":
        segclip_amd.config.bf16_resid = a.resid == "bf16"
    if os.environ.get("SEGCLIP_FUSED_HEAD", "1") == "0":     # A/B switch of the fused pooled-feature + contrastive head
        segclip_amd.config.fused_head = False
    if os.environ.get("SEGCLIP_REDUCE_SIDE", "0") == "1":    # A/B switch: trailing reductions on a side stream (measured slower)
        segclip_amd.config.reduce_side = True
    if a.text_after_blocks >= 0:
        segclip_amd.config.text_after_blocks = a.text_after_blocks
    if a.text_trim:
        segclip_amd.config.text_trim = True
    if os.environ.get("SEGCLIP_OVERLAP_TOWERS", "1") == "0":   # experiment: both towers on one stream
        segclip_amd.config.overlap_towers = False
    if os.environ.get("SEGCLIP_OVERLAP_WGRAD", "0") == "1":   # experiment switch: weight gradients on a second stream
        segclip_amd.config.overlap_wgrad = True
    torch.manual_seed(1234 + rank)
    model, targs = synth.build_model(spec, flags, rank=rank, world_size=world, device=dev)
    # the reference driver freezes these two (main_task_align.py:436-441)
    model.clip.visual.conv1.weight.requires_grad_(False)
    model.clip.visual.positional_embedding.requires_grad_(False)
    net = model
    if multi and a.grad_sync == "segclip":
        # gradient all-reduce of the data-parallel step (the reference: DDP, main_task_align.py:251-252): flat aligned
        # buckets filled directly by the weight-gradient kernels, exchanged on a communication stream as they complete
        from segclip_amd.dist import GradSync
        net = GradSync(model, compress={"auto": False, "bf16": True, "fp32": False}[a.wire])  # fp32 wire unless asked
    elif multi:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank,
                                                        find_unused_parameters=True, bucket_cap_mb=64, static_graph=True)
    batch = synth.synthetic_batch(spec, a.batch, seed=100 + rank, device=dev, with_seg=a.full_loss)

    params = [p for p in model.parameters()]   # (Module.zero_grad walks the module tree: 1.8 ms of host time per step)

    def step():
        for p in params:
            p.grad = None
        loss = net(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"],
                   image_seg=batch.get("image_seg"))
        loss.backward()
        return loss

    if os.environ.get("SEGCLIP_MAIN_HIGH", "0") == "1":   # experiment: the step on a high-priority stream (gap-filler streams stay normal)
        hs = torch.cuda.Stream(priority=-1)
        hs.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(hs)
    for _ in range(a.warmup):
        loss = step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    sampler = None
    if rank == 0:   # shader clock / socket power during the timed region (a reader thread on the host; no GPU work)
        try:
            from tools.clock_sampler import ClockSampler, device_bus_id
            sampler = ClockSampler(device_bus_id(local_rank)).start()
        except Exception:
            sampler = None
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    clocks = sampler.stop() if sampler is not None else None
    if multi:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss_val = float(loss.detach())
    ms = elapsed / a.steps * 1e3
    pairs = a.batch * world * a.steps / elapsed

    # ---- behind the timed region: the mode that meets north_star's 1e-3 / bit-exact bar, timed, and the
    # agreement of the benchmarked bf16 mode with it on the same inputs and the same injected noise.  Every rank runs the
    # same sequence (the steps contain the collectives); rank 0 reports.
    parity_mode = bf16_vs_f32 = gb2048 = None
    if not a.no_parity_leg and a.dtype == "bf16" and a.spec == "vitb16":
        noise = synth.synthetic_noise(spec, a.batch, seed=100 + rank, device=dev)
        items = [("gumbel", noise["gumbel_main"])]
        if a.full_loss:
            items += [("rand", noise["mask_noise"]), ("gumbel", noise["gumbel_mae"])]

        def probe():
            with segclip_amd.noise_injection(items):
                l_ = step()
            torch.cuda.synchronize()
            soft = model.last_mid_states["attns"][0]["soft_attn"] if model.last_mid_states.get("attns") else None
            return (float(l_.detach()), model.last_logits[0].float().clone(), model.last_mid_states["hard_idx"].clone(),
                    soft.float().clone() if soft is not None else None)

        lb, tb, hb, _ = probe()
        segclip_amd.set_compute_dtype(torch.float32)
        near_ties = None
        try:
            lf, tf_, hf, sf = probe()          # also the warm-up of the f32 kernels
            try:
                # patches whose two best NOISY assignment logits (log soft + Gumbel) are closer than the bound the B = 256 oracle test
                # accepts for a flipped patch: the places where fp32 summation order alone may turn the 8-way argmax against the CPU
                # reference ("bit-exact token indexing" holds everywhere else)
                gmb = noise["gumbel_main"].float()
                if sf is not None and sf.shape == gmb.shape:
                    y = sf.clamp_min(1e-38).log() + gmb
                    top2 = y.topk(2, dim=1).values
                    near_ties = int(((top2[:, 0] - top2[:, 1]) <= 2e-4 * top2[:, 0].abs().clamp_min(1.0)).sum())
            except Exception:
                near_ties = None
            del sf
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            t1 = time.perf_counter()
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            dt = (time.perf_counter() - t1) / 3
            # ... and the same mode with the Linear layers as bf16 x 3 products on the bf16 matrix pipe (config.f32_split), held to the
            # same parity bounds by the tests: timed, and compared with the exact run above on the same batch and noise
            parity_split = None
            try:
                segclip_amd.config.f32_split = True
                ls, ts_, hs, _ = probe()
                torch.cuda.synchronize()
                if multi:
                    dist.barrier()
                t2 = time.perf_counter()
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                if multi:
                    dist.barrier()
                dts = (time.perf_counter() - t2) / 3
                parity_split = {"dtype": "f32, Linear layers as A_hi B_hi + A_lo B_hi + A_hi B_lo on the bf16 matrix pipe (config.f32_split)",
                                "pairs_per_s": round(a.batch * world / dts, 1), "ms_per_step": round(dts * 1e3, 2), "steps": 3,
                                "vs_exact_f32": {"d_loss": round(abs(ls - lf), 7), "max_dlogit": round(float((ts_ - tf_).abs().max()), 6),
                                                 "hard_idx_differ": int((hs != hf).sum())},
                                "note": "tests/test_bench_size_gpu.py::test_b256_exact_f32_against_cpu_oracle[f32_split] and "
                                        "tests/test_model_gpu.py::test_vitb16_b4_f32_split_matches_reference_golden hold this mode to the "
                                        "same 1e-3 / index bounds as the exact one"}
                del ts_, hs
            finally:
                segclip_amd.config.f32_split = False
        finally:
            segclip_amd.set_compute_dtype(torch.bfloat16)
        parity_mode = {"dtype": "f32", "pairs_per_s": round(a.batch * world / dt, 1), "ms_per_step": round(dt * 1e3, 2), "steps": 3,
                       "note": "exact-f32 mode (v_mfma_f32_32x32x2_f32 GEMMs, fp32 activations): the mode the parity tests hold to "
                               "1e-3 / bit-exact indices against the reference; same model, same batch",
                       "hard_idx_near_ties_b256": near_ties, "hard_idx_patches": int(hf.numel()),
                       "hard_idx_note": "patches of this batch whose two best noisy assignment logits differ by <= 2e-4 (relative) in the "
                                        "exact-f32 run: only there may the 8-way argmax differ from the CPU reference by fp32 summation "
                                        "order (tests/test_bench_size_gpu.py::test_b256_exact_f32_against_cpu_oracle: every differing "
                                        "patch must be such a near-tie, at most 5; measured 1 of 50176)",
                       "f32_split": parity_split}
        bf16_vs_f32 = {"d_loss": round(abs(lb - lf), 6), "max_dlogit": round(float((tb - tf_).abs().max()), 4),
                       "hard_idx_agree": round(float((hb == hf).float().mean()), 5),
                       "note": "benchmarked bf16 mode against the exact-f32 mode, same batch and injected Gumbel noise; bf16 does not "
                               "meet the 1e-3 logit bar (disclosed in DESIGN.md 2)"}
        del tb, tf_, hb, hf
        if world == 1 and not a.full_loss and not a.global_batch and not a.force_dist and a.batch < 2048:
            # the metric's literal N = 1 point: global batch 2048 on one GPU (SURVEY 8d strong-scaling base)
            try:
                big = synth.synthetic_batch(spec, 2048, seed=100, device=dev, with_seg=False)

                def step_big():
                    for p in params:
                        p.grad = None
                    l_ = net(big["input_ids"], big["segment_ids"], big["input_mask"], big["image"])
                    l_.backward()
                    return l_
                step_big()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(3):
                    step_big()
                torch.cuda.synchronize()
                dtb = (time.perf_counter() - t1) / 3
                gb2048 = {"pairs_per_s": round(2048 / dtb, 1), "ms_per_step": round(dtb * 1e3, 2), "steps": 3,
                          "step_frac": round(2048 / dtb * GF_PER_PAIR[("vitb16", False)] / 1e3 / PEAK_BF16_TF, 4)}
                del big
                torch.cuda.empty_cache()
                step()                      # back to the benchmarked shape before the roofline leg counts a step
            except Exception as e:           # never lose the headline line over a secondary field
                gb2048 = {"error": f"{type(e).__name__}: {str(e)[:160]}"}

    text_trim = None
    if not a.no_parity_leg and a.dtype == "bf16" and a.spec == "vitb16" and not a.text_trim:
        # opt-in switch, reported beside the headline (never part of it): dead positions of the causal text tower skipped
        segclip_amd.config.text_trim = True
        try:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            t1 = time.perf_counter()
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            dtt = (time.perf_counter() - t1) / 10
            keep = model.clip._trim_len(batch["input_ids"].view(-1, batch["input_ids"].shape[-1]))
            text_trim = {"pairs_per_s": round(a.batch * world / dtt, 1), "ms_per_step": round(dtt * 1e3, 3), "steps": 10,
                         "text_positions": f"{keep} of {spec['context_length']}",
                         "note": "config.text_trim (opt-in, NOT the headline): the causal text tower runs on the positions up to the "
                                 "batch's last EOT; loss and every gradient equal the full pass (tests/test_model_gpu.py::"
                                 "test_text_trim_gives_the_same_loss_and_gradients); fewer contraction flops are executed"}
        finally:
            segclip_amd.config.text_trim = False
        step()
        torch.cuda.synchronize()

    roofline = None
    if not a.no_roofline and a.dtype == "bf16":
        # EVERY rank runs the op-count pass (one more step: its collectives must be matched on all ranks - ADVICE r3: with
        # rank 0 alone in it, rank 0 hung in the embedding all-gather / bucket all-reduces); only rank 0 starts the
        # rocprofv3 children, which run this rank's per-GPU workload as a single-GPU process (no collectives)
        work = count_step_work(step)
        if rank == 0:
            roofline = roofline_block(a, step, pairs / world, world, work, collective_free=world == 1)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline()
    if rank == 0:
        mname = MODEL_NAME.get(a.spec, a.spec)
        out = {"metric": ("image-text pairs/s fwd+bwd, %s, global batch %d (per-GPU %d)" % (mname, a.batch * world, a.batch)
                          if a.global_batch else
                          "image-text pairs/s fwd+bwd, %s, per-GPU batch %d (global %d at 8 GPUs)" % (mname, a.batch, 8 * a.batch)),
               "value": round(pairs, 1), "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "strong" if a.global_batch else "weak",
               "vs_baseline": None,
               "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": ("BASELINE configs[4]: ViT-L/14 336^2 + 77-token text, contrastive loss only"
                                       if a.spec == "vitl14_336" else
                                       "BASELINE configs[3]: ViT-B/16 224^2 + 77-token text, full SegCLIP loss"
                                       if a.full_loss else
                                       "BASELINE configs[1]/[2]: ViT-B/16 224^2 + 77-token text, contrastive loss only"),
                          "per_gpu_batch": a.batch, "global_batch": a.batch * world, "parallelism": f"dp{world}",
                          "attention_forward": ("fp8 e4m3 MFMA (QK^T, PV), per-token Q/K scales" if attn_fp8 else
                                                a.dtype + (" (--attn-fp8 auto = off: the non-scaled e4m3 MFMA has the bf16 rate and its "
                                                           "quantisation passes cost more than they save at head_dim 64)"
                                                           if a.spec == "vitl14_336" and a.attn_fp8 == "auto" else "")),
                          "grad_exchange": (None if not multi else "torch DDP fp32" if a.grad_sync == "ddp" else
                                            f"GradSync {len(net._flat)} buckets, wire " +
                                            ("bf16" if net._flat and net._use_bf16(net._flat[0]) else "fp32") +
                                            f", zero-copy grads {net.stats['zero_copy']}/{net.stats['zero_copy'] + net.stats['copies']}"
                                            f", backend {a.backend}, RCCL channels {a.rccl_channels or 'default'}"),
                          "residual_stream": ("bf16 between the blocks of a tower, fp32 at the tower boundaries"
                                              if (a.dtype == "bf16" and segclip_amd.config.bf16_resid) else "fp32"),
                          "cross_mode": segclip_amd.config.cross_mode,
                          "text_positions": ("trimmed to the batch's last EOT (--text-trim)" if a.text_trim else
                                             f"all {spec['context_length']} (full context, as the reference computes it)"),
                          "loss": round(loss_val, 5)},
               "parity_mode": parity_mode, "bf16_vs_f32": bf16_vs_f32, "global_batch_2048_single_gpu": gb2048, "text_trim": text_trim,
               "roofline": roofline, "cpu_baseline": cpu}
        if clocks:
            # the chip's state during the timed region: under this step the socket sits at its power cap and the shader clock
            # below the 2.4 GHz the 2.5 PFLOP/s dense-bf16 peak is quoted at; peak_at_clock rescales that peak linearly
            clocks["note"] = ("amdsmi GFX clock / socket power sampled every 20 ms during the timed region (first fifth dropped); "
                              "roofline.peak stays the 2.4 GHz figure, peak_bf16_tf_at_sclk = 2500 x sclk / 2400")
            clocks["peak_bf16_tf_at_sclk"] = round(PEAK_BF16_TF * clocks["sclk_mhz_mean"] / 2400.0, 1)
            if roofline and roofline.get("achieved"):
                clocks["roofline_frac_at_sclk"] = round(roofline["achieved"] / clocks["peak_bf16_tf_at_sclk"], 4)
            out["clocks"] = clocks
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()   # nobody tears the group down while another rank may still be inside a collective
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
