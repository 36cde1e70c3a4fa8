"""bench.py - image-text pairs/s, forward+backward, of the SegCLIP contrastive hot path on MI355X.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N>1 launched through
torch.distributed.run, one rank per GPU over RCCL.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1] at N=1, configs[2] at N=8): ViT-B/16 224^2 + 77-token text,
contrastive loss only, per-GPU batch 256 (weak scaling: global batch 256*N, 2048 at N=8), bf16 MFMA
kernels, fp32 master weights, synthetic images/captions, closed-form random weights.  A step is one
forward + backward (loss -> every parameter gradient, incl. the embedding all-gather and, for N>1, the
DDP gradient all-reduce); the optimizer step is excluded, as in BASELINE.json's metric definition.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import segclip_amd  # noqa: E402
from segclip_amd import ops, synth  # noqa: E402

GF_PER_PAIR_FWD_BWD = 109.675  # ViT-B/16 contrastive-only, contractions only
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
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the N>1 code path (RCCL group, GradSync, barriers) even with one rank: single-GPU check of it")
    ap.add_argument("--grad-sync", default="segclip", choices=["segclip", "ddp"],
                    help="N>1 gradient exchange: segclip_amd.dist.GradSync (default) or torch DDP (comparison only)")
    ap.add_argument("--wire", default="auto", choices=["auto", "bf16", "fp32"], help="gradient all-reduce wire format")
    return ap.parse_args()


def cpu_baseline():
    """Reference CPU path (the oracle = validated CPU restatement of the reference) on this host's cores:
    BASELINE.json configs[0]: ViT-B/16 + 77-token text, batch 4, contrastive only, fwd+bwd."""
    from oracle import segclip_oracle as so
    from tests.helpers import model_param_shapes, oracle_params
    spec = synth.SPECS["vitb16"]
    cores = min(os.cpu_count() or 1, 32)  # torch CPU eager stops scaling (and thrashes) far below 256 threads
    torch.set_num_threads(cores)
    P = oracle_params(spec, model_param_shapes(spec, {}))
    B, timed = 4, 0
    batch = synth.synthetic_batch(spec, B, seed=2, with_seg=False)
    noise = synth.synthetic_noise(spec, B, seed=2)
    total = 0.0
    while True:  # 1 warm-up, then timed steps for ~12 s (bounded: the GPU box is billed for this too)
        for p in P.values():
            p.grad = None
        t0 = time.perf_counter()
        loss, _ = so.segclip_forward(batch, P, spec, noise, {})
        loss.backward()
        dt = time.perf_counter() - t0
        if timed or total:
            total += dt
            timed += 1
        else:
            total = 1e-12  # warm-up done
        if total > 12.0 or timed >= 40:
            break
    return {"value": round(B * timed / total, 3), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"oracle (CPU restatement of the reference, torch {torch.__version__} fp32), BASELINE configs[0]: "
                      f"ViT-B/16 B=4 contrastive-only fwd+bwd, {timed} timed steps after 1 warm-up, "
                      f"{total / timed:.2f} s/step",
            "loss": round(float(loss), 6)}


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


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        n_dev = torch.cuda.device_count()
        if n_dev < a.gpus:
            raise SystemExit(f"--gpus {a.gpus} but only {n_dev} GPU(s) visible")
        respawn(a)
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.global_batch:
        if a.global_batch % world:
            raise SystemExit(f"--global-batch {a.global_batch} is not divisible by {world} ranks")
        a.batch = a.global_batch // world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or a.force_dist
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    spec = synth.SPECS[a.spec]
    flags = dict(use_seglabel=True, use_vision_mae_recon=True) if a.full_loss else {}
    segclip_amd.set_compute_dtype(torch.bfloat16 if a.dtype == "bf16" else torch.float32)
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
        net = GradSync(model, compress={"auto": "auto", "bf16": True, "fp32": False}[a.wire])
    elif multi:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank,
                                                        find_unused_parameters=True, bucket_cap_mb=64, static_graph=True)
    batch = synth.synthetic_batch(spec, a.batch, seed=100 + rank, device=dev, with_seg=a.full_loss)

    def step():
        net.zero_grad(set_to_none=True)
        loss = net(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"],
                   image_seg=batch.get("image_seg"))
        loss.backward()
        return loss

    for _ in range(a.warmup):
        loss = step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss_val = float(loss)
    ms = elapsed / a.steps * 1e3
    pairs = a.batch * world * a.steps / elapsed

    roofline = None
    if not a.no_roofline and a.dtype == "bf16":
        # dominant kernel = gemm_bf16_kernel (all layouts): per-launch HIP-event timing on the launch stream
        segclip_amd.config.overlap_towers = False  # per-kernel durations without the concurrent text stream
        step()
        ops._GemmProfile.start()
        step()
        rec = [r for r in ops._GemmProfile.stop() if r[2]]
        segclip_amd.config.overlap_towers = True
        tsum = sum(r[0] for r in rec)
        fsum = sum(r[1] for r in rec)
        big = [r for r in rec if r[1] >= 1e11]
        traffic = None  # HBM bytes per GEMM launch from the committed rocprofv3 PMC passes (profiles/)
        tf = os.path.join(ROOT, "profiles", "r01_gemm_traffic.json")
        if a.batch == 256 and a.spec == "vitb16" and not a.full_loss and os.path.exists(tf):
            traffic = round(json.load(open(tf))["hbm_bytes_per_launch"])
        roofline = {"bound": "mfma", "kernel": "gemm_bf16_dma_kernel (+gemm_bf16_kernel fallback)",
                    "achieved": round(fsum / tsum / 1e12, 1),
                    "peak": PEAK_BF16_TF, "unit": "TFLOP/s", "frac": round(fsum / tsum / 1e12 / PEAK_BF16_TF, 4),
                    "traffic": traffic, "algorithmic_flops_per_launch": round(fsum / len(rec)), "launches_per_step": len(rec), "avg_launch_us": round(tsum / len(rec) * 1e6, 1),
                    "gemm_share_of_step": round(tsum * 1e3 / ms, 3),
                    "large_gemm_tflops": round(sum(r[1] for r in big) / max(sum(r[0] for r in big), 1e-9) / 1e12, 1),
                    "step_frac": round(pairs / world * GF_PER_PAIR_FWD_BWD / 1e3 / PEAK_BF16_TF, 4)}
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline()
    if rank == 0:
        out = {"metric": "image-text pairs/s fwd+bwd, ViT-B/16 224^2, per-GPU batch 256 (global 2048 at 8 GPUs)",
               "value": round(pairs, 1), "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "strong" if a.global_batch else "weak",
               "vs_baseline": None,
               "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": ("BASELINE configs[3]: ViT-B/16 224^2 + 77-token text, full SegCLIP loss"
                                       if a.full_loss else
                                       "BASELINE configs[1]/[2]: ViT-B/16 224^2 + 77-token text, contrastive loss only"),
                          "per_gpu_batch": a.batch, "global_batch": a.batch * world, "parallelism": f"dp{world}",
                          "grad_exchange": (None if not multi else "torch DDP fp32" if a.grad_sync == "ddp" else
                                            f"GradSync {len(net._flat)} buckets, wire " +
                                            ("bf16" if net._flat and net._use_bf16(net._flat[0]) else "fp32") +
                                            f", zero-copy grads {net.stats['zero_copy']}/{net.stats['zero_copy'] + net.stats['copies']}"),
                          "cross_mode": segclip_amd.config.cross_mode, "loss": round(loss_val, 5)},
               "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
