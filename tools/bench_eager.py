"""Same-node yardstick for BASELINE configs[1] ("bf16 HIP attention/MLP kernels vs PyTorch-ROCm eager"):
the oracle's module math (oracle/segclip_oracle.py - the CPU restatement of the reference's forward) run on the GPU by
PyTorch-ROCm EAGER ops under bf16 autocast, with F.scaled_dot_product_attention as the attention core, forward + backward at
the bench batch.  This is a TOOL: the product (segclip_amd/) never imports oracle/; nothing here is part of bench.py's timed
region.  It prints one JSON object {pairs/s, ms/step, loss, top kernels (with --rocprof-child)} and, with --classes, per-op
rows (torch layer_norm fwd/bwd, SDPA fwd/bwd at the bench shapes) to set beside profiles/r04_hbm_kernels.txt / r04_attn.txt.

  python tools/bench_eager.py [--batch 256] [--steps 10] [--warmup 3] [--device cuda] [--spec vitb16] [--classes]
"""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from oracle import segclip_oracle as so
from segclip_amd import synth
from tests.helpers import model_param_shapes, oracle_params


def sdpa_core(q, k, v, n_head, causal=False, key_mask=None):
    """mha_core of the oracle with torch's fused attention (what an eager PyTorch-ROCm user would run)."""
    B, Tq, D = q.shape
    Tk = k.shape[1]
    hd = D // n_head
    qh = q.reshape(B, Tq, n_head, hd).permute(0, 2, 1, 3)
    kh = k.reshape(B, Tk, n_head, hd).permute(0, 2, 1, 3)
    vh = v.reshape(B, Tk, n_head, hd).permute(0, 2, 1, 3)
    mask = None
    if key_mask is not None:
        mask = ((1.0 - key_mask.to(qh.dtype)) * -1000000.0)[:, None, None, :]
    o = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=mask, is_causal=bool(causal) and mask is None)
    return o.permute(0, 2, 1, 3).reshape(B, Tq, D)


def timed(fn, steps, warmup, dev):
    for _ in range(warmup):
        fn()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out


def class_rows(dev, B):
    """torch eager rows for the HBM-bound / attention kernel classes at the bench shapes (us per call)."""
    rows = {}
    bf = torch.bfloat16

    def ev(fn, reps=10):
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    for name, M, D in (("vision", B * 196, 768), ("text", B * 77, 512)):
        x = torch.randn(M, D, device=dev, requires_grad=True)
        w, b = torch.ones(D, device=dev, requires_grad=True), torch.zeros(D, device=dev, requires_grad=True)
        g = torch.randn(M, D, device=dev)
        rows[f"layer_norm fwd fp32 {name} ({M}x{D})"] = ev(lambda: F.layer_norm(x, (D,), w, b))
        y = F.layer_norm(x, (D,), w, b)
        rows[f"layer_norm bwd fp32 {name} ({M}x{D})"] = ev(lambda: torch.autograd.grad(y, (x, w, b), g, retain_graph=True))
    for name, T, H, causal in (("vision T=196", 196, 12, False), ("text T=77 causal", 77, 8, True)):
        q, k, v = (torch.randn(B, H, T, 64, device=dev, dtype=bf, requires_grad=True) for _ in range(3))
        go = torch.randn(B, H, T, 64, device=dev, dtype=bf)
        rows[f"SDPA fwd bf16 {name}"] = ev(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=causal))
        o = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
        rows[f"SDPA bwd bf16 {name}"] = ev(lambda: torch.autograd.grad(o, (q, k, v), go, retain_graph=True))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--spec", default="vitb16")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-sdpa", action="store_true", help="keep the oracle's explicit softmax(QK^T)V")
    ap.add_argument("--classes", action="store_true")
    a = ap.parse_args()
    dev = torch.device(a.device)
    spec = synth.SPECS[a.spec]
    if not a.no_sdpa:
        so.mha_core = sdpa_core
    P = oracle_params(spec, model_param_shapes(spec, {}))
    P = {k: v.detach().to(dev).requires_grad_(v.requires_grad) for k, v in P.items()}
    for k in ("clip.visual.conv1.weight", "clip.visual.positional_embedding"):   # frozen by the reference driver
        P[k].requires_grad_(False)
    batch = {k: v.to(dev) for k, v in synth.synthetic_batch(spec, a.batch, seed=100).items()}
    noise = {k: v.to(dev) for k, v in synth.synthetic_noise(spec, a.batch, seed=100).items()}

    def step():
        for p in P.values():
            p.grad = None
        with torch.autocast(dev.type, dtype=torch.bfloat16, enabled=a.dtype == "bf16"):
            loss, _ = so.segclip_forward(batch, P, spec, noise, {})
        loss.backward()
        return loss

    dt, loss = timed(step, a.steps, a.warmup, dev)
    out = {"what": "PyTorch-ROCm eager (oracle module math" + ("" if a.no_sdpa else " + SDPA") + f"), {a.dtype} "
                   + ("autocast" if a.dtype == "bf16" else ""), "spec": a.spec, "batch": a.batch,
           "pairs_per_s": round(a.batch / dt, 1), "ms_per_step": round(dt * 1e3, 3), "loss": round(float(loss), 5),
           "torch": torch.__version__, "device": torch.cuda.get_device_name(0) if dev.type == "cuda" else "cpu"}
    if a.classes and dev.type == "cuda":
        out["class_rows_us"] = {k: round(v, 1) for k, v in class_rows(dev, a.batch).items()}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
