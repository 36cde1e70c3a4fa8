"""bf16 training step against the exact-f32 mode at the bench size (ViT-B/16, B = 256): loss, logits, hard_idx / MAE
index agreement and, per parameter, gradient cosine + norm ratio - with the bf16 residual STREAM (config.bf16_resid) off
and on, in both cross-attention modes ("intended": each sample attends to its own tokens, so an argmax flip stays inside
its sample; "t18": the mode the bench runs, where a flip reaches other samples), contrastive-only (BASELINE configs[1])
and, in "intended" mode, the full loss (configs[3]).  The printed numbers are what tests/test_bench_size_gpu.py's bounds
are derived from (committed: profiles/r04_accuracy_b256.txt)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import segclip_amd
from tests.test_bench_size_gpu import _run
from tests.helpers import FULL_FLAGS

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def compare(f, b, tag):
    dl = abs(f["loss"] - b["loss"])
    dlog = float((f["t2v"] - b["t2v"]).abs().max())
    agree = float((f["hard_idx"] == b["hard_idx"]).float().mean())
    ratios = {n: b["gn"][n] / f["gn"][n] for n in f["gn"] if f["gn"][n] > 1e-6}
    cos = {n: float((f["grads"][n].double() * b["grads"][n].double()).sum()) / (f["gn"][n] * b["gn"][n]) for n in ratios}
    mats = [n for n in ratios if b["dim"][n] >= 2]
    vecs = [n for n in ratios if b["dim"][n] < 2]
    print(f"== {tag}: loss {b['loss']:.5f} vs f32 {f['loss']:.5f} (d {dl:.2e}); max |dlogit| {dlog:.4f}; hard_idx agreement {agree:.4f}")
    for name, grp in (("matrices", mats), ("vectors", vecs)):
        c = np.array([cos[n] for n in grp]); r = np.array([ratios[n] for n in grp])
        print(f"   {name:8s} n={len(grp):3d}  cosine min {c.min():.4f} p5 {np.percentile(c, 5):.4f} median {np.median(c):.4f} | "
              f"norm ratio min {r.min():.4f} median {np.median(r):.4f} max {r.max():.4f}")
    for n in sorted(mats, key=lambda n: cos[n])[:6]:
        print(f"      worst matrix  {n}: cos {cos[n]:.4f} ratio {ratios[n]:.4f}")
    for n in sorted(vecs, key=lambda n: cos[n])[:4]:
        print(f"      worst vector  {n}: cos {cos[n]:.4f} ratio {ratios[n]:.4f}")
    # per-depth profile of the vision tower's c_fc weights (the chain's error would grow towards block 0)
    prof = []
    for i in range(10):
        n = f"clip.visual.transformer.layers0.{i}.mlp.c_fc.weight"
        if n in cos:
            prof.append(f"{i}:{cos[n]:.3f}")
    print("      vision c_fc cosine by block  " + " ".join(prof))
    if "ids_restore" in f and "ids_restore" in b:
        print(f"   MAE ids_restore equal: {bool(torch.equal(f['ids_restore'], b['ids_restore']))}")


for flags, name, modes in (({}, "contrastive only (configs[1])", ("intended", "t18")),
                           (FULL_FLAGS, "full SegCLIP loss (configs[3])", ("intended",))):
    for cm in modes:
        f = _run("vitb16", B, 3, torch.float32, flags, cm, keep_grads=True)
        for resid in (False, True):
            segclip_amd.config.bf16_resid = resid
            b = _run("vitb16", B, 3, torch.bfloat16, flags, cm, keep_grads=True)
            compare(f, b, f"B={B} {name}, cross_mode={cm}, bf16_resid={'on' if resid else 'off'}")
            del b
            torch.cuda.empty_cache()
        segclip_amd.config.bf16_resid = False
        del f
        torch.cuda.empty_cache()
