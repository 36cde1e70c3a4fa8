"""Per-shape breakdown of every GEMM launch of one bench step (HIP events on the launch stream)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import segclip_amd
from segclip_amd import ops, synth
segclip_amd.set_compute_dtype(torch.bfloat16)
spec = synth.SPECS["vitb16"]
model, _ = synth.build_model(spec, {}, device="cuda")
model.clip.visual.conv1.weight.requires_grad_(False); model.clip.visual.positional_embedding.requires_grad_(False)
batch = synth.synthetic_batch(spec, 256, seed=1, device="cuda", with_seg=False)
def step():
    model.zero_grad(set_to_none=True)
    loss = model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"]); loss.backward()
for _ in range(2): step()
orig = ops.p_gemm
shapes = []
def wrapped(A, B, Cc, M, N, K, sa, sb, ldc, **kw):
    shapes.append((M, N, K, sa[1] == 1, sb[1] == 1, kw.get("nb1", 1) * kw.get("nb2", 1), str(A.dtype)[-4:], str(Cc.dtype)[-4:],
                   kw.get("residual") is not None, kw.get("act", 0), bool(kw.get("mul_dact", False))))
    return orig(A, B, Cc, M, N, K, sa, sb, ldc, **kw)
ops.p_gemm = wrapped
ops._GemmProfile.start(); step(); rec = ops._GemmProfile.stop()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for s, (t, f, _, _nb) in zip(shapes, rec):
    a = agg[s]; a[0] += 1; a[1] += t; a[2] += f
tot = sum(a[1] for a in agg.values())
print(f"total GEMM time {tot*1e3:.2f} ms, {len(rec)} launches")
print("M N K | A_kcontig B_kcontig batch Adt Cdt res act dact | calls  total_ms  avg_us  TF/s")
for s, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(s, a[0], f"{a[1]*1e3:8.3f} {a[1]/a[0]*1e6:8.1f} {a[2]/a[1]/1e12:7.1f}")
