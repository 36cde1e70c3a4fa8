"""Every GEMM launch of one FULL-LOSS step (BASELINE configs[3]) by shape / dtype / caller, timed by events on the launch stream
(each launch alone: the towers' overlap is off for the measurement)."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import segclip_amd
from segclip_amd import ops, synth
segclip_amd.set_compute_dtype(torch.bfloat16)
segclip_amd.config.overlap_towers = False
# usage: full_gemms.py [batch [spec [full_loss 0|1]]]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spec = synth.SPECS[sys.argv[2] if len(sys.argv) > 2 else "vitb16"]
FULL = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
model, _ = synth.build_model(spec, dict(use_seglabel=True, use_vision_mae_recon=True) if FULL else {}, device="cuda")
model.clip.visual.conv1.weight.requires_grad_(False); model.clip.visual.positional_embedding.requires_grad_(False)
batch = synth.synthetic_batch(spec, B, seed=1, device="cuda", with_seg=FULL)
def step():
    model.zero_grad(set_to_none=True)
    loss = model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"], image_seg=batch.get("image_seg"))
    loss.backward()
for _ in range(2): step()
orig = ops.p_gemm
shapes = []
def wrapped(A, Bm, Cc, M, N, K, sa, sb, ldc, **kw):
    fr = [f for f in traceback.extract_stack()[:-1] if "segclip_amd" in f.filename]
    who = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:][::-1])
    shapes.append((M, N, K, sa[1] == 1, sb[1] == 1, kw.get("nb1", 1) * kw.get("nb2", 1), str(A.dtype)[-4:], str(Cc.dtype)[-4:],
                   kw.get("residual") is not None, kw.get("act", 0), bool(kw.get("mul_dact", False)), who))
    return orig(A, Bm, Cc, M, N, K, sa, sb, ldc, **kw)
ops.p_gemm = wrapped
ops._GemmProfile.start(); step(); rec = ops._GemmProfile.stop()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
assert len(shapes) == len(rec), (len(shapes), len(rec))
for s, r in zip(shapes, rec):
    a = agg[s]; a[0] += 1; a[1] += r[0]; a[2] += r[1]
tot = sum(a[1] for a in agg.values())
print(f"total GEMM time {tot*1e3:.2f} ms, {len(rec)} launches (launches inside segclip_resblock_fwd / grouped weight gradients are not p_gemm calls)")
print("M N K | A_kcontig B_kcontig batch Adt Cdt res act dact caller | calls  total_ms  avg_us  TF/s")
for s, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(s[:11], a[0], f"{a[1]*1e3:8.3f} {a[1]/a[0]*1e6:8.1f} {a[2]/max(a[1],1e-9)/1e12:7.1f}", s[11])
print("fp32-operand GEMMs:")
for s, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if s[6] == 'at32':
        print(s[:11], a[0], f"{a[1]*1e3:8.3f} {a[1]/a[0]*1e6:8.1f} {a[2]/max(a[1],1e-9)/1e12:7.1f}", s[11])
