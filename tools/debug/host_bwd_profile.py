"""Host time inside the autograd thread: cProfile around ops.ResStackFn.backward (the towers' hand-scheduled backward)."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, segclip_amd
from segclip_amd import synth, ops
segclip_amd.set_compute_dtype(torch.bfloat16)
spec = synth.SPECS["vitb16"]
model, _ = synth.build_model(spec, {}, device="cuda")
model.clip.visual.conv1.weight.requires_grad_(False); model.clip.visual.positional_embedding.requires_grad_(False)
batch = synth.synthetic_batch(spec, int(os.environ.get("B", 64)), seed=1, device="cuda", with_seg=False)
pr = cProfile.Profile()
orig = ops.ResStackFn.backward
def wrapped(ctx, g):
    pr.enable()
    try:
        return orig(ctx, g)
    finally:
        pr.disable()
ops.ResStackFn.backward = staticmethod(wrapped)
def step():
    for p in model.parameters(): p.grad = None
    loss = model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"]); loss.backward()
for _ in range(3): step()
torch.cuda.synchronize(); pr.clear()
for _ in range(5): step()
torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30); print(s.getvalue()[:7000])
