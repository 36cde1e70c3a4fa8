"""Whole-step hipGraph capture probe: forward + backward of the bench model under torch.cuda.graph, replay timing vs eager."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import segclip_amd
from segclip_amd import synth
B = int(os.environ.get("B", 256))
dev = torch.device("cuda", 0)
spec = synth.SPECS["vitb16"]
segclip_amd.set_compute_dtype(torch.bfloat16)
segclip_amd.config.overlap_towers = os.environ.get("OVERLAP", "1") == "1"
MODE = os.environ.get("MODE", "full")
print("MODE", MODE, "overlap_towers", segclip_amd.config.overlap_towers, flush=True)
torch.manual_seed(1234)
model, targs = synth.build_model(spec, {}, rank=0, world_size=1, device=dev)
model.clip.visual.conv1.weight.requires_grad_(False)
model.clip.visual.positional_embedding.requires_grad_(False)
batch = synth.synthetic_batch(spec, B, seed=100, device=dev, with_seg=False)

def step():
    model.zero_grad(set_to_none=True)
    loss = model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"], image_seg=batch.get("image_seg"))
    loss.backward()
    return loss

def timeit(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for _ in range(4): l0 = step()
print("eager ms/step", round(timeit(step), 3), "loss", float(l0.detach()))
del l0
g0 = {n: p.grad.detach().clone() for n, p in list(model.named_parameters())[:400] if p.grad is not None}
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
model.zero_grad(set_to_none=True)
try:
    with torch.cuda.graph(g):
        loss = model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"], image_seg=batch.get("image_seg"))
        if MODE == "full":
            loss.backward()
except Exception as e:
    import traceback; traceback.print_exc()
    print("CAPTURE FAILED:", type(e).__name__, str(e)[:500])
    sys.exit(1)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("graph ms/step", round(timeit(g.replay), 3), "loss", float(loss))
worst = 0.0
for n, p in model.named_parameters():
    if n in g0 and p.grad is not None:
        d = float((p.grad.float() - g0[n].float()).norm() / (g0[n].float().norm() + 1e-20))
        worst = max(worst, d)
print("max rel grad diff eager vs graph:", worst)
