"""Where the host time of a step goes (cProfile over 10 steps of the contrastive step at B = 64: the host-bound configuration)."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, segclip_amd
from segclip_amd import synth
segclip_amd.set_compute_dtype(torch.bfloat16)
spec = synth.SPECS["vitb16"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model, _ = synth.build_model(spec, {}, device="cuda")
b = synth.synthetic_batch(spec, B, seed=0, device="cuda", with_seg=False)
params = list(model.parameters())
def step():
    for p in params: p.grad = None
    loss = model(b["input_ids"], b["segment_ids"], b["input_mask"], b["image"])
    loss.backward()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"B={B}: enqueue {(t1 - t0) * 100:.2f} ms per step, done {(t2 - t0) * 100:.2f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(10): step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(38)
print(s.getvalue()[:9000])
