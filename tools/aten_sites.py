"""Which Python lines of the training step still launch ATen kernels / runtime copies?  One profiled fwd+bwd pass at
BATCH (default 256) with stacks: every device kernel whose name is not one of the library's, grouped by the innermost
segclip_amd frame of the op that launched it (backward ops: the frame of the autograd Function's backward, or "autograd
engine" for torch's own nodes), with launches per pass."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import segclip_amd
from segclip_amd import synth
from torch.profiler import profile, ProfilerActivity

B = int(os.environ.get("BATCH", "256"))
segclip_amd.set_compute_dtype(torch.bfloat16)
spec = synth.SPECS["vitb16"]
model, _ = synth.build_model(spec, {}, device="cuda")
b = synth.synthetic_batch(spec, B, seed=0, device="cuda", with_seg=False)


def step():
    model.zero_grad(set_to_none=True)
    loss = model(b["input_ids"], b["segment_ids"], b["input_mask"], b["image"])
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()

evs = prof.events()
ours = ("(anonymous namespace)", "segclip")
sites = collections.Counter()
times = collections.Counter()
total = collections.Counter()
for e in evs:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    for k in e.kernels:
        total["all"] += 1
        if any(o in k.name for o in ours) and "at::" not in k.name:
            continue
        frame = "autograd engine / no python frame"
        for f in (e.stack or []):
            if "segclip_amd" in f:
                frame = f.split("segclip_amd/")[-1]
                break
        key = (k.name[:70], e.name, frame)
        sites[key] += 1
        times[key] += k.duration
print(f"B={B}: {total['all'] / N:.0f} kernels per pass, {sum(sites.values()) / N:.0f} of them not the library's")
for key, c in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(f"{c / N:6.1f}x {times[key] / N:8.1f} us  {key[0]:70s} {key[1]:28s} {key[2]}")
