"""Kernel time of the learnable-center stage alone (SemanticLearnerModule forward + backward at B = 256, bf16): which of its
~90 launches carry the time?  (torch profiler kernel table)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, segclip_amd
from segclip_amd import synth
segclip_amd.set_compute_dtype(torch.bfloat16)
segclip_amd.set_cross_mode(os.environ.get("CROSS_MODE", "t18"))
spec = synth.SPECS["vitb16"]
model, _ = synth.build_model(spec, {}, device="cuda")
sl = model.clip.visual.transformer.semantic_layer2
x = torch.randn(256, 196, 768, device="cuda", requires_grad=True)
def step():
    for p in sl.parameters(): p.grad = None
    x.grad = None
    out, hard, soft, q = sl(x)
    (out.float().sum() * 1e-3).backward()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total) for e in prof.key_averages()]
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
print(f"total kernel time per pass {tot / 3 / 1e3:.3f} ms over {sum(r[1] for r in rows) / 3:.0f} launches")
for k, c, t in rows[:40]:
    print(f"{t / 3:9.1f} us/pass  {c / 3:5.1f}x  {k[:110]}")
if os.environ.get("DETAIL"):
    # launch order of the LAST pass with per-kernel duration and the idle gap before it
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    evs.sort(key=lambda e: e.time_range.start)
    n = len(evs) // 3
    last = evs[2 * n:]
    prev_end = last[0].time_range.start
    print(f"--- last pass: {len(last)} launches, span {(last[-1].time_range.end - last[0].time_range.start) / 1e3:.3f} ms")
    for e in last:
        print(f"{e.time_range.end - e.time_range.start:8.1f} us  gap {e.time_range.start - prev_end:7.1f}  {e.name[:100]}")
        prev_end = e.time_range.end
