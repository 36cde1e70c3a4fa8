"""Which Python call sites issue device-to-device memcpys (aten::copy_ / clone / contiguous) during one training step?"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import segclip_amd
from segclip_amd import synth
segclip_amd.set_compute_dtype(torch.bfloat16)
spec = synth.SPECS["vitb16"]
model, _ = synth.build_model(spec, {}, device="cuda")
model.clip.visual.conv1.weight.requires_grad_(False); model.clip.visual.positional_embedding.requires_grad_(False)
batch = synth.synthetic_batch(spec, 256, seed=1, device="cuda", with_seg=False)
def step():
    model.zero_grad(set_to_none=True)
    loss = model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"]); loss.backward()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
cnt = collections.Counter()
for e in ev:
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::zeros", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul", "aten::div", "aten::sum", "aten::to", "aten::_to_copy"):
        st = [s for s in (e.stack or []) if "segclip_amd" in s or "bench" in s or "autograd" in s]
        site = st[0] if st else "(no python frame: autograd engine)"
        shp = str(e.input_shapes)[:60] if e.input_shapes else ""
        cnt[(e.name, site[-90:], shp)] += 1
for (name, site, shp), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:4d}  {name:18s} {site}  {shp}")
