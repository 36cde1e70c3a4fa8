"""How far ahead of the GPU is the host at the end of the forward / backward enqueue?  (un-profiled; events + perf_counter)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import segclip_amd
from segclip_amd import synth
B = int(os.environ.get("B", 256))
dev = torch.device("cuda", 0)
spec = synth.SPECS["vitb16"]
segclip_amd.set_compute_dtype(torch.bfloat16)
torch.manual_seed(1234)
model, targs = synth.build_model(spec, {}, rank=0, world_size=1, device=dev)
model.clip.visual.conv1.weight.requires_grad_(False)
model.clip.visual.positional_embedding.requires_grad_(False)
batch = synth.synthetic_batch(spec, B, seed=100, device=dev, with_seg=False)
def fwd():
    return model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"], image_seg=batch.get("image_seg"))
for _ in range(4):
    model.zero_grad(set_to_none=True); fwd().backward()
torch.cuda.synchronize()
rows = []
for it in range(6):
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    t0 = time.perf_counter(); e0.record()
    model.zero_grad(set_to_none=True)
    loss = fwd()
    t1 = time.perf_counter(); e1.record()
    loss.backward()
    t2 = time.perf_counter(); e2.record()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    rows.append(((t1 - t0) * 1e3, e0.elapsed_time(e1), (t2 - t1) * 1e3, e1.elapsed_time(e2), (t3 - t0) * 1e3))
    del loss
for r in rows:
    print("host fwd enqueue %.2f ms | GPU fwd done at %.2f ms || host bwd enqueue %.2f ms | GPU bwd span %.2f ms || step %.2f ms" % r)
