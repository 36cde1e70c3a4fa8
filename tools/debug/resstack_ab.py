"""Same-box A/B of the tower execution modes: one autograd node per block | ResStackFn with fp32 residual gradient |
ResStackFn with the bf16 residual-gradient chain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import segclip_amd
from segclip_amd import synth
spec = synth.SPECS["vitb16"]
segclip_amd.set_compute_dtype(torch.bfloat16)
model, _ = synth.build_model(spec, {}, device="cuda")
model.clip.visual.conv1.weight.requires_grad_(False)
model.clip.visual.positional_embedding.requires_grad_(False)
batch = synth.synthetic_batch(spec, 256, seed=100, device="cuda", with_seg=False)
def bench(steps=8, warm=3):
    def step():
        model.zero_grad(set_to_none=True)
        loss = model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"]); loss.backward()
        return loss
    for _ in range(warm): step()
    torch.cuda.synchronize(); t0 = time.perf_counter(); h = 0.0
    for _ in range(steps):
        t1 = time.perf_counter(); loss = step(); h += time.perf_counter() - t1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, h / steps * 1e3, float(loss)
for rep in range(2):
    for name, fuse, chain in (("per-block nodes", False, False), ("stack, fp32 resgrad", True, False), ("stack, bf16 resgrad", True, True)):
        segclip_amd.config.fuse_res_stack, segclip_amd.config.bf16_resgrad = fuse, chain
        ms, host, loss = bench()
        print(f"{name:22s} {ms:6.2f} ms/step  host {host:5.2f} ms  loss {loss:.5f}", flush=True)
