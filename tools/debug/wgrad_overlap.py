"""Step time with the weight-gradient GEMMs on a side stream (config.overlap_wgrad) vs on the main stream."""
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
        model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"]).backward()
    for _ in range(warm): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for ow in (False, True, False, True):
    segclip_amd.config.overlap_wgrad = ow
    print(f"overlap_wgrad={ow}: {bench():.2f} ms/step", flush=True)
