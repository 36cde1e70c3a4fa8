"""PCIe-inclusive step rate: the bench step with its inputs arriving from pinned host memory every step
(a) copied on the compute stream before the forward, (b) prefetched on a copy stream during the previous step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import segclip_amd
from segclip_amd import synth
B = 256
spec = synth.SPECS["vitb16"]
segclip_amd.set_compute_dtype(torch.bfloat16)
model, _ = synth.build_model(spec, {}, device="cuda")
model.clip.visual.conv1.weight.requires_grad_(False)
model.clip.visual.positional_embedding.requires_grad_(False)
dev_batch = synth.synthetic_batch(spec, B, seed=100, device="cuda", with_seg=False)
keys = ("input_ids", "segment_ids", "input_mask", "image")
host = {k: dev_batch[k].cpu().pin_memory() for k in keys}
nbytes = sum(host[k].numel() * host[k].element_size() for k in keys)
def fwd_bwd(b):
    model.zero_grad(set_to_none=True)
    model(b["input_ids"], b["segment_ids"], b["input_mask"], b["image"]).backward()
def timed(fn, n=8, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
t_res = timed(lambda: fwd_bwd(dev_batch))
def copy_only():
    return {k: host[k].to("cuda", non_blocking=True) for k in keys}
t_copy = timed(copy_only)
t_inline = timed(lambda: fwd_bwd(copy_only()))
copy_stream = torch.cuda.Stream()
state = {"next": None}
def prefetch():
    with torch.cuda.stream(copy_stream):
        state["next"] = copy_only()
prefetch()
def step_prefetched():
    torch.cuda.current_stream().wait_stream(copy_stream)
    cur = state["next"]
    for t in cur.values(): t.record_stream(torch.cuda.current_stream())
    prefetch()                      # next batch travels while this step computes
    fwd_bwd(cur)
t_pref = timed(step_prefetched)
print(f"inputs per step: {nbytes / 1e6:.1f} MB;  H2D copy alone {t_copy:.2f} ms ({nbytes / t_copy / 1e6:.1f} GB/s)")
print(f"inputs resident in HBM      : {t_res:.2f} ms/step  {B / t_res * 1e3:.0f} pairs/s")
print(f"H2D on the compute stream   : {t_inline:.2f} ms/step  {B / t_inline * 1e3:.0f} pairs/s")
print(f"H2D prefetched (copy stream): {t_pref:.2f} ms/step  {B / t_pref * 1e3:.0f} pairs/s")
