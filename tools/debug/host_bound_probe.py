"""Is a configuration's step bound by the host?  Per step: time until the host has enqueued everything (step() returns)
against the time until the GPU has finished (synchronize)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, segclip_amd
from segclip_amd import synth
segclip_amd.set_compute_dtype(torch.bfloat16)
spec = synth.SPECS["vitb16"]
for name, flags, B in (("contrastive B=256", {}, 256), ("full loss B=256", dict(use_seglabel=True, use_vision_mae_recon=True), 256),
                       ("contrastive B=64", {}, 64), ("contrastive B=128", {}, 128)):
    model, _ = synth.build_model(spec, flags, device="cuda")
    b = synth.synthetic_batch(spec, B, seed=0, device="cuda", with_seg=bool(flags))
    params = list(model.parameters())
    def step():
        for p in params: p.grad = None
        loss = model(b["input_ids"], b["segment_ids"], b["input_mask"], b["image"], image_seg=b.get("image_seg"))
        t_f = time.perf_counter()
        loss.backward()
        return t_f
    for _ in range(4): step()
    torch.cuda.synchronize()
    enq, tot, fwd = [], [], []
    for _ in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); tf = step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        enq.append(t1 - t0); tot.append(t2 - t0); fwd.append(tf - t0)
    m = lambda v: sorted(v)[len(v) // 2] * 1e3
    print(f"{name:20s}: forward enqueued {m(fwd):6.2f} ms, step enqueued {m(enq):6.2f} ms, GPU done {m(tot):6.2f} ms (single synchronised steps)", flush=True)
    del model, b, params
    torch.cuda.empty_cache()
