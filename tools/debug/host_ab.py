"""Host enqueue time and step time of the contrastive step at small per-GPU batches, with and without config.c_exec (same process,
alternating)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, segclip_amd
from segclip_amd import synth
segclip_amd.set_compute_dtype(torch.bfloat16)
spec = synth.SPECS["vitb16"]
for B in (int(a) for a in (sys.argv[1:] or ["64", "96", "128", "256"])):
    model, _ = synth.build_model(spec, {}, device="cuda")
    b = synth.synthetic_batch(spec, B, seed=0, device="cuda", with_seg=False)
    params = list(model.parameters())
    def step():
        for p in params: p.grad = None
        loss = model(b["input_ids"], b["segment_ids"], b["input_mask"], b["image"])
        tf = time.perf_counter()
        loss.backward()
        return tf
    out = {}
    for rep in range(2):
        for ce in (False, True):
            segclip_amd.config.c_exec = ce
            for _ in range(4): step()
            torch.cuda.synchronize()
            t0 = time.perf_counter(); fw = 0.0
            for _ in range(20):
                s0 = time.perf_counter(); tf = step(); fw += tf - s0
            t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            out[(rep, ce)] = ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3, fw / 20 * 1e3)
    segclip_amd.config.c_exec = True
    for (rep, ce), (enq, tot, fw) in sorted(out.items()):
        print(f"B={B:4d} c_exec={int(ce)} pass {rep}: forward enqueued {fw:6.2f} ms, step enqueued {enq:6.2f} ms, step {tot:6.2f} ms = {B / tot * 1e3:7.1f} pairs/s", flush=True)
    del model, b, params
    torch.cuda.empty_cache()
