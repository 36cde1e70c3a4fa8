"""How many parameter gradients does config.fold_param_grads fold per step of the full-loss configuration, and how many
aten adds remain?  (torch profiler kernel counts, fold off / on)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, segclip_amd
from segclip_amd import synth, ops
from torch.profiler import profile, ProfilerActivity
segclip_amd.set_compute_dtype(torch.bfloat16)
spec = synth.SPECS["vitb16"]
model, _ = synth.build_model(spec, dict(use_seglabel=True, use_vision_mae_recon=True), device="cuda")
b = synth.synthetic_batch(spec, 256, seed=0, device="cuda", with_seg=True)
params = list(model.parameters())
folded = []
real = ops._GradFold.flush
ops._GradFold.flush = staticmethod(lambda adds: (folded.append(len(adds)), real(adds))[1])
def step():
    for p in params: p.grad = None
    loss = model(b["input_ids"], b["segment_ids"], b["input_mask"], b["image"], image_seg=b.get("image_seg"))
    loss.backward()
for fold in (False, True):
    segclip_amd.config.fold_param_grads = fold
    for _ in range(2): step()
    torch.cuda.synchronize()
    folded.clear()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step(); torch.cuda.synchronize()
    adds = sum(e.count for e in prof.key_averages() if "CUDAFunctor_add" in e.key)
    tot = sum(e.count for e in prof.key_averages())
    print(f"fold={fold}: folded per step {folded}, aten add launches {adds}, all launches {tot}")
    cnt = {e.key[:90]: e.count for e in prof.key_averages()}
    if fold:
        for k in sorted(set(cnt) | set(prev), key=lambda k: -abs(cnt.get(k, 0) - prev.get(k, 0)))[:12]:
            print(f"fold   delta {cnt.get(k, 0) - prev.get(k, 0):+5d}  ({prev.get(k, 0)} -> {cnt.get(k, 0)})  {k}")
    prev = cnt
