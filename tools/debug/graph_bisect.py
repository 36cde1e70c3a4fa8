"""Which part of the backward breaks whole-step hipGraph capture?  CASE=linear|ln|vision|text|loss (one per process)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import segclip_amd
from segclip_amd import synth, ops
CASE = os.environ.get("CASE", "linear")
dev = torch.device("cuda", 0)
segclip_amd.set_compute_dtype(torch.bfloat16)
segclip_amd.config.overlap_towers = False
torch.manual_seed(0)
print("CASE", CASE, flush=True)
if CASE in ("linear", "ln"):
    x = torch.randn(4096, 768, device=dev, requires_grad=True)
    w = torch.randn(768, 768, device=dev, requires_grad=True)
    b = torch.randn(768, device=dev, requires_grad=True)
    params = [x, w, b]
    if CASE == "linear":
        fn = lambda: ops.linear(x.to(torch.bfloat16), w, b).float().sum()
    else:
        fn = lambda: ops.layer_norm(x, b, b, 1e-5).float().sum()
else:
    spec = synth.SPECS["vitb16"]
    model, targs = synth.build_model(spec, {}, rank=0, world_size=1, device=dev)
    batch = synth.synthetic_batch(spec, 32, seed=100, device=dev, with_seg=False)
    params = [p for p in model.parameters()]
    if CASE == "vision":
        def fn():
            ops.refresh_weight_shadows(force=True)
            return model.clip.encode_image(batch["image"].view(-1, 3, 224, 224)).float().sum()
    elif CASE == "text":
        def fn():
            ops.refresh_weight_shadows(force=True)
            return model.clip.encode_text_eot(batch["input_ids"].view(-1, batch["input_ids"].shape[-1])).float().sum()
    elif CASE == "both":
        def fn():
            ops.refresh_weight_shadows(force=True)
            return (model.clip.encode_image(batch["image"].view(-1, 3, 224, 224)).float().sum()
                    + model.clip.encode_text_eot(batch["input_ids"].view(-1, batch["input_ids"].shape[-1])).float().sum())
    elif CASE == "towers_sim":
        model.train()
        def fn():
            ops.refresh_weight_shadows(force=True)
            seq = model.clip.encode_text_eot(batch["input_ids"].view(-1, batch["input_ids"].shape[-1])).unsqueeze(1)
            vo, vh, mid = model.get_visual_output(batch["image"], shaped=False, image_frame=1, return_hidden=True)
            t, v = model._loose_similarity(seq, vo)
            return (ops.CrossEntropyFn.apply(t, 0) + ops.CrossEntropyFn.apply(v, 0)) / 2.
    elif CASE == "ce":
        lg = torch.randn(32, 32, device=dev, requires_grad=True); params = [lg]
        fn = lambda: ops.CrossEntropyFn.apply(lg * 3.0, 0)
    elif CASE == "sim":
        model.train()
        a_ = torch.randn(32, 1, 512, device=dev, requires_grad=True); b_ = torch.randn(32, 1, 512, device=dev, requires_grad=True)
        params = [a_, b_, model.clip.logit_scale]
        def fn():
            t, v = model._loose_similarity(a_, b_)
            return t.sum() + v.sum()
    elif CASE == "simce":
        model.train()
        a_ = torch.randn(32, 1, 512, device=dev, requires_grad=True); b_ = torch.randn(32, 1, 512, device=dev, requires_grad=True)
        params = [a_, b_, model.clip.logit_scale]
        def fn():
            t, v = model._loose_similarity(a_, b_)
            return (ops.CrossEntropyFn.apply(t, 0) + ops.CrossEntropyFn.apply(v, 0)) / 2.
    elif CASE == "vishid":
        model.train()
        def fn():
            ops.refresh_weight_shadows(force=True)
            vo, vh, mid = model.get_visual_output(batch["image"], shaped=False, image_frame=1, return_hidden=True)
            return vo.float().sum()
    else:
        if CASE == "contrastive":
            model.use_text_mae_recon = False
        print("flags", model.use_seglabel, model.use_text_mae_recon, model.use_vision_mae_recon, flush=True)
        fn = lambda: model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"])

def step():
    for p in params: p.grad = None
    l = fn(); l.backward(); return l

for _ in range(2): step()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
for p in params: p.grad = None
with torch.cuda.graph(g):
    l = fn(); l.backward()
torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
print("CASE", CASE, "captured and replayed OK, loss", float(l.detach()), flush=True)
