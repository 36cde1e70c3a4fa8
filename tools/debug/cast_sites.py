"""Who launches the cast kernels of a step?  (monkeypatched ops.p_cast / p_cast_into: caller line + shape, one step)"""
import os, sys, collections, traceback
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import segclip_amd
from segclip_amd import synth, ops
dev = torch.device("cuda", 0)
spec = synth.SPECS["vitb16"]
segclip_amd.set_compute_dtype(torch.bfloat16)
torch.manual_seed(1234)
model, targs = synth.build_model(spec, {}, rank=0, world_size=1, device=dev)
model.clip.visual.conv1.weight.requires_grad_(False)
model.clip.visual.positional_embedding.requires_grad_(False)
batch = synth.synthetic_batch(spec, 256, seed=100, device=dev, with_seg=False)
def step():
    model.zero_grad(set_to_none=True)
    model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"]).backward()
for _ in range(2): step()
cnt = collections.Counter()
orig, orig_into = ops.p_cast, ops.p_cast_into
def site():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "cast_sites" not in f.filename:
            return f"{os.path.basename(f.filename)}:{f.lineno} {f.name}"
def p_cast(t, dtype):
    if t.dtype != dtype: cnt[(site(), tuple(t.shape), str(t.dtype).split('.')[-1] + "->" + str(dtype).split('.')[-1])] += 1
    return orig(t, dtype)
def p_cast_into(x, out):
    cnt[(site(), tuple(x.shape), "into")] += 1
    return orig_into(x, out)
ops.p_cast, ops.p_cast_into = p_cast, p_cast_into
step(); torch.cuda.synchronize()
for (s, shape, kind), c in sorted(cnt.items(), key=lambda kv: -kv[1] * torch.Size(kv[0][1]).numel()):
    print(f"{c:3d} x {kind:22s} {str(shape):24s} {s}")
