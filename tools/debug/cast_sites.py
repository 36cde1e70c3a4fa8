"""Which ops.p_cast / p_cast_into calls does one training step make, how large are they and who asked (bf16 mode, B = 256)?"""
import os, sys, traceback, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, segclip_amd
from segclip_amd import synth, ops
segclip_amd.set_compute_dtype(torch.bfloat16)
spec = synth.SPECS["vitb16"]
model, _ = synth.build_model(spec, {}, device="cuda")
batch = synth.synthetic_batch(spec, 256, seed=1, device="cuda")
def step():
    for p in model.parameters(): p.grad = None
    loss = model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"])
    loss.backward()
for _ in range(2): step()
rec = collections.Counter()
orig = ops.p_cast
def spy(x, dt):
    fr = [f for f in traceback.extract_stack()[:-1] if "segclip_amd" in f.filename][-2:]
    rec[(tuple(x.shape), str(x.dtype).replace("torch.", ""), str(dt).replace("torch.", ""), " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr)))] += 1
    return orig(x, dt)
ops.p_cast = spy
step()
torch.cuda.synchronize()
for (shape, sd, dd, who), n in sorted(rec.items(), key=lambda kv: -kv[1] * torch.Size(kv[0][0]).numel()):
    print(f"{n:3d}x {str(shape):24s} {sd:9s}->{dd:9s} {torch.Size(shape).numel() * n / 1e6:8.1f} M elems  {who}")
