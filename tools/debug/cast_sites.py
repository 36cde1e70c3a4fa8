"""Who calls the cast kernel in one default bench step (caller, element count, dtypes)."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import segclip_amd
from segclip_amd import ops, synth
segclip_amd.set_compute_dtype(torch.bfloat16)
spec = synth.SPECS["vitb16"]
model, _ = synth.build_model(spec, {}, device="cuda")
model.clip.visual.conv1.weight.requires_grad_(False); model.clip.visual.positional_embedding.requires_grad_(False)
batch = synth.synthetic_batch(spec, 256, seed=1, device="cuda", with_seg=False)
def step():
    model.zero_grad(set_to_none=True)
    loss = model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"]); loss.backward()
for _ in range(2): step()
agg = collections.Counter()
def spy(name):
    orig = getattr(ops, name)
    def w(*a, **k):
        fr = [f for f in traceback.extract_stack()[:-1] if "segclip_amd" in f.filename]
        who = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:][::-1])
        x = a[0]
        agg[(name, tuple(x.shape), str(x.dtype)[6:], who)] += 1
        return orig(*a, **k)
    setattr(ops, name, w)
for n in ("p_cast", "p_cast_into"):
    spy(n)
step(); torch.cuda.synchronize()
for k, v in sorted(agg.items(), key=lambda kv: -kv[1] * (torch.tensor(kv[0][1]).prod().item() if kv[0][1] else 1)):
    print(v, k)
