"""Debug probe: why are weight gradients not produced in their GradSync slots on the tiny model?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import segclip_amd
from segclip_amd import synth, ops
from segclip_amd.dist import GradSync

spec = synth.SPECS[sys.argv[1] if len(sys.argv) > 1 else "tiny"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
segclip_amd.set_compute_dtype(torch.bfloat16)
model, _ = synth.build_model(spec, {}, device="cuda")
batch = synth.synthetic_batch(spec, B, seed=13, device="cuda", with_seg=False)
net = GradSync(model)
orig = ops._slot_out
log = []
def probe(slot, shape):
    out = orig(slot, shape)
    log.append((slot is not None, out is not None, tuple(shape)))
    return out
ops._slot_out = probe
for it in range(3):
    net.zero_grad(set_to_none=True)
    loss = net(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"])
    loss.backward()
    torch.cuda.synchronize()
    print(it, "stats", net.stats, "slot_out calls", len(log), "with slot", sum(a for a, _, _ in log), "got buffer", sum(b for _, b, _ in log))
    log.clear()
names = {id(p): n for n, p in model.named_parameters()}
for i, p in enumerate(net._params):
    s = net._slots.get(i)
    if s is not None and p.dim() == 2 and p.numel() > 4096:
        print(names[id(p)], tuple(p.shape), p.stride(), "grad in slot:", p.grad is not None and p.grad.data_ptr() == s.view().data_ptr())
        break
