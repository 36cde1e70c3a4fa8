"""Which parameters report to GradSync more or less than once per pass with ResStackFn's early publication?
(single process, gloo group of 1, tiny spec, full loss, f32)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
import segclip_amd
from segclip_amd import synth
from segclip_amd.dist import GradSync
from tests.helpers import FULL_FLAGS, noise_items
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29551")
dist.init_process_group("gloo", rank=0, world_size=1)
spec = synth.SPECS["tiny"]
segclip_amd.set_compute_dtype(torch.float32)
model, _ = synth.build_model(spec, FULL_FLAGS, device="cuda")
batch = synth.synthetic_batch(spec, 4, seed=3, device="cuda")
noise = synth.synthetic_noise(spec, 4, seed=3, device="cuda")
def run(net):
    net.zero_grad(set_to_none=True)
    with segclip_amd.noise_injection(noise_items(noise, FULL_FLAGS)):
        loss = net(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"], image_seg=batch["image_seg"])
    loss.backward(); torch.cuda.synchronize()
run(model)
local = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
names = {id(p): n for n, p in model.named_parameters()}
calls = collections.Counter()
orig = GradSync._on_grad
def wrapped(self, p):
    calls[names[id(p)]] += 1
    return orig(self, p)
GradSync._on_grad = wrapped
net = GradSync(model)
for it in range(3):
    calls.clear(); run(net)
    bad = {n: c for n, c in calls.items() if c != 1}
    worst = max(((float((p.grad - local[n]).abs().max()), n) for n, p in model.named_parameters() if n in local), default=(0, ""))
    missing = [n for n, p in model.named_parameters() if p.requires_grad and n in local and n not in calls]
    print(f"pass {it}: reports {sum(calls.values())} params {len(calls)} missing {missing[:5]} not-once {dict(list(bad.items())[:8])}  worst grad diff {worst}  pending {net._pending}")
