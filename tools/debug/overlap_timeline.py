"""Host / GPU timeline of the two towers per step (why does the text-tower overlap come and go?).
usage: overlap_timeline.py [first|none]   - `first` creates an RCCL group of 1 before the model"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
import segclip_amd
from segclip_amd import synth

if len(sys.argv) > 1 and sys.argv[1] == "first":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29546")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
spec = synth.SPECS["vitb16"]
segclip_amd.set_compute_dtype(torch.bfloat16)
model, _ = synth.build_model(spec, {}, device="cuda")
model.clip.visual.conv1.weight.requires_grad_(False)
model.clip.visual.positional_embedding.requires_grad_(False)
batch = synth.synthetic_batch(spec, 256, seed=100, device="cuda", with_seg=False)
EV = lambda: torch.cuda.Event(enable_timing=True)
rec = []
cur = {}
orig_text, orig_vis = model.clip.encode_text_eot, model.get_visual_output


def text(*a, **k):
    e0, e1 = EV(), EV(); h0 = time.perf_counter()
    e0.record(); out = orig_text(*a, **k); e1.record()
    cur.update(text=(e0, e1), text_host=(h0, time.perf_counter()))
    return out


def vis(*a, **k):
    e0, e1 = EV(), EV(); h0 = time.perf_counter()
    e0.record(); out = orig_vis(*a, **k); e1.record()
    cur.update(vis=(e0, e1), vis_host=(h0, time.perf_counter()))
    return out


model.clip.encode_text_eot = text
model.get_visual_output = vis


def step():
    cur.clear()
    s0 = EV(); s0.record(); h0 = time.perf_counter()
    model.zero_grad(set_to_none=True)
    loss = model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"])
    h1 = time.perf_counter(); f1 = EV(); f1.record()
    loss.backward()
    h2 = time.perf_counter(); s1 = EV(); s1.record()
    rec.append(dict(cur, s0=s0, f1=f1, s1=s1, host=(h0, h1, h2), allocs=torch.cuda.memory_stats()["num_device_alloc"]))


def run(n, tag):
    rec.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / n * 1e3
    print(f"--- {tag}: {wall:.2f} ms/step")
    base = rec[0]["s0"]
    for i, r in enumerate(rec):
        g = lambda e: base.elapsed_time(e)
        h = [(x - t0) * 1e3 for x in r["host"]]
        print(f" step {i}: host fwd {h[0]:7.1f}->{h[1]:7.1f} bwd->{h[2]:7.1f} | gpu step {g(r['s0']):7.1f}->{g(r['s1']):7.1f} fwd_end {g(r['f1']):7.1f}"
              f" | text fwd gpu {g(r['text'][0]):7.1f}->{g(r['text'][1]):7.1f} (host {(r['text_host'][0]-t0)*1e3:6.1f}->{(r['text_host'][1]-t0)*1e3:6.1f})"
              f" | vis fwd gpu {g(r['vis'][0]):7.1f}->{g(r['vis'][1]):7.1f} (host ->{(r['vis_host'][1]-t0)*1e3:6.1f}) allocs {r['allocs']}")


for _ in range(3):
    step()
run(6, "overlap on")
segclip_amd.config.overlap_towers = False
for _ in range(2):
    step()
run(4, "towers serialised")
segclip_amd.config.overlap_towers = True
for _ in range(2):
    step()
run(6, "overlap on again")
