"""Does an initialised RCCL process group (world size 1) change the step time of the PLAIN model, and does it matter
whether the group exists before the model's first step?   usage: pg_cost.py [first|after] [eager|lazy]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
import segclip_amd
from segclip_amd import synth, ops, streams

order = sys.argv[1] if len(sys.argv) > 1 else "after"
eager = (sys.argv[2] if len(sys.argv) > 2 else "eager") == "eager"


def init_pg():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29542")
    kw = dict(device_id=torch.device("cuda", 0)) if eager else {}
    dist.init_process_group("nccl", rank=0, world_size=1, **kw)
    t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()


if order == "first":
    init_pg()
spec = synth.SPECS["vitb16"]
segclip_amd.set_compute_dtype(torch.bfloat16)
model, _ = synth.build_model(spec, {}, device="cuda")
model.clip.visual.conv1.weight.requires_grad_(False)
model.clip.visual.positional_embedding.requires_grad_(False)
batch = synth.synthetic_batch(spec, 256, seed=100, device="cuda", with_seg=False)


def bench(net, steps=8, warm=3):
    def step():
        net.zero_grad(set_to_none=True)
        loss = net(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"])
        loss.backward()
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def both(tag):
    a = bench(model)
    segclip_amd.config.overlap_towers = False
    b = bench(model)
    segclip_amd.config.overlap_towers = True
    segclip_amd.config.overlap_wgrad = False
    c = bench(model)
    segclip_amd.config.overlap_wgrad = True
    print("%-34s %.2f ms   towers serialised %.2f   wgrad on main stream %.2f" % (tag, a, b, c), flush=True)


print(f"order={order} eager={eager} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}")
both("group exists" if order == "first" else "no process group")
print("  side streams:", streams.stats["picked"])
if order != "first":
    init_pg()
    both("group of 1 created after")
dist.destroy_process_group()
both("group destroyed")
