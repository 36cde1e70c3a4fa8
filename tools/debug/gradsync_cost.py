"""Where does the N>1 path's per-step cost come from?  ViT-B/16 B=256 bf16 on one rank (nccl group of 1):
plain | GradSync hooks+slots only | + fp32 all-reduce | + bf16 wire | torch DDP."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
import segclip_amd
from segclip_amd import synth
import segclip_amd.dist as sd

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
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
        t0 = time.perf_counter()
        loss.backward()
        return time.perf_counter() - t0
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host_bwd = 0.0
    for _ in range(steps):
        host_bwd += step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, host_bwd / steps * 1e3


print("plain            %.2f ms/step (host time inside backward() %.2f ms)" % bench(model))
for name, kw, noex in (("hooks+slots only", dict(compress=False), True), ("fp32 all-reduce ", dict(compress=False), False),
                       ("bf16 wire       ", dict(compress=True), False)):
    sd._NO_EXCHANGE = noex
    net = sd.GradSync(model, **kw)
    ms = bench(net)
    print("GradSync %s %.2f ms/step (host backward %.2f ms)  buckets %d stats %s" % (name, ms[0], ms[1], len(net._flat), net.stats))
    net.remove()
sd._NO_EXCHANGE = False
ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=True, bucket_cap_mb=64, static_graph=True)
print("torch DDP        %.2f ms/step (host backward %.2f ms)" % bench(ddp))
dist.destroy_process_group()
