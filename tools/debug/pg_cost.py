"""Does an initialised RCCL process group (world size 1) change the step time of the PLAIN model?  Times the step
before and after init_process_group, with the embedding all-gather on and off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
import segclip_amd
from segclip_amd import synth, ops

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


print("no process group            %.2f ms" % bench(model))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29542")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
print("group of 1, gather on       %.2f ms" % bench(model))
real = ops.AllGatherFn.apply
ops.AllGatherFn.apply = staticmethod(lambda x: x)
print("group of 1, gather bypassed %.2f ms" % bench(model))
ops.AllGatherFn.apply = real
print("group of 1, gather on       %.2f ms" % bench(model))
segclip_amd.config.overlap_towers = False
print("  towers serialised         %.2f ms" % bench(model))
dist.destroy_process_group()
print("group destroyed             %.2f ms" % bench(model))
