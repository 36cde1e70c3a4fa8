"""How much of a step is host-side enqueue time?  (python tools/host_time.py [--ddp])"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import segclip_amd
from segclip_amd import synth

ddp = "--ddp" in sys.argv
segclip_amd.set_compute_dtype(torch.bfloat16)
spec = synth.SPECS["vitb16"]
model, _ = synth.build_model(spec, {}, device="cuda")
model.clip.visual.conv1.weight.requires_grad_(False)
model.clip.visual.positional_embedding.requires_grad_(False)
net = model
if ddp:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=("--no-unused" not in sys.argv),
                                                    gradient_as_bucket_view=("--no-view" not in sys.argv),
                                                    bucket_cap_mb=int(os.environ.get("BUCKET_MB", "64")),
                                                    static_graph=("--static" in sys.argv))
b = synth.synthetic_batch(spec, 256, seed=0, device="cuda", with_seg=False)


def step():
    net.zero_grad(set_to_none=("--keep-grads" not in sys.argv))
    loss = net(b["input_ids"], b["segment_ids"], b["input_mask"], b["image"])
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"ddp={ddp} host enqueue {1e3 * (t1 - t0) / 10:.2f} ms/step, wall {1e3 * (t2 - t0) / 10:.2f} ms/step")
