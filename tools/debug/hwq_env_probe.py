"""Is GPU_MAX_HW_QUEUES still honoured when it is set (a) before `import torch`, (b) after the import but before the first
HIP call?   usage: hwq_env_probe.py [before|after|never]   (prints the step time with an RCCL group created first)"""
import os, sys, time
mode = sys.argv[1]
if mode == "before":
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
import torch
if mode == "after":
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.distributed as dist
import segclip_amd
from segclip_amd import synth
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29547")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
spec = synth.SPECS["vitb16"]
segclip_amd.set_compute_dtype(torch.bfloat16)
model, _ = synth.build_model(spec, {}, device="cuda")
batch = synth.synthetic_batch(spec, 256, seed=100, device="cuda", with_seg=False)
def step():
    model.zero_grad(set_to_none=True)
    model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"]).backward()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): step()
torch.cuda.synchronize()
print(f"GPU_MAX_HW_QUEUES set {mode:7s}: {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms/step")
dist.destroy_process_group()
