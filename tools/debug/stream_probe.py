"""Which of torch's pool streams run beside the current stream?  (HIP hardware-queue assignment, see segclip_amd/streams.py)
usage: python tools/debug/stream_probe.py [pg]   - `pg` initialises an RCCL group of 1 first"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from segclip_amd import streams
if "pg" in sys.argv:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda"); dist.all_reduce(t)
main = torch.cuda.current_stream()
c = [torch.cuda.Stream() for _ in range(12)]
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), " pg" if "pg" in sys.argv else "")
print("overlap with main :", "".join("Y" if streams.overlaps(main, s) else "." for s in c))
for i in range(6):
    print(f"overlap with c[{i}] :", "".join(("Y" if streams.overlaps(c[i], s) else ".") if s is not c[i] else "-" for s in c))
for role in ("text", "wgrad", "comm"):
    streams.side_stream(role)
print(streams.stats)
