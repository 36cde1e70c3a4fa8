"""Multi-GPU evidence that does not need the node: when does each GradSync bucket become ready inside the
backward pass, and how much of its all-reduce would stay exposed at 8 GPUs?

Runs the benchmarked step through the N > 1 code path with a 1-rank RCCL group (`bench.py --force-dist` conditions: GradSync
buckets, hooks, communication stream; the all-reduce itself is empty at world size 1, so the timeline is undisturbed), records
a device event when every gradient of a bucket exists (GradSync.timeline) and prints, per bucket: bytes and ready time
relative to the END of the backward pass, plus a serial-channel model of the exchange at W = 8:
    ring   : 2 (W-1)/W bytes / 153 GB/s   (one ring = one xGMI link per hop)
    direct : 2 (bytes/W) / 153 GB/s       (reduce-scatter + all-gather over all 7 links at once)
for the fp32 wire (what the reference's DDP exchanges) and the bf16 wire, for the default weight-gradient grouping and for
groups capped at 3 blocks (config.wgrad_group_blocks_dist).
usage: python tools/bucket_timeline.py [out.txt]"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

import segclip_amd
from segclip_amd import synth
from segclip_amd.dist import GradSync

LINK = 153e9
W = 8


def run(cap, lines):
    segclip_amd.config.wgrad_group_blocks_dist = cap
    spec = synth.SPECS["vitb16"]
    model, _ = synth.build_model(spec, {}, device="cuda")
    model.clip.visual.conv1.weight.requires_grad_(False)
    model.clip.visual.positional_embedding.requires_grad_(False)
    net = GradSync(model)
    batch = synth.synthetic_batch(spec, 256, seed=100, device="cuda", with_seg=False)
    params = list(model.parameters())

    def step(record=False):
        for p in params:
            p.grad = None
        loss = net(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"])
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        if record:
            net.timeline = []
        e0.record()
        loss.backward()
        e1.record()
        return e0, e1

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    best = None
    for _ in range(5):          # the pass with the shortest backward (least host jitter)
        e0, e1 = step(record=True)
        torch.cuda.synchronize()
        tl = [(b, nb, e0.elapsed_time(ev)) for b, nb, ev in net.timeline]
        tot = e0.elapsed_time(e1)
        if best is None or tot < best[0]:
            best = (tot, tl)
    net.timeline = None
    tot, tl = best
    lines.append(f"== weight-gradient groups: {'package default (vision 7 + 3 blocks, text 5 + 5 + 1 + 1)' if cap >= 12 else 'capped at %d blocks' % cap}; "
                 f"backward pass {tot:.2f} ms on the device, {len(tl)} buckets, {sum(nb for _, nb, _ in tl) / 1e6:.1f} MB of fp32 gradients")
    lines.append(f"{'bucket':>6s} {'MB':>8s} {'ready at (ms)':>14s} {'before end (ms)':>16s}")
    for b, nb, t in tl:
        lines.append(f"{b:6d} {nb / 1e6:8.1f} {t:14.2f} {tot - t:16.2f}")
    for wire, scale in (("fp32", 1.0), ("bf16", 0.5)):
        for algo, f in (("ring", lambda n: 2 * (W - 1) / W * n / LINK), ("direct", lambda n: 2 * (n / W) / LINK)):
            busy = 0.0
            for b, nb, t in tl:
                start = max(t, busy)
                busy = start + f(nb * scale) * 1e3
            total = sum(f(nb * scale) for _, nb, _ in tl) * 1e3
            lines.append(f"   W = {W}, {wire} wire, {algo:6s}: {total:6.2f} ms of transfers, last one ends {busy - tot:+6.2f} ms relative to the end of "
                         f"backward -> exposed {max(busy - tot, 0.0):5.2f} ms")
    del net, model
    torch.cuda.empty_cache()


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    segclip_amd.set_compute_dtype(torch.bfloat16)
    lines = ["# tools/bucket_timeline.py: GradSync bucket-ready timeline of the benchmarked step (ViT-B/16, B = 256, bf16, contrastive",
             "# loss) through the N > 1 code path with a 1-rank RCCL group, and a serial-channel model of the W = 8 exchange over xGMI",
             "# (153 GB/s per link; MI355X_MICROARCH.md).  'ready' = a device event on the communication stream behind its waits on",
             "# every stream that produced a gradient of the bucket.  No multi-GPU hardware was involved."]
    for cap in (12, 3):
        run(cap, lines)
    dist.destroy_process_group()
    txt = "\n".join(lines) + "\n"
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt)


if __name__ == "__main__":
    main()
