"""Two gloo ranks on one GPU: per-parameter error of GradSync's averaged gradients vs all_reduce(local)/2 with ResStackFn."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist, torch.multiprocessing as mp

def worker(rank, world, port, fuse):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import segclip_amd
    from segclip_amd import synth
    from segclip_amd.dist import GradSync
    from tests.helpers import FULL_FLAGS, noise_items
    segclip_amd.config.fuse_res_stack = fuse
    spec = synth.SPECS["tiny"]; B = 4
    segclip_amd.set_compute_dtype(torch.float32)
    model, _ = synth.build_model(spec, FULL_FLAGS, rank=rank, world_size=world, device="cuda")
    gb = synth.synthetic_batch(spec, B * world, seed=5, device="cuda"); gn = synth.synthetic_noise(spec, B * world, seed=5, device="cuda")
    sl = slice(rank * B, (rank + 1) * B)
    batch = {k: v[sl] for k, v in gb.items()}; noise = {k: v[sl] for k, v in gn.items()}
    def run(net):
        net.zero_grad(set_to_none=True)
        with segclip_amd.noise_injection(noise_items(noise, FULL_FLAGS)):
            loss = net(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"], image_seg=batch["image_seg"])
        loss.backward(); torch.cuda.synchronize()
    run(model)
    local = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    net = GradSync(model)
    for it in range(3):
        run(net)
        bad = []
        for n, p in model.named_parameters():
            if n not in local: continue
            e = local[n].clone(); dist.all_reduce(e); e /= world
            err = float((p.grad - e).abs().max()) / (float(e.abs().max()) + 1e-12)
            errl = float((p.grad - local[n]).abs().max()) / (float(e.abs().max()) + 1e-12)
            if err > 1e-4: bad.append((n, round(err, 4), "==local" if errl < 1e-5 else round(errl, 4), net._slots[net._index[id(p)]].bucket if net._steady else -1))
        if rank == 0:
            print(f"fuse={fuse} pass {it}: {len(bad)} wrong of {len(local)}; buckets {len(net._flat)}; first {bad[:6]}", flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    for fuse in (False, True):
        mp.spawn(worker, args=(2, 29560 + int(fuse), fuse), nprocs=2, join=True)
