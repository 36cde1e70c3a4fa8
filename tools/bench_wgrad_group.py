"""Grouped weight gradients in isolation (ops.WgradGroup / segclip_wgrad_group): time of one grouped launch (+ its combine)
for n consecutive ViT-B/16 or text blocks, next to the library's time model and to one launch per gradient.
usage: python tools/bench_wgrad_group.py [vision|text]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from segclip_amd import ops, _lib as L

which = sys.argv[1] if len(sys.argv) > 1 else "vision"
R, D, F4 = (50176, 768, 3072) if which == "vision" else (19712, 512, 2048)
lib = L.load()
dev = "cuda"
g = torch.Generator(device="cpu").manual_seed(0)
def mk(r, c):
    return (torch.randn(r, c, generator=g) * 0.5).to(dev).to(torch.bfloat16)
# one block's operands (dy, x): c_proj, c_fc, out_proj, in_proj
one = [(mk(R, D), mk(R, F4)), (mk(R, F4), mk(R, D)), (mk(R, D), mk(R, D)), (mk(R, 3 * D), mk(R, D))]
tiles_blk = (4 * D * D + 2 * F4 * D) // 65536


def timed(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def single():
    for dy, x in one:
        ops.p_wgrad(dy, x)
t1 = timed(single)
flops_blk = sum(2.0 * R * dy.shape[1] * x.shape[1] for dy, x in one)
print(f"# {which}: R = {R}, D = {D}, {tiles_blk} output tiles per block, {R // 64} K steps; one launch per gradient (+ its split-K combine): "
      f"{t1:.0f} us per block = {flops_blk / t1 / 1e6:.0f} TF/s")
print("blocks  K ranges  items  rounds   measured us  per block   TF/s   model us per block")
for n in (1, 2, 3, 4, 5, 7, 10, 12):
    s = lib.segclip_wgrad_group_splits(tiles_blk * n, R // 64)
    model = lib.segclip_wgrad_group_model_us(tiles_blk * n, R // 64, s)

    def grouped():
        wg = ops.WgradGroup()
        for _ in range(n):
            for dy, x in one:
                wg.add(dy, x)
        wg.flush()
    t = timed(grouped, 4)
    items = tiles_blk * n * s
    print(f"{n:6d}  {s:8d}  {items:5d}  {items / 256:6.2f}  {t:12.0f}  {t / n:9.0f}  {flops_blk * n / t / 1e6:5.0f}   {model / n:8.0f}")
