"""Would two half-batch pipelines on two streams beat one full-batch pipeline?  (tail rounds of the N = 768 GEMMs: 588 tiles =
2.3 rounds of 256 CUs.)  Times NB residual blocks forward + backward on the vision shape, full batch on one stream against two
halves on two streams (each half with its own weight-gradient launches), no autograd."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import segclip_amd
from segclip_amd import ops, streams
BF = torch.bfloat16
B, T, D, H, NB = 256, 196, 768, 12, 6
dev = "cuda"
torch.manual_seed(0)
def params():
    P = [torch.ones(D), torch.zeros(D), torch.randn(3 * D, D) * D ** -0.5, torch.zeros(3 * D), torch.randn(D, D) * D ** -0.5 * 0.5, torch.zeros(D),
         torch.ones(D), torch.zeros(D), torch.randn(4 * D, D) * D ** -0.5, torch.zeros(4 * D), torch.randn(D, 4 * D) * (4 * D) ** -0.5 * 0.5, torch.zeros(D)]
    return [torch.nn.Parameter(p.to(dev)) for p in P]
blocks = [params() for _ in range(NB)]
x = torch.randn(B * T, D, device=dev)
g = torch.randn(B * T, D, device=dev).to(BF)
cfg = lambda b: (b, T, D, H, False, ops.ACT_QUICK_GELU, BF)
none4, none8 = (None,) * 4, (None,) * 8
need = (True,) * 13

def pipeline(xh, gh, b):
    cur, saved = xh, []
    for P in blocks:
        cur, sv = ops._resblock_fwd(cur, P, b, T, H, False, ops.ACT_QUICK_GELU, 1e-5, BF, None)
        saved.append(sv)
    g16 = gh
    for sv in reversed(saved):
        _, g16, _ = ops._resblock_bwd(sv, cfg(b), None, none4, none8, need, None, g16, True)
    return g16

def full():
    pipeline(x, g, B)

main = torch.cuda.current_stream()
s2 = streams.side_stream("half2")
hb = B // 2
xa, xb, ga, gb = x[:hb * T], x[hb * T:], g[:hb * T], g[hb * T:]
def split():
    s2.wait_stream(main)
    with torch.cuda.stream(s2):
        pipeline(xb, gb, hb)
    pipeline(xa, ga, hb)
    main.wait_stream(s2)

def timeit(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

for r in range(3):
    print(f"round {r}: full batch, one stream {timeit(full):7.2f} ms | two halves, two streams {timeit(split):7.2f} ms   ({NB} blocks fwd+bwd)", flush=True)
def serial_halves():
    pipeline(xa, ga, hb); pipeline(xb, gb, hb)
print(f"two halves, ONE stream {timeit(serial_halves):7.2f} ms")
