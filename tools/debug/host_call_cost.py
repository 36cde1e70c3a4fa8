"""Host cost of one launch through the Python layer: p_linear / p_ln_fwd on tiny problems (the GPU finishes each in a few
microseconds, the loop is host-bound), per call and by function (cProfile)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from segclip_amd import ops
BF = torch.bfloat16
x = torch.randn(256, 256, device="cuda").to(BF); w = torch.randn(256, 256, device="cuda").to(BF); b = torch.randn(256, device="cuda")
xf = torch.randn(256, 768, device="cuda"); g = torch.ones(768, device="cuda"); be = torch.zeros(768, device="cuda")
r32 = torch.randn(256, 256, device="cuda")


def loop(fn, n=3000):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6


for name, fn in (("p_linear bias", lambda: ops.p_linear(x, w, b)), ("p_linear + fp32 residual", lambda: ops.p_linear(x, w, b, residual=r32, out_dtype=torch.float32)),
                 ("p_dgrad", lambda: ops.p_dgrad(x, w, BF)), ("p_ln_fwd", lambda: ops.p_ln_fwd(xf, g, be, 1e-5, BF)),
                 ("torch.empty", lambda: torch.empty((256, 256), dtype=BF, device="cuda"))):
    print(f"{name:28s} {loop(fn):6.2f} us per call", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(3000):
    ops.p_linear(x, w, b)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
