"""Cost of the act'-multiplying dgrad epilogue (fc2 dgrad of the MLP) against the plain dgrad, and of the two-output
activation epilogue (fc1 forward) against the plain forward, at the vision-tower shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from segclip_amd import ops
from tools.bench_gemm import timeit
M, D, F = 50176, 768, 3072
BF = torch.bfloat16
dy = torch.randn(M, D, device="cuda").to(BF); w2 = (torch.randn(D, F, device="cuda") * F ** -0.5).to(BF)
u = torch.randn(M, F, device="cuda").to(BF)
x = torch.randn(M, D, device="cuda").to(BF); w1 = (torch.randn(F, D, device="cuda") * D ** -0.5).to(BF); b1 = torch.randn(F, device="cuda")
fl = 2.0 * M * D * F
for name, fn in (("dgrad plain            ", lambda: ops.p_dgrad(dy, w2, BF)),
                 ("dgrad * act'(u)        ", lambda: ops.p_dgrad(dy, w2, BF, aux=u, act=ops.ACT_QUICK_GELU)),
                 ("dgrad * 1 (aux loaded) ", lambda: ops.p_dgrad(dy, w2, BF, aux=u, act=ops.ACT_NONE)),
                 ("dgrad * act'(u) +colsum", lambda: ops.p_dgrad(dy, w2, BF, aux=u, act=ops.ACT_QUICK_GELU, want_colsum=True)),
                 ("fwd bias               ", lambda: ops.p_linear(x, w1, b1)),
                 ("fwd bias gelu          ", lambda: ops.p_linear(x, w1, b1, act=ops.ACT_QUICK_GELU)),
                 ("fwd bias gelu + aux    ", lambda: ops.p_linear(x, w1, b1, act=ops.ACT_QUICK_GELU, want_aux=True))):
    t = timeit(fn)
    print(f"{name} {t * 1e6:7.1f} us  {fl / t / 1e12:7.1f} TF")
