"""Output row pitch vs speed of the store-heavy GEMMs (is the epilogue's store burst camping on HBM channels?)"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from segclip_amd import ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
M = 50176
for N, K in ((3072, 768), (2304, 768), (768, 768), (768, 3072)):
    x = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    for pad in (0, 64, 128, 256, 512):
        y = torch.empty(M, N + pad, dtype=BF, device="cuda")
        t0 = timeit(lambda: ops.p_gemm(x, w, y, M, N, K, (K, 1), (K, 1), N + pad))
        print(f"N={N} K={K} ldc=N+{pad}: {t0*1e6:8.1f} us  {2.0*M*N*K/t0/1e12:7.1f} TF/s", flush=True)
