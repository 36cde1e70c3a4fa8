"""Row pitch of the GEMM operands vs speed: is the K loop's memory path losing to L2/HBM channel camping?
(rows of 768 bf16 = 1536 B = 12 cache lines: a 256-row K-slice touches every 12th line)"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from segclip_amd import ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
M = 50176
for N, K in ((2304, 768), (3072, 768), (768, 3072), (768, 768)):
    for pa, pb in ((0, 0), (64, 0), (0, 64), (64, 64), (192, 192), (32, 32)):
        x = torch.randn(M, K + pa, device="cuda").to(BF)
        w = (torch.randn(N, K + pb, device="cuda") * K ** -0.5).to(BF)
        y = torch.empty(M, N, dtype=BF, device="cuda")
        t0 = timeit(lambda: ops.p_gemm(x, w, y, M, N, K, (K + pa, 1), (K + pb, 1), N))
        ref = x[:512, :K].float() @ w[:, :K].float().t()
        err = float((y[:512].float() - ref).norm() / ref.norm())
        print(f"N={N} K={K} pitch A +{pa} B +{pb}: {t0*1e6:8.1f} us  {2.0*M*N*K/t0/1e12:7.1f} TF/s  err {err:.1e}", flush=True)
