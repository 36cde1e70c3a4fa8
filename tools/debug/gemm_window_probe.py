"""Run with SEGCLIP_P8_EPI_ABL=0/2/3/4/5: operand rows from a cache-resident window (results wrong for != 0)."""
import sys, os
os.environ.setdefault("SEGCLIP_TUNING", "1")   # the library honours its kernel-selection switches only with this set
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from segclip_amd import ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
M = 50176
out = []
for N, K in ((2304, 768), (3072, 768), (768, 3072), (768, 768)):
    x = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    y = torch.empty(M, N, dtype=BF, device="cuda")
    t0 = timeit(lambda: ops.p_gemm(x, w, y, M, N, K, (K, 1), (K, 1), N))
    out.append(f"N={N} K={K}: {t0*1e6:7.1f} us")
print(f"abl={os.environ.get('SEGCLIP_P8_EPI_ABL','0')}  " + "   ".join(out))
