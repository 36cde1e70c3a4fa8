import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from segclip_amd import ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
M, N = 50176, 2304
for K in (768, 3072):
    x = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    y = torch.empty(M, N, dtype=BF, device="cuda")
    t0 = timeit(lambda: ops.p_gemm(x, w, y, M, N, K, (K, 1), (K, 1), N))
    t1 = timeit(lambda: ops.p_gemm(x, w, y, M, N, K, (0, 1), (K, 1), N))     # every A row = row 0: always cache-resident
    t2 = timeit(lambda: ops.p_gemm(x, w, y, M, N, K, (0, 1), (0, 1), N))     # A and B rows all the same
    tiles = (M // 256) * (N // 256) / 256
    print(f"K={K}: normal {t0*1e6:.1f} us ({t0*1e6/tiles:.2f}/round)  A-rows-identical {t1*1e6:.1f} us ({t1*1e6/tiles:.2f})  A,B identical {t2*1e6:.1f} us ({t2*1e6/tiles:.2f})")
