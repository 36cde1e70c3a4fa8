import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segclip_amd import ops
from tools.bench_gemm import timeit
dev, BF = "cuda", torch.bfloat16
M, N = 50176, int(sys.argv[1]) if len(sys.argv) > 1 else 2304
for K in (64, 128, 256, 512, 768, 1536, 3072):
    x = torch.randn(M, K, device=dev).to(BF)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
    t = timeit(lambda: ops.p_linear(x, w, None))
    tiles = (M // 256) * (N // 256)
    print(f"K={K:5d} {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF/s  per-tile-round {t*1e6/(tiles/256):6.2f} us")
