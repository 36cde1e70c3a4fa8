import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from segclip_amd import ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
for (M, N, K) in ((512, 512, 128), (1024, 1536, 384), (50432, 1536, 384)):
    x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF); b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, dtype=BF, device="cuda"); aux = torch.empty(M, N, dtype=torch.uint8, device="cuda"); auxb = torch.empty(M, N, dtype=BF, device="cuda")
    for act in (ops.ACT_QUICK_GELU, ops.ACT_GELU_ERF):
        for kind, a in ((2, aux), (0, auxb)):
            try:
                f = lambda: ops.p_gemm(x, w, y, M, N, K, (K, 1), (K, 1), N, bias=b, aux=a, ldaux=N, act=act, aux_kind=kind)
                f(); t = timeit(f)
                print(f"M{M} N{N} K{K} act {act} aux_kind {kind}: {t * 1e6:7.1f} us", flush=True)
            except Exception as e:
                print(f"M{M} N{N} K{K} act {act} aux_kind {kind}: {type(e).__name__} {str(e)[-90:]}", flush=True)
