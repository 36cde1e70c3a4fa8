"""K sweep of the persistent GEMM (intercept = per-tile fixed cost, slope = time per K-tile) and output-path ablations."""
import os, sys
os.environ.setdefault("SEGCLIP_TUNING", "1")   # the library honours its kernel-selection switches only with this set
os.environ["SEGCLIP_GEMM_PQ"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from segclip_amd import ops
from tools.bench_pq import mode, timeit

BF = torch.bfloat16
M = 50176
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2304
rounds = -(-(M // 256) * (N // 256) // 256)
print(f"M {M} N {N}: {(M // 256) * (N // 256)} tiles = {(M // 256) * (N // 256) / 256:.2f} rounds")
for K in (64, 128, 256, 512, 768, 1536, 3072):
    x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF); b = torch.randn(N, device="cuda")
    row = [f"K {K:5d}"]
    for tag, pq, ps, abl in (("p8", 0, 0, 0), ("pq", 1, 0, 0), ("pq-nostore", 1, 0, 1), ("pq-noepi", 1, 0, 2)):
        mode(pq, ps); os.environ["SEGCLIP_PQ_ABL"] = str(abl)
        ops.p_linear(x, w, b); ops.p_linear(x, w, b)
        ts = sorted(timeit(lambda: ops.p_linear(x, w, b)) for _ in range(3))
        row.append(f"{tag} {ts[1]:7.1f}")
    os.environ["SEGCLIP_PQ_ABL"] = "0"
    print(" | ".join(row), flush=True)
