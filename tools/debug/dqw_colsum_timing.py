"""Why tools/bench_attn.py (no token-sum buffer) and tools/check_attn_dqw.py (with one) time the same kernel differently."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from segclip_amd import ops
from tools.bench_gemm import timeit
B, T, H, hd = 256, 196, 12, 64
D = H * hd
qkv = torch.randn(B * T, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.empty(B * T, D, dtype=torch.bfloat16, device="cuda")
do = torch.randn(B * T, D, device="cuda").to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
cs = torch.empty((B, 3 * D), dtype=torch.float32, device="cuda")
s3 = (T * 3 * D, 3 * D)
desc = lambda: ops._attn_desc(qkv, qkv, qkv, o, B, H, T, T, hd, s3, s3, s3, (T * D, D), 1 / math.sqrt(hd), False, 0, D, 2 * D)
stats = ops.p_attn_fwd(desc(), qkv)
for rep in range(3):
    for name, kw in (("no colsum", {}), ("colsum", {"colsum_part": cs})):
        t = timeit(lambda: ops.p_attn_bwd(desc(), stats, do, dqkv, dqkv, dqkv, s3, s3, s3, (T * D, D), 0, D, 2 * D, **kw))
        print(f"{name:10s} {t * 1e6:7.1f} us", flush=True)
