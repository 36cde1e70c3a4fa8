import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segclip_amd import ops
from tools.bench_gemm import timeit
dev, BF = "cuda", torch.bfloat16
for (B, T, H, causal) in [(256, 196, 12, False), (256, 77, 8, True)]:
    hd, D = 64, H * 64
    qkv = torch.randn(B * T, 3 * D, device=dev).to(BF)
    o = torch.empty(B * T, D, dtype=BF, device=dev)
    do = torch.randn(B * T, D, device=dev).to(BF)
    dqkv = torch.empty_like(qkv)
    s3 = (T * 3 * D, 3 * D)
    def desc():
        return ops._attn_desc(qkv, qkv, qkv, o, B, H, T, T, hd, s3, s3, s3, (T * D, D), 1 / math.sqrt(hd), causal, 0, D, 2 * D)
    stats = ops.p_attn_fwd(desc(), qkv)
    tf = timeit(lambda: ops.p_attn_fwd(desc(), qkv))
    tb = timeit(lambda: ops.p_attn_bwd(desc(), stats, do, dqkv, dqkv, dqkv, s3, s3, s3, (T * D, D), 0, D, 2 * D))
    fl = 4.0 * B * H * T * T * hd * (0.5 if causal else 1.0)
    print(f"B{B} T{T} H{H} causal={causal}: fwd {tf*1e6:7.1f} us ({fl/tf/1e12:6.1f} TF/s)  bwd {tb*1e6:7.1f} us ({2.5*fl/tb/1e12:6.1f} TF/s)")
