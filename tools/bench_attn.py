import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segclip_amd import ops
from tools.bench_gemm import timeit as _timeit_short


def timeit(fn, warm=60, reps=60):
    """steady state: the 3 + 10 launches of tools/bench_gemm.timeit (~3 ms of work) end before the clock governor has ramped - the same
    kernel measured 240 -> 227 -> 208 us over three back-to-back calls of that helper (tools/debug/dqw_colsum_timing.py)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


dev, BF = "cuda", torch.bfloat16
SHAPES = [(256, 196, 12, False), (256, 77, 8, True)]
for (B, T, H, causal) in SHAPES:
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
# same-node yardstick, AFTER all rows above (its allocations and clocks must not sit between them): torch's fused attention
# (scaled_dot_product_attention, bf16) on the same shapes
import torch.nn.functional as F
for (B, T, H, causal) in SHAPES:
    hd = 64
    fl = 4.0 * B * H * T * T * hd * (0.5 if causal else 1.0)
    q, k, v = (torch.randn(B, H, T, hd, device=dev, dtype=BF, requires_grad=True) for _ in range(3))
    go = torch.randn(B, H, T, hd, device=dev, dtype=BF)
    try:
        ts = timeit(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=causal))
        oo = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
        tsb = timeit(lambda: torch.autograd.grad(oo, (q, k, v), go, retain_graph=True))
        print(f"[torch SDPA] B{B} T{T} H{H} causal={causal}: fwd {ts*1e6:7.1f} us ({fl/ts/1e12:6.1f} TF/s)  bwd {tsb*1e6:7.1f} us ({2.5*fl/tsb/1e12:6.1f} TF/s)")
    except Exception as e:
        print(f"[torch SDPA] not available here: {type(e).__name__}: {str(e)[:120]}")
