"""A few launches of the attention forward + backward kernels at one shape (for rocprofv3 --pmc runs).
usage: python tools/one_attn.py [B T H causal]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segclip_amd import ops
B, T, H, causal = (int(x) for x in (sys.argv[1:5] + ["256", "197", "12", "0"][len(sys.argv) - 1:]))
dev, BF = "cuda", torch.bfloat16
hd, D = 64, H * 64
qkv = torch.randn(B * T, 3 * D, device=dev).to(BF)
o = torch.empty(B * T, D, dtype=BF, device=dev)
do = torch.randn(B * T, D, device=dev).to(BF)
dqkv = torch.empty_like(qkv)
s3 = (T * 3 * D, 3 * D)
desc = lambda: ops._attn_desc(qkv, qkv, qkv, o, B, H, T, T, hd, s3, s3, s3, (T * D, D), 1 / math.sqrt(hd), bool(causal), 0, D, 2 * D)
for _ in range(3):
    stats = ops.p_attn_fwd(desc(), qkv)
    ops.p_attn_bwd(desc(), stats, do, dqkv, dqkv, dqkv, s3, s3, s3, (T * D, D), 0, D, 2 * D)
torch.cuda.synchronize()
