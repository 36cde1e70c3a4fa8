"""attention forward (bf16) against the fp32 expression at several sequence lengths; run with SEGCLIP_ATTN_FWD_LEAN=0/1"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from segclip_amd import ops
BF = torch.bfloat16
for (B, T, H, causal) in [(2, 196, 12, False), (2, 256, 4, False), (2, 288, 4, False), (2, 576, 16, False), (2, 577, 16, False), (3, 77, 8, True), (2, 320, 4, True)]:
    hd = 64; D = H * hd
    torch.manual_seed(T)
    qkv = (torch.randn(B * T, 3 * D, device="cuda") * 1.5).to(BF)
    o = torch.empty(B * T, D, dtype=BF, device="cuda")
    s3 = (T * 3 * D, 3 * D)
    d = ops._attn_desc(qkv, qkv, qkv, o, B, H, T, T, hd, s3, s3, s3, (T * D, D), 1 / math.sqrt(hd), causal, 0, D, 2 * D)
    stats = ops.p_attn_fwd(d, qkv)
    q, k, v = (qkv.float().view(B, T, 3, H, hd)[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if causal:
        s = s + torch.full((T, T), float("-inf"), device="cuda").triu_(1)
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * T, D)
    lse = torch.logsumexp(s, -1).reshape(-1)
    err = (o.float() - ref).abs()
    print(f"T={T} causal={causal}: max err {float(err.max()):.4e} rel rms {float(err.norm() / ref.norm()):.3e}  lse max err {float((stats[:lse.numel()] - lse).abs().max()):.3e}", flush=True)
