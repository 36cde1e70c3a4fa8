"""Where the streaming backward differs from an fp32 reference: section (dQ|dK|dV), sample, token, head."""
import sys, os, math
os.environ.setdefault("SEGCLIP_TUNING", "1")   # the library honours its kernel-selection switches only with this set
sys.path.insert(0, os.getcwd())
os.environ["SEGCLIP_ATTN_BWD_SQ"] = "1"
import torch
from segclip_amd import ops
B, T, H, hd = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (2, 161, 3, 64)))
D = H * hd
g = torch.Generator(device="cuda").manual_seed(104)
qkv = torch.randn(B * T, 3 * D, device="cuda", generator=g).to(torch.bfloat16)
do = torch.randn(B * T, D, device="cuda", generator=g).to(torch.bfloat16)
o = torch.empty(B * T, D, dtype=torch.bfloat16, device="cuda")
s3 = (T * 3 * D, 3 * D)
desc = lambda: ops._attn_desc(qkv, qkv, qkv, o, B, H, T, T, hd, s3, s3, s3, (T * D, D), 1 / math.sqrt(hd), False, 0, D, 2 * D)
stats = ops.p_attn_fwd(desc(), qkv)
dqkv = torch.full_like(qkv, float("nan"))
ops.p_attn_bwd(desc(), stats, do, dqkv, dqkv, dqkv, s3, s3, s3, (T * D, D), 0, D, 2 * D)
torch.cuda.synchronize()
q, k, v = (qkv[:, j * D:(j + 1) * D].float().view(B, T, H, hd).transpose(1, 2).requires_grad_() for j in range(3))
p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1)
oo = (p @ v).transpose(1, 2).reshape(B * T, D)
gq, gk, gv = torch.autograd.grad(oo, (q, k, v), do.float())
ref = torch.cat([x.transpose(1, 2).reshape(B * T, D) for x in (gq, gk, gv)], 1)
err = (dqkv.float() - ref).abs().view(B, T, 3, H, hd)
bad = (err > 0.03)
print("bad elements", int(bad.sum()), "of", bad.numel())
for sec in range(3):
    e = bad[:, :, sec]
    if not e.any(): continue
    idx = e.nonzero()
    print("section", "QKV"[sec], "count", len(idx))
    print("  samples", sorted(set(idx[:, 0].tolist())), "heads", sorted(set(idx[:, 2].tolist())))
    toks = sorted(set(idx[:, 1].tolist())); print("  tokens", toks[:40], "..." if len(toks) > 40 else "")
    cols = sorted(set(idx[:, 3].tolist())); print("  cols", cols[:70])
