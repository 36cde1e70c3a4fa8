import sys, os, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from segclip_amd import ops
BF = torch.bfloat16
B, T, H, hd = int(os.environ.get("B", 2)), 196, 12, 64
D = H * hd
torch.manual_seed(0)
qkv = torch.randn(B * T, 3 * D, device="cuda").to(BF)
o = torch.empty(B * T, D, dtype=BF, device="cuda")
s3 = (T * 3 * D, 3 * D)
d = ops._attn_desc(qkv, qkv, qkv, o, B, H, T, T, hd, s3, s3, s3, (T * D, D), 1 / math.sqrt(hd), False, 0, D, 2 * D)
stats = ops.p_attn_fwd(d, qkv)
do = torch.randn(B * T, D, device="cuda").to(BF)
dqkv = torch.zeros(B * T, 3 * D, dtype=BF, device="cuda")
part = torch.full((B, 3 * D), float("nan"), device="cuda")
ops.p_attn_bwd(d, stats, do, dqkv, dqkv, dqkv, s3, s3, s3, (T * D, D), 0, D, 2 * D, colsum_part=part)
qr = qkv.float().view(B, T, 3, H, hd).requires_grad_()
q, k, v = qr[:, :, 0].transpose(1, 2), qr[:, :, 1].transpose(1, 2), qr[:, :, 2].transpose(1, 2)
p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1)
ref = (p @ v).transpose(1, 2).reshape(B, T, D)
ref.backward(do.float().view(B, T, D))
got = dqkv.float().view(B, T, 3, H, hd)
err = (got - qr.grad).abs()
bad = ~(err <= 3e-2 + 3e-2 * qr.grad.abs())
print("bad total", int(bad.sum()), "nan", int(torch.isnan(got).sum()))
for part_i, nm in enumerate("qkv"):
    bb = bad[:, :, part_i]
    print(nm, "bad per (b,h):", bb.sum((1, 3)).tolist())
    print(nm, "bad per token tile:", [int(bb[:, t0:t0 + 32].sum()) for t0 in range(0, T, 32)])
cs_ref = dqkv.float().view(B, T, 3 * D).sum(1)
print("colsum maxerr", float((part - cs_ref).abs().max()), "ref max", float(cs_ref.abs().max()))
