"""attention_dqw.inc against an fp32 torch reference on a few small cases: error of dQ | dK | dV per 32-row tile and of the token sums."""
import math, os, sys
os.environ.setdefault("SEGCLIP_TUNING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from segclip_amd import ops
for (B, T, H) in [(1, 196, 1), (2, 196, 3), (40, 200, 12)]:
    hd = 64; D = H * hd
    g = torch.Generator(device="cuda").manual_seed(7)
    qkv = torch.randn(B * T, 3 * D, device="cuda", generator=g).to(torch.bfloat16)
    do = torch.randn(B * T, D, device="cuda", generator=g).to(torch.bfloat16)
    o = torch.empty(B * T, D, dtype=torch.bfloat16, device="cuda")
    s3 = (T * 3 * D, 3 * D)
    desc = lambda: ops._attn_desc(qkv, qkv, qkv, o, B, H, T, T, hd, s3, s3, s3, (T * D, D), 1 / math.sqrt(hd), False, 0, D, 2 * D)
    stats = ops.p_attn_fwd(desc(), qkv)
    dqkv = torch.full_like(qkv, float("nan"))
    cs = torch.full((B, 3 * D), float("nan"), dtype=torch.float32, device="cuda")
    ops.p_attn_bwd(desc(), stats, do, dqkv, dqkv, dqkv, s3, s3, s3, (T * D, D), 0, D, 2 * D, colsum_part=cs)
    torch.cuda.synchronize()
    q, k, v = (qkv[:, j * D:(j + 1) * D].float().view(B, T, H, hd).transpose(1, 2).requires_grad_() for j in range(3))
    p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1)
    oo = (p @ v).transpose(1, 2).reshape(B * T, D)
    gq, gk, gv = torch.autograd.grad(oo, (q, k, v), do.float())
    ref = torch.cat([x.transpose(1, 2).reshape(B * T, D) for x in (gq, gk, gv)], 1)
    out = dqkv.float()
    print(f"== B{B} T{T} H{H}: nan {int(out.isnan().sum())}")
    for name, j in (("dQ", 0), ("dK", 1), ("dV", 2)):
        e = (out[:, j * D:(j + 1) * D] - ref[:, j * D:(j + 1) * D]).abs().view(B, T, D)
        per_tile = [float(e[:, t0:t0 + 32].max()) for t0 in range(0, T, 32)]
        per_item = e.view(B, T, H, hd).amax(dim=(1, 3))
        print(f"  {name}: max {float(e.max()):.4f} (ref max {float(ref[:, j*D:(j+1)*D].abs().max()):.3f}); per q/k tile: " + " ".join(f"{x:.3f}" for x in per_tile))
        if B * H <= 36: print("     per item:", " ".join(f"{float(x):.3f}" for x in per_item.flatten()))
    cref = ref.view(B, T, 3 * D).sum(1)
    ce = (cs - cref).abs()
    for name, j in (("sum dQ", 0), ("sum dK", 1), ("sum dV", 2)):
        print(f"  {name}: max err {float(ce[:, j*D:(j+1)*D].max()):.4f} (ref max {float(cref[:, j*D:(j+1)*D].abs().max()):.3f})")
