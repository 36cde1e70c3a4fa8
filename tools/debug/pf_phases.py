"""Phase times of the persistent attention forward (library built with -DSEGCLIP_EXPERIMENTS, SEGCLIP_ATTN_PF_ABL=6)."""
import sys, os, math
sys.path.insert(0, os.getcwd())
os.environ["SEGCLIP_ATTN_PF_ABL"] = "6"
import torch
from segclip_amd import ops
B, T, H, hd = 256, 196, 12, 64
D = H * hd
qkv = torch.randn(B * T, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.empty(B * T, D, dtype=torch.bfloat16, device="cuda")
s3 = (T * 3 * D, 3 * D)
d = ops._attn_desc(qkv, qkv, qkv, o, B, H, T, T, hd, s3, s3, s3, (T * D, D), 1 / math.sqrt(hd), False, 0, D, 2 * D)
for _ in range(3):
    st = ops.p_attn_fwd(d, qkv)
torch.cuda.synchronize()
t = st[:256 * 7 * 16].view(256, 7, 16).double()
names = ["loop/PV tail", "wait K frags", "issue QK+LDS", "S arrives", "max/xchg/alpha", "exp/sum/pack", "rowsum xchg+rescale", "issue PV",
         "flush O(i-1)", "(setup)", "edge tile", "O->LDS", "wait vmcnt(0)", "barrier", "issue next loads", "-"]
tot = t.sum(-1).mean()
print(f"mean cycles per wave (12 items): {tot:.0f}  = {tot/12:.0f} per item")
for i, n in enumerate(names):
    print(f"  {n:22s} {t[:, :, i].mean():10.0f}  ({100 * t[:, :, i].mean() / tot:5.1f} %)   per item {t[:, :, i].mean() / 12:8.0f}   min wave {t[:, :, i].mean(0).min():9.0f} max wave {t[:, :, i].mean(0).max():9.0f}")
