"""Phase times (s_memtime) of the streaming attention backward (attention_sq.inc); experiments build,
SEGCLIP_ATTN_ABL=9 SEGCLIP_ATTN_BWD_SQ=1."""
import sys, os, math
os.environ.setdefault("SEGCLIP_TUNING", "1")   # the library honours its kernel-selection switches only with this set
sys.path.insert(0, os.getcwd())
os.environ["SEGCLIP_ATTN_ABL"] = "9"
os.environ["SEGCLIP_ATTN_BWD_SQ"] = "1"
import torch
from segclip_amd import ops
B, T, H, hd = 256, 196, 12, 64
D = H * hd
qkv = torch.randn(B * T, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.empty(B * T, D, dtype=torch.bfloat16, device="cuda")
do = torch.randn(B * T, D, device="cuda").to(torch.bfloat16)
dqkv = torch.zeros(B * T, 3 * D, dtype=torch.bfloat16, device="cuda")
s3 = (T * 3 * D, 3 * D)
desc = lambda: ops._attn_desc(qkv, qkv, qkv, o, B, H, T, T, hd, s3, s3, s3, (T * D, D), 1 / math.sqrt(hd), False, 0, D, 2 * D)
stats = ops.p_attn_fwd(desc(), qkv)
dbg = torch.zeros(B * T * D, dtype=torch.bfloat16, device="cuda")     # stands in for dQ (contiguous): the stamps land at its head
for _ in range(3):
    ops.p_attn_bwd(desc(), stats, do, dbg.view(B * T, D), dqkv, dqkv, (T * D, D), s3, s3, (T * D, D), 0, D, 2 * D)
torch.cuda.synchronize()
allw = dbg.view(torch.float32)[:256 * 8 * 8].view(256, 8, 8).double()
t = allw[:, :7, :]
names = ["wait K tile", "wait query tile (ready)", "S/dP products + softmax", "wait scratch buffer (dqd of tile g-3)", "park dS^T + dV/dK products",
         "owner: wait for all dS^T (cnt)", "owner: dQ product + park", "end of item: dK/dV flush, token sums"]
tot = t.sum(-1).mean()
print(f"mean cycles per compute wave (12 items): {tot:.0f} = {tot / 12:.0f} per item")
for i, nme in enumerate(names):
    print(f"  {nme:44s} {t[:, :, i].mean() / 12:9.0f} per item ({100 * t[:, :, i].mean() / tot:5.1f} %)   min wave {t[:, :, i].mean(0).min() / 12:8.0f} max wave {t[:, :, i].mean(0).max() / 12:8.0f}")
lt = allw[:, 7, :]
ln = ["issue a tile (+K, dQ store)", "wait vmcnt", "finish a tile (D, dO -> LDS, publish)", "idle (slot not free)", "drain"]
print(f"loader wave: {lt.sum(-1).mean() / 12:.0f} cycles per item")
for i, nme in enumerate(ln):
    print(f"  {nme:44s} {lt[:, i].mean() / 12:9.0f} per item")
