"""Phase times (s_memtime) of the loader-wave attention backward; experiments build, SEGCLIP_ATTN_ABL=9."""
import sys, os, math
os.environ.setdefault("SEGCLIP_TUNING", "1")   # the library honours its kernel-selection switches only with this set
sys.path.insert(0, os.getcwd())
os.environ["SEGCLIP_ATTN_ABL"] = "9"
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
t = dbg.view(torch.float32)[:256 * 8 * 8].view(256, 8, 8)[:, :7, :6].double()
names = ["wait at S0 (loader: Q/dO landed, staging read)", "S0->S1: K/V fragments, zero acc", "main loop", "loop end -> S2 (skew)",
         "S2->S3: dq acc read, park dK/dV (+ wait loader fill)", "S3->end: dQ flush, token sums"]
tot = t.sum(-1).mean()
print(f"mean cycles per compute wave (12 items): {tot:.0f} = {tot / 12:.0f} per item")
for i, nme in enumerate(names):
    print(f"  {nme:52s} {t[:, :, i].mean() / 12:9.0f} per item ({100 * t[:, :, i].mean() / tot:5.1f} %)   min wave {t[:, :, i].mean(0).min() / 12:8.0f} max wave {t[:, :, i].mean(0).max() / 12:8.0f}")

lt = dbg.view(torch.float32)[65536:65536 + 256 * 128].view(256, 128)[:, 112:122].double()
ln = ["flush dK/dV (ds_read + store)", "wait vmcnt(0): Q DMA + stores", "barrier S0", "token sums + barrier S1", "prefetch: issue + reduce",
      "wait: K DMA", "barrier S2 (waits for compute)", "Q DMA issue + dO ds_write", "wait parked counter", "-"]
print(f"loader wave: {lt.sum(-1).mean() / 12:.0f} cycles per item")
for i, nme in enumerate(ln):
    print(f"  {nme:40s} {lt[:, i].mean() / 12:9.0f} per item")
