"""Phase times (s_memtime) of the streaming attention backward with a dQ wave (attention_dqw.inc).  Needs the experiments
library of tools/build_exp_attn.sh (SEGCLIP_ATTN_ABL=9; the stamps overwrite the head of dQ)."""
import sys, os, math
os.environ["SEGCLIP_TUNING"] = "1"
os.environ["SEGCLIP_ATTN_ABL"] = "9"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segclip_amd import _lib
_lib._LIB_PATH = os.path.join(os.path.dirname(_lib._LIB_PATH), "libsegclip_hip_exp.so")
import torch
from segclip_amd import ops
B, T, H, hd = 256, 196, 12, 64
D = H * hd
qkv = torch.randn(B * T, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.empty(B * T, D, dtype=torch.bfloat16, device="cuda")
do = torch.randn(B * T, D, device="cuda").to(torch.bfloat16)
dqkv = torch.zeros(B * T, 3 * D, dtype=torch.bfloat16, device="cuda")
cs = torch.zeros(B, 3 * D, dtype=torch.float32, device="cuda")
s3 = (T * 3 * D, 3 * D)
desc = lambda: ops._attn_desc(qkv, qkv, qkv, o, B, H, T, T, hd, s3, s3, s3, (T * D, D), 1 / math.sqrt(hd), False, 0, D, 2 * D)
stats = ops.p_attn_fwd(desc(), qkv)
dbg = torch.zeros(B * T * D, dtype=torch.bfloat16, device="cuda")     # stands in for dQ (contiguous): the stamps land at its head
for _ in range(3):
    ops.p_attn_bwd(desc(), stats, do, dbg.view(B * T, D), dqkv, dqkv, (T * D, D), s3, s3, (T * D, D), 0, D, 2 * D, colsum_part=cs)
torch.cuda.synchronize()
t = dbg.view(torch.float32)[:256 * 8 * 8].view(256, 8, 8).double()
items, steps = 12, 12 * 7
kn = ["barrier", "park dK + memory issue (pos 0)", "K/V frags + S/dP next tile + D share (+ mem pos 1)", "softmax + park dS (+ mem pos 2)", "dV/dK products (issue) (+ mem pos 3)", "end of step (lgkmcnt, vmcnt)", "end of item (next K / V fragments, park dV)", "barrier K + park dK (step 0 of an item)"]
k = t[:, :7, :8]
tot = k.sum(-1).mean()
print(f"key-owner waves: {tot:.0f} cycles = {tot / items:.0f} per item = {tot / steps:.0f} per step")
for i, n in enumerate(kn):
    print(f"  {n:52s} {k[:, :, i].mean() / steps:8.0f} per step ({100 * k[:, :, i].mean() / tot:5.1f} %)  by wave: " + " ".join(f"{x / steps:6.0f}" for x in k[:, :, i].mean(0)))
ln = ["barrier", "dQ product", "tile out (pack, patch, 4 stores)", "token sums + K^T reload"]
l = t[:, 7, :4]
ltot = l.sum(-1).mean()
print(f"dQ wave: {ltot:.0f} cycles = {ltot / steps:.0f} per step")
for i, n in enumerate(ln):
    print(f"  {n:52s} {l[:, i].mean() / steps:8.0f} per step ({100 * l[:, i].mean() / ltot:5.1f} %)")
m = dbg.view(torch.float32)[65536:65536 + 256 * 8 * 4].view(256, 8, 4)[:, :7, :3].double()
for i, n in enumerate(["tile pieces (2 per step; wave 6: 1)", "parked-tile stores (8 per item)", "own K / V pieces (8 per item)"]):
    print(f"  memory issue: {n:40s} {m[:, :, i].mean() / steps:8.0f} per step  by wave: " + " ".join(f"{x / steps:6.0f}" for x in m[:, :, i].mean(0)))
