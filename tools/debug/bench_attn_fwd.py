import sys, os, math
os.environ.setdefault("SEGCLIP_TUNING", "1")   # the library honours its kernel-selection switches only with this set
sys.path.insert(0, os.getcwd())
import torch
from segclip_amd import ops
from tools.bench_gemm import timeit
B, T, H, hd = 256, 196, 12, 64
D = H * hd
qkv = torch.randn(B * T, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.empty(B * T, D, dtype=torch.bfloat16, device="cuda")
s3 = (T * 3 * D, 3 * D)
def desc():
    return ops._attn_desc(qkv, qkv, qkv, o, B, H, T, T, hd, s3, s3, s3, (T * D, D), 1 / math.sqrt(hd), False, 0, D, 2 * D)
ops.p_attn_fwd(desc(), qkv)
t = timeit(lambda: ops.p_attn_fwd(desc(), qkv))
print(f"PF={os.environ.get('SEGCLIP_ATTN_FWD_PF')} ABL={os.environ.get('SEGCLIP_ATTN_PF_ABL')} GRID={os.environ.get('SEGCLIP_ATTN_FWD_GRID')}: fwd {t*1e6:7.1f} us")
