"""the e4m3 attention forward (experiments library, tools/build_exp_attn.sh) against the bf16 forward at the ViT-L/14@336
shape (B = 128, T = 576, H = 16), steady-state timing."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segclip_amd import _lib
_lib._LIB_PATH = os.path.join(os.path.dirname(_lib._LIB_PATH), "libsegclip_hip_exp.so")
import torch
from segclip_amd import ops


def timeit(fn, warm=40, reps=40):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for (B, T, H) in ((128, 576, 16), (256, 196, 12)):
    hd, D = 64, H * 64
    qkv = torch.randn(B * T, 3 * D, device="cuda").to(torch.bfloat16)
    o = torch.empty(B * T, D, dtype=torch.bfloat16, device="cuda")
    s3 = (T * 3 * D, 3 * D)
    fl = 4.0 * B * H * T * T * hd
    for fp8 in (False, True, False, True):
        desc = lambda: ops._attn_desc(qkv, qkv, qkv, o, B, H, T, T, hd, s3, s3, s3, (T * D, D), 1 / math.sqrt(hd), False, 0, D, 2 * D, fp8=fp8)
        t = timeit(lambda: ops.p_attn_fwd(desc(), qkv))
        print(f"B{B} T{T} H{H} {'e4m3' if fp8 else 'bf16'} forward {t * 1e6:8.1f} us ({fl / t / 1e12:6.1f} TF/s)", flush=True)
