"""attention_dqw.inc at B = 256, T = 196 under SEGCLIP_ATTN_ABL (experiments library of tools/build_exp_attn.sh; results garbage for abl != 0):
0 the kernel, 1 no memory instructions inside the step loop (compute only), 2 no compute (memory stream only), 11 / 12 / 13 s_setprio 1 for the dQ wave / key-owner waves 4-6 / all key-owners."""
import sys, os, math, subprocess
os.environ["SEGCLIP_TUNING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    os.environ["SEGCLIP_ATTN_ABL"] = sys.argv[1]
    from segclip_amd import _lib
    _lib._LIB_PATH = os.path.join(os.path.dirname(_lib._LIB_PATH), "libsegclip_hip_exp.so")
    import torch
    from segclip_amd import ops
    from tools.bench_gemm import timeit
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
    t = timeit(lambda: ops.p_attn_bwd(desc(), stats, do, dqkv, dqkv, dqkv, s3, s3, s3, (T * D, D), 0, D, 2 * D, colsum_part=cs))
    print(f"SEGCLIP_ATTN_ABL={sys.argv[1]}: {t * 1e6:.1f} us", flush=True)
else:
    for abl in sys.argv[2:] or ("0", "1", "2", "11", "12", "13", "0"):
        subprocess.run([sys.executable, __file__, abl], check=True)
