"""Same-box A/B of attention-backward builds: python tools/debug/attn_ab.py <lib.so> [<lib.so> ...] [--rounds N]
Each library (a full libsegclip_hip.so variant, see tools/relink_attn.sh) is loaded in its own process; prints the T=196 (ViT-B) and
T=576 (ViT-L) backward times, alternating the libraries."""
import sys, os, math, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def one(lib):
    from segclip_amd import _lib
    _lib._LIB_PATH = os.path.abspath(lib)
    import torch
    from segclip_amd import ops
    from tools.bench_gemm import timeit
    out = []
    for (B, T, H, hd) in ((256, 196, 12, 64), (128, 576, 16, 64)):
        D = H * hd
        qkv = torch.randn(B * T, 3 * D, device="cuda").to(torch.bfloat16)
        do = torch.randn(B * T, D, device="cuda").to(torch.bfloat16)
        o = torch.empty(B * T, D, dtype=torch.bfloat16, device="cuda")
        s3 = (T * 3 * D, 3 * D)
        desc = lambda: ops._attn_desc(qkv, qkv, qkv, o, B, H, T, T, hd, s3, s3, s3, (T * D, D), 1 / math.sqrt(hd), False, 0, D, 2 * D)
        stats = ops.p_attn_fwd(desc(), qkv)
        dqkv = torch.empty_like(qkv)
        cs = torch.empty((B, 3 * D), dtype=torch.float32, device="cuda")
        t = timeit(lambda: ops.p_attn_bwd(desc(), stats, do, dqkv, dqkv, dqkv, s3, s3, s3, (T * D, D), 0, D, 2 * D, colsum_part=cs))
        out.append(f"T{T} {t * 1e6:7.1f} us")
    print(f"{os.path.basename(lib):40s} " + "   ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--one":
        one(sys.argv[2]); sys.exit(0)
    args = sys.argv[1:]
    rounds = 2
    if "--rounds" in args:
        i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
    for _ in range(rounds):
        for lib in args:
            subprocess.run([sys.executable, __file__, "--one", lib], check=True, stderr=subprocess.DEVNULL)
