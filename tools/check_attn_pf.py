"""Bit-equality of the persistent forward (attention_pf.inc) with the one-workgroup-per-item forward on the same inputs:
runs itself twice (SEGCLIP_ATTN_FWD_PF=0 / 1: the switch is read once per process) and compares the saved outputs."""
import math, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [(256, 196, 12, 64, False, False), (64, 77, 8, 64, True, False), (5, 197, 8, 48, False, False),
         (7, 170, 3, 64, True, False), (9, 77, 8, 64, False, True), (3, 224, 2, 64, False, False), (300, 196, 12, 64, False, False)]


def run(path):
    import torch
    from segclip_amd import ops
    out = {}
    for i, (B, T, H, hd, causal, use_klen) in enumerate(CASES):
        D = H * hd
        g = torch.Generator(device="cuda").manual_seed(100 + i)
        qkv = torch.randn(B * T, 3 * D, device="cuda", generator=g).to(torch.bfloat16)
        o = torch.full((B * T, D), float("nan"), dtype=torch.bfloat16, device="cuda")
        s3 = (T * 3 * D, 3 * D)
        klen = (torch.arange(B, device="cuda", dtype=torch.int32) * 7 % T + 1).contiguous() if use_klen else None
        d = ops._attn_desc(qkv, qkv, qkv, o, B, H, T, T, hd, s3, s3, s3, (T * D, D), 1 / math.sqrt(hd), causal, 0, D, 2 * D, klen=klen)
        stats = ops.p_attn_fwd(d, qkv)
        torch.cuda.synchronize()
        out[f"o{i}"] = o.float().cpu(); out[f"s{i}"] = stats.cpu()
    torch.save(out, path)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1]); sys.exit(0)
    import torch
    with tempfile.TemporaryDirectory() as td:
        res = []
        for pf in ("0", "1"):
            p = os.path.join(td, f"pf{pf}.pt")
            subprocess.run([sys.executable, __file__, p], check=True, env=dict(os.environ, SEGCLIP_TUNING="1", SEGCLIP_ATTN_FWD_PF=pf))
            res.append(torch.load(p))
        bad = 0
        for k in res[0]:
            a, b = res[0][k], res[1][k]
            eq = torch.equal(a, b) or bool(((a == b) | (a.isnan() & b.isnan())).all())
            # the persistent kernel takes the row maximum over 64 keys at a time: bf16 roundings of P differ -> tolerance
            tol = 2e-2 if k.startswith("o") else 2e-5
            fin = ~(a.isinf() & b.isinf() & (a == b))
            close = bool(a.isnan().sum() == b.isnan().sum()) and bool(((a - b)[fin].abs() <= tol).all())
            if eq or close:
                print(f"ok {k} {tuple(a.shape)} max |d| {float((a - b)[fin].abs().max()) if fin.any() else 0.0:.3g}")
            elif not eq:
                bad += 1
                d = (a - b).abs()
                print(f"MISMATCH {k}: max |d| {float(d[~d.isnan()].max()) if (~d.isnan()).any() else float('nan')}, nan in new {int(b.isnan().sum())} old {int(a.isnan().sum())}, differing {int((a != b).sum())} of {a.numel()}")
        print("ALL CLOSE" if bad == 0 else f"{bad} MISMATCHES")
        sys.exit(1 if bad else 0)
