"""The streaming attention backward (attention_dqw.inc, SEGCLIP_ATTN_BWD_DQW=1) against the loader-wave kernel it replaces
(SEGCLIP_ATTN_BWD_DQW=0) and against an fp32 torch reference, same inputs: runs itself twice (the switch is read once per
process) and compares dQ | dK | dV and the per-sample token sums.  Also prints the kernel time of each variant."""
import math, os, subprocess, sys, tempfile
os.environ.setdefault("SEGCLIP_TUNING", "1")   # the library honours its kernel-selection switches only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# B, T, H, hd
# `python tools/check_attn_dqw.py long`: the key-chunked variant for sequences of more than 256 tokens (SEGCLIP_ATTN_BWD_DQW_LONG)
# against the two streaming launches it replaces
LONG = os.environ.get("SEGCLIP_CHECK_LONG") == "1" or (len(sys.argv) > 1 and sys.argv[1] == "long")
SWITCH = "SEGCLIP_ATTN_BWD_DQW_LONG" if LONG else "SEGCLIP_ATTN_BWD_DQW"
CASES_LONG = [(128, 576, 16, 64), (2, 576, 3, 64), (3, 288, 2, 64), (2, 320, 1, 64), (128, 577, 16, 64), (2, 577, 3, 64), (1, 257, 1, 64), (3, 449, 2, 64), (2, 478, 5, 64), (1, 1021, 2, 64), (40, 290, 7, 64), (600, 321, 1, 64)]
CASES = CASES_LONG if LONG else [(256, 196, 12, 64), (5, 197, 8, 64), (3, 222, 2, 64), (300, 196, 12, 64), (2, 193, 3, 64), (700, 200, 5, 64), (2, 196, 3, 64), (1, 196, 1, 64), (23, 211, 12, 64)]


def run(path):
    import torch
    from segclip_amd import ops
    from tools.bench_gemm import timeit
    out = {}
    for i, (B, T, H, hd) in enumerate(CASES):
        D = H * hd
        g = torch.Generator(device="cuda").manual_seed(100 + i)
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
        out[f"d{i}"] = dqkv.float().cpu(); out[f"c{i}"] = cs.cpu()
        if i == 0:
            t = timeit(lambda: ops.p_attn_bwd(desc(), stats, do, dqkv, dqkv, dqkv, s3, s3, s3, (T * D, D), 0, D, 2 * D, colsum_part=cs))
            print(f"{SWITCH}={os.environ.get(SWITCH, '0')}: B{B} T{T} H{H} bwd {t * 1e6:.1f} us", flush=True)
        if B * T <= 4096:   # fp32 reference
            q, k, v = (qkv[:, j * D:(j + 1) * D].float().view(B, T, H, hd).transpose(1, 2).requires_grad_() for j in range(3))
            p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1)
            oo = (p @ v).transpose(1, 2).reshape(B * T, D)
            gq, gk, gv = torch.autograd.grad(oo, (q, k, v), do.float())
            out[f"r{i}"] = torch.cat([x.transpose(1, 2).reshape(B * T, D) for x in (gq, gk, gv)], 1).cpu()
    torch.save(out, path)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] != "long":
        run(sys.argv[1]); sys.exit(0)
    import torch
    with tempfile.TemporaryDirectory() as td:
        res = []
        for sq in ("0", "1"):
            p = os.path.join(td, f"dqw{sq}.pt")
            subprocess.run([sys.executable, __file__, p], check=True, env=dict(os.environ, SEGCLIP_TUNING="1", SEGCLIP_CHECK_LONG="1" if LONG else "0", **{SWITCH: sq}))
            res.append(torch.load(p))
        bad = 0
        for k in sorted(res[0]):
            a, b = res[0][k], res[1][k]
            if k.startswith("r"):
                continue
            nan_b = int(b.isnan().sum())
            d = (a - b).abs()
            # both are bf16 results of the same arithmetic in a different summation order (dQ: one product over all keys
            # instead of seven partial ones summed in fp32; token sums: cs[key] in two bf16 parts either way)
            tol = (2e-2 * max(1.0, float(a.abs().max())) if k.startswith("d") else 1e-2 * max(1.0, float(a.abs().max())))
            ok = nan_b == 0 and float(d.max()) <= tol
            line = f"{'ok' if ok else 'MISMATCH'} {k} {tuple(a.shape)} max |d| {float(d[~d.isnan()].max()):.3g} (tol {tol:.3g}) nan {nan_b}"
            r = res[0].get("r" + k[1:]) if k.startswith("d") else None
            if r is not None:
                ea, eb = float((a - r).abs().max()), float((b - r).abs().max())
                ma, mb = float((a - r).abs().mean()), float((b - r).abs().mean())
                line += f" | vs fp32: old max {ea:.3g} mean {ma:.3g}, new max {eb:.3g} mean {mb:.3g}"
                if not (mb <= 1.25 * ma + 1e-6):
                    ok = False; line = "WORSE-THAN-OLD " + line
            print(line)
            bad += 0 if ok else 1
        print("ALL CLOSE" if bad == 0 else f"{bad} MISMATCHES")
        sys.exit(1 if bad else 0)
