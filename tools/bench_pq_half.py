"""Half-tile tail of gemm_bf16_pq.hip (the tiles of the last, partial round of 256 workgroups run as 128 x 256 workgroups)
against the same kernel without it, in ONE process: SEGCLIP_PQ_HALF=2 makes the dispatcher consult SEGCLIP_PQ_HALF_NOW at
every call.  Outputs must be bit-identical (same products, same k order, same rounding); timing in interleaved rounds."""
import os, sys
os.environ.setdefault("SEGCLIP_TUNING", "1")   # the library honours its kernel-selection switches only with this set
os.environ["SEGCLIP_PQ_HALF"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segclip_amd import ops

BF = torch.bfloat16
dev = "cuda"


def mode(h):
    os.environ["SEGCLIP_PQ_HALF_NOW"] = str(h)


def timeit(fn, reps=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def ab(name, fn, flops, rounds=5):
    res = {}
    variants = (("full", 0), ("half", 1))
    for tag, h in variants:
        mode(h); fn(); fn()
    for r in range(rounds):
        for tag, h in variants:
            mode(h)
            res.setdefault(tag, []).append(timeit(fn))
    out = []
    for tag, _ in variants:
        v = sorted(res[tag]); med = v[len(v) // 2]
        out.append(f"{tag} {med:7.1f} us {flops / med / 1e6:6.0f} TF (min {v[0]:.1f})")
    print(f"{name:40s} " + " | ".join(out), flush=True)


def check(name, fn):
    mode(0); a = fn()
    a = torch.cat([t.float().flatten() for t in a]) if isinstance(a, tuple) else a.clone()
    mode(1); b = fn()
    b = torch.cat([t.float().flatten() for t in b]) if isinstance(b, tuple) else b.clone()
    torch.cuda.synchronize()
    ok = bool(torch.equal(a, b))
    msg = f"check {name:40s} half==full {ok}"
    if not ok:
        d = (a.float() - b.float()).abs()
        bad = (d > 0).nonzero()
        msg += f"  max|d| {float(d.max()):.3e} nbad {bad.shape[0]} first {bad[0].tolist()} last {bad[-1].tolist()}"
    print(msg, flush=True)
    return ok


def main():
    torch.manual_seed(0)
    ok = True
    # tile counts with a tail of 1..128 tiles, K from one K-tile up (the loop's end conditions)
    small = [(300 * 256 // 3, 768, 64), (300 * 256 // 3, 768, 128), (300 * 256 // 3, 768, 192), (300 * 256 // 3, 768, 256),
             (300 * 256 // 3, 768, 448), (257 * 256, 256, 320), (128 * 256 * 3, 256, 128), (50432, 768, 768)]
    for (M, N, K) in small:
        x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
        b = torch.randn(N, device=dev); dy = torch.randn(M, N, device=dev).to(BF)
        wk = (torch.randn(K, N, device=dev) * K ** -0.5).to(BF)
        r32 = torch.randn(M, N, device=dev) * 3
        r16 = r32.to(BF)
        tag = f"M{M} N{N} K{K}"
        ok &= check(f"fwd   {tag}", lambda: ops.p_linear(x, w, b)[0])
        ok &= check(f"fwd0  {tag}", lambda: ops.p_linear(x, w, None)[0])
        ok &= check(f"dgrad {tag}", lambda: ops.p_dgrad(x, wk, BF))
        ok &= check(f"res32 {tag}", lambda: ops.p_linear(x, w, b, residual=r32, out_dtype=torch.float32)[0])
        ok &= check(f"res16 {tag}", lambda: ops.p_linear(x, w, b, residual=r16)[0])
        G = ops.ACT_QUICK_GELU
        ok &= check(f"act8  {tag}", lambda: ops.p_linear(x, w, b, act=G, want_aux=True, aux_kind=2)[:2])
        u8 = ops.p_linear(x, w, b, act=G, want_aux=True, aux_kind=2)[1]
        ok &= check(f"dact8 {tag}", lambda: ops.p_dgrad(x, wk, BF, aux=u8, act=G, aux_kind=2))
        ok &= check(f"dact8+colsum {tag}", lambda: tuple(ops.p_dgrad(x, wk, BF, aux=u8, act=G, aux_kind=2, want_colsum=True)))
        del u8
        mode(1)
        y = ops.p_linear(x, w, b)[0].float(); ref = x.float() @ w.float().t() + b
        print(f"      relerr vs fp32 torch {float((y - ref).norm() / ref.norm()):.2e}", flush=True)
        del x, w, dy, wk, r32, r16, y, ref
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 50432
    for (N, K) in [(768, 768), (768, 3072), (768, 2304), (3072, 768)]:
        x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
        b = torch.randn(N, device=dev)
        wk = (torch.randn(K, N, device=dev) * K ** -0.5).to(BF)
        r32 = torch.randn(M, N, device=dev)
        fl = 2.0 * M * N * K
        ab(f"fwd bias        M{M} N{N} K{K}", lambda: ops.p_linear(x, w, b), fl)
        ab(f"fwd + f32 resid M{M} N{N} K{K}", lambda: ops.p_linear(x, w, b, residual=r32, out_dtype=torch.float32), fl)
        ab(f"dgrad           M{M} N{N} K{K}", lambda: ops.p_dgrad(x, wk, BF), fl)
        if N >= 2048:
            G = ops.ACT_QUICK_GELU
            ok &= check(f"act8  M{M} N{N} K{K}", lambda: ops.p_linear(x, w, b, act=G, want_aux=True, aux_kind=2, pitched=True)[:2])
            ab(f"c_fc fwd gelu+aux8 M{M} N{N} K{K}", lambda: ops.p_linear(x, w, b, act=G, want_aux=True, aux_kind=2, pitched=True), fl)
            u8 = ops.p_linear(x, w, b, act=G, want_aux=True, aux_kind=2, pitched=True)[1]
            ok &= check(f"dact8+colsum M{M} N{N} K{K}", lambda: tuple(ops.p_dgrad(x, wk, BF, aux=u8, act=G, aux_kind=2, want_colsum=True, pitched=True)))
            ab(f"c_proj dgrad x act8 +colsum N{N} K{K}", lambda: ops.p_dgrad(x, wk, BF, aux=u8, act=G, aux_kind=2, want_colsum=True, pitched=True), fl)
            del u8
        mode(0)
        t = min(timeit(lambda: torch.matmul(x, w.t())) for _ in range(3))
        print(f"  [torch.matmul] {t:7.1f} us {fl / t / 1e6:6.0f} TF", flush=True)
    print("ALL CHECKS OK" if ok else "MISMATCH", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
