"""gemm_bf16_pq.hip (accumulators in the accumulator file, transposed result blocks, packed-bf16 output path) against the
8-phase kernel gemm_bf16_p8.hip in ONE process: bit-exact comparison of the outputs (same products, same k order, same
rounding) and interleaved timing rounds (SEGCLIP_GEMM_PQ=2 makes the dispatcher consult SEGCLIP_GEMM_PQ_NOW at every call)."""
import os, sys
os.environ.setdefault("SEGCLIP_TUNING", "1")   # the library honours its kernel-selection switches only with this set
os.environ["SEGCLIP_GEMM_PQ"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segclip_amd import ops

BF = torch.bfloat16
dev = "cuda"


def mode(pq, persist=0):
    os.environ["SEGCLIP_GEMM_PQ_NOW"] = str(pq)


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
    """fn() under p8 / pq, interleaved rounds; prints median us and TF/s"""
    res = {}
    variants = (("p8", 0), ("pq", 1))
    for tag, pq in variants:
        mode(pq); fn(); fn()
    for r in range(rounds):
        for tag, pq in variants:
            mode(pq)
            res.setdefault(tag, []).append(timeit(fn))
    out = []
    for tag, _ in variants:
        v = sorted(res[tag]); med = v[len(v) // 2]
        out.append(f"{tag} {med:7.1f} us {flops / med / 1e6:6.0f} TF (min {v[0]:.1f})")
    print(f"{name:36s} " + " | ".join(out), flush=True)


def same(a, b):
    if isinstance(a, (tuple, list)):
        return all(same(x, y) for x, y in zip(a, b) if x is not None)
    if a.dtype == torch.uint8:   # the one-byte derivative: rint() ties may fall either way (a few elements per million, 1 step)
        d = (a.int() - b.int()).abs()
        return bool(d.max() <= 1) and int((d > 0).sum()) <= max(4, a.numel() // 100000)
    return bool(torch.equal(a, b))


def check(name, fn, ref_fn=None, exact=True):
    """fn() -> tensor or tuple of tensors; compared bit-exactly between the two kernels"""
    mode(0); a = fn(); a = tuple(t.clone() for t in a) if isinstance(a, tuple) else a.clone()
    mode(1); b = fn(); b = tuple(t.clone() for t in b) if isinstance(b, tuple) else b.clone()
    torch.cuda.synchronize()
    ok = same(a, b)
    msg = f"check {name:34s} pq==p8 {ok}"
    b0, a0 = (b[0], a[0]) if isinstance(b, tuple) else (b, a)
    if ref_fn is not None:
        r = ref_fn()
        msg += f"  relerr vs fp32 torch {float((b0.float() - r).norm() / r.norm()):.2e}"
    if not ok:
        for i, (x, y) in enumerate(zip(a if isinstance(a, tuple) else (a,), b if isinstance(b, tuple) else (b,))):
            if x is None:
                continue
            d = (x.float() - y.float()).abs()
            msg += f"  [out {i}: max|d| {float(d.max()):.3e} nbad {(d > 0).sum().item()} rel {float(d.norm() / x.float().norm()):.2e}]"
    print(msg, flush=True)
    return ok


def main():
    torch.manual_seed(0)
    ok = True
    for (M, N, K) in [(256, 256, 64), (512, 768, 128), (2048, 512, 192), (4096, 2304, 768), (19712, 512, 512)]:
        x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
        b = torch.randn(N, device=dev); dy = torch.randn(M, N, device=dev).to(BF)
        ok &= check(f"fwd  M{M} N{N} K{K}", lambda: ops.p_linear(x, w, b)[0], lambda: x.float() @ w.float().t() + b)
        ok &= check(f"fwd0 M{M} N{N} K{K}", lambda: ops.p_linear(x, w, None)[0])
        ok &= check(f"dgrd M{M} N{N} K{K}", lambda: ops.p_dgrad(dy, w, BF), lambda: dy.float() @ w.float())
        if M % 64 == 0:   # weight gradient dW (N, K) = dy^T x: both operands k-strided, fp32 out, split-K slabs
            ok &= check(f"wgrd M{M} N{N} K{K}", lambda: ops.p_wgrad(dy, x), lambda: dy.float().t() @ x.float())
        r32 = torch.randn(M, N, device=dev) * 3
        ok &= check(f"res32 M{M} N{N} K{K}", lambda: ops.p_linear(x, w, b, residual=r32, out_dtype=torch.float32)[0],
                    lambda: x.float() @ w.float().t() + b + r32)
        # + bf16 residual (bf16 residual stream): the pq kernel rounds the product before the add, so not bit-equal to p8
        r16 = (torch.randn(M, N, device=dev) * 3).to(BF)
        ref = x.float() @ w.float().t() + b + r16.float()
        mode(0); y0 = ops.p_linear(x, w, b, residual=r16)[0].float(); mode(1); y1 = ops.p_linear(x, w, b, residual=r16)[0].float()
        e0, e1 = float((y0 - ref).norm() / ref.norm()), float((y1 - ref).norm() / ref.norm())
        print(f"check res  M{M} N{N} K{K}: relerr vs fp32 torch p8 {e0:.2e} pq {e1:.2e}  max|pq-p8| {float((y0 - y1).abs().max()):.3e}", flush=True)
        ok &= e1 < 1.3 * e0 + 1e-4
        # QuickGELU + one-byte saved derivative; data gradient times the saved derivative, with fused column sums
        ok &= check(f"act8 M{M} N{N} K{K}", lambda: ops.p_linear(x, w, b, act=ops.ACT_QUICK_GELU, want_aux=True, aux_kind=2))
        u8 = ops.p_linear(x, w, b, act=ops.ACT_QUICK_GELU, want_aux=True, aux_kind=2)[1]
        wk = (torch.randn(N, K, device=dev) * N ** -0.5).to(BF)     # (N_in, K_out): dgrad of a Linear K -> N ... used as dy(M,K?)
        dyk = torch.randn(M, K, device=dev).to(BF)
        # dx (M, N) = dyk (M, K) @ wk2 (K, N) * act'(u8 (M, N))
        wk2 = (torch.randn(K, N, device=dev) * K ** -0.5).to(BF)
        ok &= check(f"dact M{M} N{N} K{K}", lambda: ops.p_dgrad(dyk, wk2, BF, aux=u8, act=ops.ACT_QUICK_GELU, aux_kind=2),
                    lambda: (dyk.float() @ wk2.float()) * (u8.float() / 204.0 - 0.125))
        def dact_cs():
            dx, cs = ops.p_dgrad(dyk, wk2, BF, aux=u8, act=ops.ACT_QUICK_GELU, aux_kind=2, want_colsum=True)
            return dx, cs
        mode(0); dx0, cs0 = dact_cs(); mode(1); dx1, cs1 = dact_cs(); torch.cuda.synchronize()
        ref = dx1.float().sum(0)
        e0 = float((cs0 - ref).norm() / ref.norm()); e1 = float((cs1 - ref).norm() / ref.norm())
        good = bool(torch.equal(dx0, dx1)) and e1 < 2e-3
        print(f"check dact+colsum M{M} N{N} K{K}: dx equal {bool(torch.equal(dx0, dx1))}  colsum relerr vs sum(dx) p8 {e0:.2e} pq {e1:.2e}", flush=True)
        ok &= good
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 50176
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)] if M > 20000 else [(1536, 512), (512, 512), (2048, 512), (512, 2048)]
    for (N, K) in shapes:
        x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
        b = torch.randn(N, device=dev); dy = torch.randn(M, N, device=dev).to(BF)
        ok &= check(f"fwd  M{M} N{N} K{K}", lambda: ops.p_linear(x, w, b)[0])
        ok &= check(f"dgrd M{M} N{N} K{K}", lambda: ops.p_dgrad(dy, w, BF))
        fl = 2.0 * M * N * K
        ab(f"fwd   M{M} N{N} K{K}", lambda: ops.p_linear(x, w, b), fl)
        ab(f"dgrad M{M} N{K} K{N}", lambda: ops.p_dgrad(dy, w, BF), fl)
        ok &= check(f"wgrd M{M} N{N} K{K}", lambda: ops.p_wgrad(dy, x))
        ab(f"wgrad M{N} N{K} K{M} (+combine)", lambda: ops.p_wgrad(dy, x), fl)
        if N == 768:    # out_proj / c_proj forward with the residual stream: bf16 stream (pq) and fp32 stream (p8)
            r16 = torch.randn(M, N, device=dev).to(BF); r32 = r16.float()
            ab(f"fwd + bf16 residual  N{N} K{K}", lambda: ops.p_linear(x, w, b, residual=r16), fl)
            ok &= check(f"res32 M{M} N{N} K{K}", lambda: ops.p_linear(x, w, b, residual=r32, out_dtype=torch.float32)[0])
            ab(f"fwd + fp32 residual  N{N} K{K}", lambda: ops.p_linear(x, w, b, residual=r32, out_dtype=torch.float32), fl)
        if N >= 2048:   # the MLP pair: c_fc forward (QuickGELU + saved derivative) and the c_proj data gradient (x derivative, + column sums)
            ok &= check(f"act8 M{M} N{N} K{K}", lambda: ops.p_linear(x, w, b, act=ops.ACT_QUICK_GELU, want_aux=True, aux_kind=2, pitched=True))
            ab(f"c_fc fwd gelu+aux8 N{N} K{K}", lambda: ops.p_linear(x, w, b, act=ops.ACT_QUICK_GELU, want_aux=True, aux_kind=2, pitched=True), fl)
            h, u8 = ops.p_linear(x, w, b, act=ops.ACT_QUICK_GELU, want_aux=True, aux_kind=2, pitched=True)
            wt = (torch.randn(K, N, device=dev) * K ** -0.5).to(BF)      # c_proj weight (D, 4D): dgrad dx(M,4D) = g(M,D) @ wt(D,4D)
            ok &= check(f"dact M{M} N{N} K{K}", lambda: ops.p_dgrad(x, wt, BF, aux=u8, act=ops.ACT_QUICK_GELU, aux_kind=2, pitched=True))
            ab(f"c_proj dgrad x act8       N{N} K{K}", lambda: ops.p_dgrad(x, wt, BF, aux=u8, act=ops.ACT_QUICK_GELU, aux_kind=2, pitched=True), fl)
            ab(f"c_proj dgrad x act8 +colsum N{N}", lambda: ops.p_dgrad(x, wt, BF, aux=u8, act=ops.ACT_QUICK_GELU, aux_kind=2, want_colsum=True, pitched=True), fl)
        mode(0)
        t = min(timeit(lambda: torch.matmul(x, w.t())) for _ in range(3))
        print(f"  [torch.matmul] {t:7.1f} us {fl / t / 1e6:6.0f} TF", flush=True)
    print("ALL CHECKS OK" if ok else "MISMATCH", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
