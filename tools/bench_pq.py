"""Persistent 256x256 GEMM (gemm_bf16_pq.hip) against the one-tile-per-workgroup 8-phase kernel (gemm_bf16_p8.hip) in ONE
process: bit-exact comparison of the outputs (same products, same k order, same rounding) and interleaved timing rounds
(SEGCLIP_GEMM_PQ=2 makes the dispatcher consult SEGCLIP_GEMM_PQ_NOW / SEGCLIP_PQ_PERSIST_NOW at every call)."""
import os, sys
os.environ["SEGCLIP_GEMM_PQ"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segclip_amd import ops

BF = torch.bfloat16
dev = "cuda"


def mode(pq, persist=1):
    os.environ["SEGCLIP_GEMM_PQ_NOW"] = str(pq)
    os.environ["SEGCLIP_PQ_PERSIST_NOW"] = str(persist)


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
    """fn() under p8 / pq one-tile / pq persistent, interleaved; prints median us and TF/s"""
    res = {}
    variants = (("p8", 0, 0), ("pq1", 1, 0), ("pqP", 1, 1))
    for tag, pq, ps in variants:
        mode(pq, ps); fn(); fn()
    for r in range(rounds):
        for tag, pq, ps in variants:
            mode(pq, ps)
            res.setdefault(tag, []).append(timeit(fn))
    out = []
    for tag, _, _ in variants:
        v = sorted(res[tag]); med = v[len(v) // 2]
        out.append(f"{tag} {med:7.1f} us {flops / med / 1e6:6.0f} TF (min {v[0]:.1f})")
    print(f"{name:34s} " + " | ".join(out), flush=True)


def check(name, fn, ref_fn=None):
    mode(0); a = fn().clone()
    mode(1, 0); b = fn().clone()
    mode(1, 1); c = fn().clone()
    torch.cuda.synchronize()
    same1, sameP = bool(torch.equal(a, b)), bool(torch.equal(a, c))
    msg = f"check {name:28s} pq1==p8 {same1}  pqP==p8 {sameP}"
    if ref_fn is not None:
        r = ref_fn()
        msg += f"  relerr vs fp32 torch {float((c.float() - r).norm() / r.norm()):.2e}"
    if not (same1 and sameP):
        d = (a.float() - c.float()).abs()
        msg += f"  max|d| {float(d.max()):.3e} at {int(d.argmax())} nbad {(d > 0).sum().item()}"
    print(msg, flush=True)
    return same1 and sameP


def main():
    torch.manual_seed(0)
    ok = True
    # correctness on small and odd tile counts (588-tile shape included below at full M)
    for (M, N, K) in [(256, 256, 64), (512, 768, 128), (2048, 512, 192), (4096, 2304, 768), (19712, 512, 512)]:
        x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
        b = torch.randn(N, device=dev); dy = torch.randn(M, N, device=dev).to(BF)
        ok &= check(f"fwd  M{M} N{N} K{K}", lambda: ops.p_linear(x, w, b)[0], lambda: x.float() @ w.float().t() + b)
        ok &= check(f"fwd0 M{M} N{N} K{K}", lambda: ops.p_linear(x, w, None)[0])
        ok &= check(f"dgrd M{M} N{N} K{K}", lambda: ops.p_dgrad(dy, w, BF), lambda: dy.float() @ w.float())
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
        mode(0)
        t = timeit(lambda: torch.matmul(x, w.t()))
        print(f"  [torch.matmul] {t:7.1f} us {fl / t / 1e6:6.0f} TF", flush=True)
    print("ALL BIT-EXACT" if ok else "MISMATCH", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
