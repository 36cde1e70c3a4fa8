"""Micro-benchmark of the GEMM layouts at the ViT-B/16 B=256 shapes (random data, HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segclip_amd import ops

dev = "cuda"
BF = torch.bfloat16


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 50176
    rows = []
    shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
    if len(sys.argv) > 2 and sys.argv[2] == "text":
        shapes = [(1536, 512), (512, 512), (2048, 512), (512, 2048)]
    for (N, K) in shapes:
        x = torch.randn(M, K, device=dev).to(BF)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
        b = torch.randn(N, device=dev)
        dy = torch.randn(M, N, device=dev).to(BF)
        dyf = dy.float()
        res = torch.randn(M, N, device=dev)
        fl = 2.0 * M * N * K
        t = timeit(lambda: ops.p_linear(x, w, b))
        rows.append((f"fwd NT bf16out  M{M} N{N} K{K}", fl / t / 1e12, t * 1e6))
        t = timeit(lambda: ops.p_linear(x, w, b, residual=res, out_dtype=torch.float32))
        rows.append((f"fwd NT +res f32 M{M} N{N} K{K}", fl / t / 1e12, t * 1e6))
        t = timeit(lambda: ops.p_linear(x, w, b, act=ops.ACT_QUICK_GELU, want_aux=True))
        rows.append((f"fwd NT gelu+aux M{M} N{N} K{K}", fl / t / 1e12, t * 1e6))
        t = timeit(lambda: ops.p_dgrad(dy, w, BF))
        rows.append((f"dgrad bf16      M{M} N{N} K{K}", fl / t / 1e12, t * 1e6))
        t = timeit(lambda: ops.p_dgrad(dyf, w, BF))
        rows.append((f"dgrad f32 A     M{M} N{N} K{K}", fl / t / 1e12, t * 1e6))
        t = timeit(lambda: ops.p_wgrad(dy, x))
        rows.append((f"wgrad bf16      M{M} N{N} K{K}", fl / t / 1e12, t * 1e6))
        t = timeit(lambda: ops.p_wgrad(dyf, x))
        rows.append((f"wgrad f32 A     M{M} N{N} K{K}", fl / t / 1e12, t * 1e6))
        t = timeit(lambda: ops.p_colsum(dy))
        rows.append((f"colsum bf16     M{M} N{N}", M * N * 2 / t / 1e12, t * 1e6))
        # torch (hipBLASLt) reference point for the plain NT product
        t = timeit(lambda: torch.matmul(x, w.t()))
        rows.append((f"[torch.matmul]  M{M} N{N} K{K}", fl / t / 1e12, t * 1e6))
    for name, tf, us in rows:
        print(f"{name:42s} {tf:8.1f} T(F|B)/s {us:9.1f} us")


if __name__ == "__main__":
    main()
