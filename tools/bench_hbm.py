"""HBM-bound kernels of the step, timed alone (HIP events, 20 launches, buffers rotated so that the 256 MiB memory-side
cache does not serve the reads) against their ALGORITHMIC bytes.  Peak 8 TB/s (MI355X_MICROARCH.md).
usage: python tools/bench_hbm.py  -> table for profiles/<round>_hbm_kernels.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segclip_amd import ops

dev, BF, F32 = "cuda", torch.bfloat16, torch.float32
NROT = 6


def timeit(fns, reps=20):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


rows_out = []


def report(name, nbytes, t):
    rows_out.append((name, nbytes / 1e6, t * 1e6, nbytes / t / 1e12))


for rows, cols, tag in ((256 * 197, 768, "vision"), (256 * 77, 512, "text")):
    xs = [torch.randn(rows, cols, device=dev) for _ in range(NROT)]
    w, b = torch.ones(cols, device=dev), torch.zeros(cols, device=dev)
    t = timeit([lambda x=x: ops.p_ln_fwd(x, w, b, 1e-5, BF) for x in xs])
    report(f"layernorm fwd  fp32 -> bf16      {tag} [{rows},{cols}]", rows * cols * (4 + 2), t)
    y, mean, rstd = ops.p_ln_fwd(xs[0], w, b, 1e-5, BF)
    dys = [torch.randn(rows, cols, device=dev).to(BF) for _ in range(NROT)]
    drs = [torch.randn(rows, cols, device=dev) for _ in range(NROT)]
    t = timeit([lambda i=i: ops.p_ln_bwd(dys[i], xs[i], w, mean, rstd, dres=drs[i], dx_dtype=F32, want_bf16=True,
                                         want_dres_colsum=True) for i in range(NROT)])
    report(f"layernorm bwd  +dres, dx fp32+bf16, dgamma/dbeta/colsum  {tag}", rows * cols * (2 + 4 + 4 + 4 + 2), t)
    t = timeit([lambda i=i: ops.p_colsum(dys[i]) for i in range(NROT)])
    report(f"colsum bf16    {tag} [{rows},{cols}]", rows * cols * 2, t)
    t = timeit([lambda i=i: ops.p_cast(xs[i], BF) for i in range(NROT)])
    report(f"cast fp32->bf16 {tag} [{rows},{cols}]", rows * cols * 6, t)
    del xs, dys, drs

# split-K weight gradient: GEMM + slab reduction; the reduction's bytes = S slabs read + one fp32 write
M = 256 * 197
for (N, K) in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    dy = torch.randn(M, N, device=dev).to(BF); x = torch.randn(M, K, device=dev).to(BF)
    ops._GemmProfile.start() if hasattr(ops, "_GemmProfile") else None
    t = timeit([lambda: ops.p_wgrad(dy, x)])
    if hasattr(ops, "_GemmProfile"):
        rec = ops._GemmProfile.stop()
    report(f"wgrad dW[{N},{K}] GEMM + split-K reduce (operands {M * (N + K) * 2 / 1e6:.0f} MB)", M * (N + K) * 2 + N * K * 4, t)

print("%-78s %9s %9s %8s %6s" % ("kernel (algorithmic bytes)", "MB", "us", "TB/s", "frac"))
for name, mb, us, tbs in rows_out:
    print("%-78s %9.1f %9.1f %8.2f %6.2f" % (name, mb, us, tbs, tbs / 8.0))
