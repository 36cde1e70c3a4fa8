"""Fused-epilogue GEMM variants of one residual block at the vision-tower shape against the plain kernel, with a
numerical check of every variant against a torch fp32 product (random data, HIP events).
SEGCLIP_P8_TOUCH=0/1 switches the side-tile touch of the 8-phase kernel (read once per process)."""
import os, sys
os.environ.setdefault("SEGCLIP_TUNING", "1")   # the library honours its kernel-selection switches only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segclip_amd import ops
from tools.bench_gemm import timeit

BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 50176
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
F = 4 * D
dev = "cuda"
torch.manual_seed(0)


def qg(x):
    return x * torch.sigmoid(1.702 * x)


def dqg(x):
    s = torch.sigmoid(1.702 * x)
    return s * (1 + 1.702 * x * (1 - s))


def relerr(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


x = torch.randn(M, D, device=dev).to(BF)
h = torch.randn(M, F, device=dev).to(BF)
g = torch.randn(M, D, device=dev).to(BF)
wqkv = (torch.randn(3 * D, D, device=dev) * D ** -0.5).to(BF); bqkv = torch.randn(3 * D, device=dev)
wo = (torch.randn(D, D, device=dev) * D ** -0.5).to(BF); bo = torch.randn(D, device=dev)
wfc = (torch.randn(F, D, device=dev) * D ** -0.5).to(BF); bfc = torch.randn(F, device=dev)
wpr = (torch.randn(D, F, device=dev) * F ** -0.5).to(BF); bpr = torch.randn(D, device=dev)
res = torch.randn(M, D, device=dev)
u = torch.randn(M, F, device=dev).to(BF)
du_ = dqg(u.float()).to(BF)
rows = []


def add(name, fl, fn, check=None):
    t = timeit(fn)
    err = check() if check is not None else float("nan")
    rows.append((name, fl / t / 1e12, t * 1e6, err))


sl = slice(0, 2048)   # rows checked against torch (full columns)
fl_fc, fl_o, fl_qkv = 2.0 * M * D * F, 2.0 * M * D * D, 2.0 * M * D * 3 * D
add("qkv fwd bias                N=3D K=D", fl_qkv, lambda: ops.p_linear(x, wqkv, bqkv),
    lambda: relerr(ops.p_linear(x, wqkv, bqkv)[0][sl], x[sl].float() @ wqkv.float().t() + bqkv))
add("out_proj fwd plain bf16     N=D K=D", fl_o, lambda: ops.p_linear(x, wo, bo))
add("out_proj fwd +res f32       N=D K=D", fl_o, lambda: ops.p_linear(x, wo, bo, residual=res, out_dtype=torch.float32),
    lambda: relerr(ops.p_linear(x, wo, bo, residual=res, out_dtype=torch.float32)[0][sl], x[sl].float() @ wo.float().t() + bo + res[sl]))
add("c_fc fwd bias               N=4D K=D", fl_fc, lambda: ops.p_linear(x, wfc, bfc))
for kind in (0, 1):
    def chk(kind=kind):
        y, a = ops.p_linear(x, wfc, bfc, act=ops.ACT_QUICK_GELU, want_aux=True, aux_kind=kind)
        pre = x[sl].float() @ wfc.float().t() + bfc
        return max(relerr(y[sl], qg(pre)), relerr(a[sl], dqg(pre) if kind else pre))
    add(f"c_fc fwd gelu + aux kind {kind}  N=4D K=D", fl_fc,
        lambda kind=kind: ops.p_linear(x, wfc, bfc, act=ops.ACT_QUICK_GELU, want_aux=True, aux_kind=kind), chk)
add("c_proj fwd plain bf16       N=D K=4D", fl_fc, lambda: ops.p_linear(h, wpr, bpr))
add("c_proj fwd +res f32         N=D K=4D", fl_fc, lambda: ops.p_linear(h, wpr, bpr, residual=res, out_dtype=torch.float32),
    lambda: relerr(ops.p_linear(h, wpr, bpr, residual=res, out_dtype=torch.float32)[0][sl], h[sl].float() @ wpr.float().t() + bpr + res[sl]))
add("c_proj dgrad plain          N=4D K=D", fl_fc, lambda: ops.p_dgrad(g, wpr, BF),
    lambda: relerr(ops.p_dgrad(g, wpr, BF)[sl], g[sl].float() @ wpr.float()))
add("c_proj dgrad * act'(u) k0   N=4D K=D", fl_fc, lambda: ops.p_dgrad(g, wpr, BF, aux=u, act=ops.ACT_QUICK_GELU, aux_kind=0),
    lambda: relerr(ops.p_dgrad(g, wpr, BF, aux=u, act=ops.ACT_QUICK_GELU, aux_kind=0)[sl], (g[sl].float() @ wpr.float()) * dqg(u[sl].float())))
add("c_proj dgrad * aux     k1   N=4D K=D", fl_fc, lambda: ops.p_dgrad(g, wpr, BF, aux=du_, act=ops.ACT_QUICK_GELU, aux_kind=1),
    lambda: relerr(ops.p_dgrad(g, wpr, BF, aux=du_, act=ops.ACT_QUICK_GELU, aux_kind=1)[sl], (g[sl].float() @ wpr.float()) * du_[sl].float()))
def chk_cs():
    dx, cs = ops.p_dgrad(g, wpr, BF, aux=du_, act=ops.ACT_QUICK_GELU, aux_kind=1, want_colsum=True)
    return relerr(cs, dx.float().sum(0))
add("c_proj dgrad * aux k1 +colsum        ", fl_fc, lambda: ops.p_dgrad(g, wpr, BF, aux=du_, act=ops.ACT_QUICK_GELU, aux_kind=1, want_colsum=True), chk_cs)
add("c_fc dgrad plain            N=D K=4D", fl_fc, lambda: ops.p_dgrad(h, wfc, BF))
add("out_proj dgrad plain        N=D K=D", fl_o, lambda: ops.p_dgrad(g, wo, BF))
qkvg = torch.randn(M, 3 * D, device=dev).to(BF)
add("qkv dgrad plain             N=D K=3D", fl_qkv, lambda: ops.p_dgrad(qkvg, wqkv, BF))
print(f"# M={M} D={D}  SEGCLIP_P8_TOUCH={os.environ.get('SEGCLIP_P8_TOUCH', '1')}")
for name, tf, us, err in rows:
    print(f"{name:40s} {tf:8.1f} TF/s {us:9.1f} us   relerr {err:.2e}")
