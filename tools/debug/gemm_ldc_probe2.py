"""c_fc forward (two bf16 outputs) and c_proj data gradient (side operand + output) with padded row pitches."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from segclip_amd import ops
from tools.bench_gemm import timeit
BF = torch.bfloat16
for M, D in ((50176, 768), (19712, 512)):
    F = 4 * D
    x = torch.randn(M, D, device="cuda").to(BF); g = torch.randn(M, D, device="cuda").to(BF)
    wfc = (torch.randn(F, D, device="cuda") * D ** -0.5).to(BF); bfc = torch.randn(F, device="cuda")
    wpr = (torch.randn(D, F, device="cuda") * F ** -0.5).to(BF)
    for pad in (0, 256, 512, 1024):
        y = torch.empty(M, F + pad, dtype=BF, device="cuda"); a = torch.empty(M, F + pad, dtype=BF, device="cuda")
        du = torch.empty(M, F + pad, dtype=BF, device="cuda")
        t1 = timeit(lambda: ops.p_gemm(x, wfc, y, M, F, D, (D, 1), (D, 1), F + pad, bias=bfc, aux=a, ldaux=F + pad, act=ops.ACT_QUICK_GELU, aux_kind=1))
        t2 = timeit(lambda: ops.p_gemm(g, wpr, du, M, F, D, (D, 1), (1, F), F + pad, aux=a, ldaux=F + pad, act=ops.ACT_QUICK_GELU, mul_dact=True, aux_kind=1))
        out = torch.empty(M, D, dtype=BF, device="cuda")
        t3 = timeit(lambda: ops.p_gemm(y, wpr, out, M, D, F, (F + pad, 1), (F, 1), D))
        t4 = timeit(lambda: ops.p_gemm(du, wfc, out, M, D, F, (F + pad, 1), (1, D), D))
        print(f"M={M} D={D} pitch F+{pad}: c_fc fwd act+aux {t1*1e6:7.1f} us | c_proj dgrad*aux {t2*1e6:7.1f} us | c_proj fwd (A=h) {t3*1e6:7.1f} us | c_fc dgrad (A=du) {t4*1e6:7.1f} us", flush=True)
