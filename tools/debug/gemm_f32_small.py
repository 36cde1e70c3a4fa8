import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from segclip_amd import ops
from tools.bench_gemm import timeit
for (M, N, K) in [(256, 256, 512), (256, 512, 768), (256, 512, 512), (2048, 2048, 512), (50176, 768, 768)]:
    a = torch.randn(1, M, K, device="cuda"); b = torch.randn(1, N, K, device="cuda")
    t = timeit(lambda: ops.bmm(a, b, transB=True))
    ref = (a[0].double() @ b[0].double().t()).float()
    err = float((ops.bmm(a, b, transB=True)[0] - ref).abs().max())
    bt = torch.randn(1, K, N, device="cuda")
    t2 = timeit(lambda: ops.bmm(a, bt))
    err2 = float((ops.bmm(a, bt)[0] - (a[0].double() @ bt[0].double()).float()).abs().max())
    print(f"SMALL={os.environ.get('SEGCLIP_GEMM_F32_SMALL','auto')} {M}x{N}x{K}: NT {t*1e6:7.1f} us err {err:.2e} | NN {t2*1e6:7.1f} us err {err2:.2e}")
