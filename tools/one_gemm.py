import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segclip_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "nt"
dev, BF = "cuda", torch.bfloat16
x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
dy = torch.randn(M, N, device=dev).to(BF)
for _ in range(3):
    if mode == "nt": ops.p_linear(x, w, None)
    elif mode == "dgrad": ops.p_dgrad(dy, w, BF)
    else: ops.p_wgrad(dy, x)
torch.cuda.synchronize()
