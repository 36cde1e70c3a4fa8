"""The assignment-logits products of the semantic-group block (exact fp32, batched, M = 8 centers) and their backward."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from segclip_amd import ops
from tools.bench_gemm import timeit
B, G, T, D = 256, 8, 196, 768
q = torch.randn(B, G, D, device="cuda", requires_grad=True); k = torch.randn(B, T, D, device="cuda", requires_grad=True)
t = timeit(lambda: ops.bmm(q, k, transB=True, out_dtype=torch.float32))
out = ops.bmm(q, k, transB=True, out_dtype=torch.float32)
ref = q.detach().double() @ k.detach().double().transpose(1, 2)
print(f"fwd attn = q k^T  (B,8,196): {t*1e6:7.1f} us  err {float((out - ref.float()).abs().max()):.2e}")
g = torch.randn_like(out)
def bwd():
    q.grad = None; k.grad = None
    o = ops.bmm(q, k, transB=True, out_dtype=torch.float32); o.backward(g)
tb = timeit(bwd)
print(f"fwd+bwd: {tb*1e6:7.1f} us  -> bwd ~{(tb - t)*1e6:7.1f} us")
bwd()
dq_ref = (g.double() @ k.detach().double()).float(); dk_ref = (g.double().transpose(1, 2) @ q.detach().double()).float()
print("dq err", float((q.grad - dq_ref).abs().max()), "dk err", float((k.grad - dk_ref).abs().max()))
