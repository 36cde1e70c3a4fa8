"""Ablation timing of the phase-pipelined GEMM (set SEGCLIP_P8_ABL before starting): forward layout, two shapes."""
import sys, os
os.environ.setdefault("SEGCLIP_TUNING", "1")   # the library honours its kernel-selection switches only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segclip_amd import ops
BF = torch.bfloat16
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
M = 50176
out = []
for N, K in ((768, 3072), (2304, 768), (768, 768)):
    x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
    us = timeit(lambda: ops.p_linear(x, w, None))
    rounds = -(-(M // 256) * (N // 256) // 256)
    out.append(f"N{N} K{K}: {us:7.1f} us  ({us / rounds:6.1f} us/round, {2.0 * M * N * K / us / 1e6:7.1f} TF)")
print("ABL=%s  " % os.environ.get("SEGCLIP_P8_ABL", "0") + " | ".join(out))
