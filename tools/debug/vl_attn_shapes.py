import sys, os
sys.path.insert(0, "/root/repo")
import torch, segclip_amd
from segclip_amd import ops
orig = ops.p_attn_bwd
seen = set()
def spy(d, *a, **k):
    key = (d.B, d.H, d.Tq, d.Tk, d.hd, d.causal, bool(d.klen), d.flags)
    if key not in seen:
        seen.add(key); print("attn_bwd", key, flush=True)
    return orig(d, *a, **k)
ops.p_attn_bwd = spy
sys.argv = ["bench.py", "--no-roofline", "--no-cpu-baseline", "--no-parity-leg", "--no-traffic", "--spec", "vitl14_336", "--batch", "16", "--steps", "1", "--warmup", "1"]
exec(open("/root/repo/bench.py").read())
