"""Time the training-step tail (grad norm + fused AdaptAdamW + finish) and a whole train iteration on one GPU.
usage: python tools/bench_train_tail.py [--batch 256] [--steps 10]"""
import argparse
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import segclip_amd  # noqa: E402
from segclip_amd import synth, train  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
spec = synth.SPECS["vitb16"]
segclip_amd.set_compute_dtype(torch.bfloat16)
model, _ = synth.build_model(spec, {}, device="cuda")
args = types.SimpleNamespace(lr=4e-3, lower_lr=4e-6, lower_text_lr=0.0, weight_decay=0.05, opt_b1=0.9, opt_b2=0.98, eps=1e-6,
                             warmup_proportion=0.15, lr_start=0.0, lr_end=0.0, clip_grad=1.0, freeze_layer_num=0,
                             freeze_text_layer_num=0, first_stage_layer=10, pretrained_clip_name="ViT-B/16")
train.freeze_parameters(args, model)
opt, _, model, _ = train.prep_optimizer(args, model, 1000)
tail = train.TrainTail(model, opt, 1.0)
b = synth.synthetic_batch(spec, a.batch, seed=0, device="cuda")


def fwd_bwd():
    loss = model(b["input_ids"], b["segment_ids"], b["input_mask"], b["image"])
    loss.backward()
    return loss


def timed(fn, n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for _ in range(3):
    tail.run(fwd_bwd())
full = timed(lambda: tail.run(fwd_bwd()), a.steps)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tails = []
for _ in range(a.steps):
    loss = fwd_bwd()
    e0.record()
    tail.run(loss)
    e1.record()
    torch.cuda.synchronize()
    tails.append(e0.elapsed_time(e1))
n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
tail_ms = sorted(tails)[len(tails) // 2]
traffic = n_train * (4 + 16 + 12 + 2)  # norm read + step read + step write + bf16 shadow
print(f"train iteration {full:.2f} ms ({a.batch / full * 1e3:.0f} pairs/s incl. optimizer), tail {tail_ms:.3f} ms for "
      f"{n_train / 1e6:.1f}M trainable params -> {traffic / tail_ms / 1e6:.0f} GB/s of 8000; stats {tail.read()}")
