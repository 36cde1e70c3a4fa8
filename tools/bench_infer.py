"""Inference subset: eval-mode encoders of the bf16 build - images/s of clip.encode_image at 224^2 (retrieval) and
448^2 (the zero-shot segmentation tier's sliding-window size: 784 patches, positional table resampled), captions/s of
clip.encode_text.  usage: python tools/bench_infer.py [batch=64]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import segclip_amd
from segclip_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
segclip_amd.set_compute_dtype(torch.bfloat16)
spec = synth.SPECS["vitb16"]
model, _ = synth.build_model(spec, {}, device="cuda")
model.eval()
b = synth.synthetic_batch(spec, B, seed=0, device="cuda", with_seg=False)
ids = b["input_ids"].view(-1, b["input_ids"].shape[-1])


def run(fn, n=10):
    with torch.no_grad():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for res in (224, 448):
    img = torch.randn(B if res == 224 else max(B // 4, 1), 3, res, res, device="cuda")
    dt = run(lambda: model.clip.encode_image(img, return_hidden=True))
    print(f"encode_image {res}^2  batch {img.shape[0]:4d}: {dt * 1e3:8.2f} ms  {img.shape[0] / dt:9.1f} images/s", flush=True)
dt = run(lambda: model.clip.encode_text(ids))
print(f"encode_text  77 tok   batch {B:4d}: {dt * 1e3:8.2f} ms  {B / dt:9.1f} captions/s", flush=True)
