"""cProfile of the host side of 5 training steps (GPU box)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import segclip_amd
from segclip_amd import synth

segclip_amd.set_compute_dtype(torch.bfloat16)
spec = synth.SPECS["vitb16"]
model, _ = synth.build_model(spec, {}, device="cuda")
B = int(os.environ.get("BATCH", "256"))
b = synth.synthetic_batch(spec, B, seed=0, device="cuda", with_seg=False)


def step():
    model.zero_grad(set_to_none=True)
    loss = model(b["input_ids"], b["segment_ids"], b["input_mask"], b["image"])
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(os.environ.get("TOP", "22")))
