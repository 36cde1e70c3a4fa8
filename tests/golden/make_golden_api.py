"""Golden vectors for the remaining public entry points of the path's module API,
produced by the REAL reference on CPU in eval mode (no RNG draws: the center assignment is noise-free in eval).

Run in the build container only:   python tests/golden/make_golden_api.py
Output (committed): tests/golden/api_tiny.npz
  * CLIP.forward(image, text)                       modules/module_clip.py:145-159
  * SegCLIP.get_sequence_output (+return_hidden)     modules/modeling.py:258-279
  * SegCLIP.get_visual_output (+return_hidden)       modules/modeling.py:281-303
  * SegCLIP.get_sequence_visual_output               modules/modeling.py:305-320
  * SegCLIP._loose_similarity / get_similarity_logits in eval mode (t2v, t2v.T)   modules/modeling.py:338-368
Weights are the closed-form generators of segclip_amd/synth.py, inputs synth.synthetic_batch(tiny, B=3, seed=41).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_harness as rh  # noqa: E402
from segclip_amd import synth  # noqa: E402

B, SEED = 3, 41

if __name__ == "__main__":
    torch.manual_seed(0)
    spec = synth.SPECS["tiny"]
    model, _ = rh.build_reference_model(spec, {}, rank=0, world_size=1, cross_mode="t18")
    synth.apply_closed_form_weights(model)
    model.eval()
    batch = synth.synthetic_batch(spec, B, seed=SEED, with_seg=False)
    ids, seg, msk, image = batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"]
    out = {"B": np.asarray(B), "seed": np.asarray(SEED)}
    with torch.no_grad():
        lpi, lpt = model.clip(image[:, 0], ids[:, 0])
        out["clip_logits_per_image"], out["clip_logits_per_text"] = lpi.numpy(), lpt.numpy()
        so = model.get_sequence_output(ids, seg, msk)
        out["sequence_output"] = so.numpy()
        so2, sh2 = model.get_sequence_output(ids, seg, msk, return_hidden=True)
        out["sequence_output_h"], out["sequence_hidden"] = so2.numpy(), sh2.numpy()
        vo = model.get_visual_output(image)
        out["visual_output"] = vo.numpy()
        vo2, vh2, mid = model.get_visual_output(image, return_hidden=True)
        out["visual_output_h"], out["visual_hidden"] = vo2.numpy(), vh2.numpy()
        out["hard_idx"] = mid["attns"][0]["hard_attn"].argmax(dim=1).numpy()
        s3, v3 = model.get_sequence_visual_output(ids, seg, msk, image)
        out["sv_sequence_output"], out["sv_visual_output"] = s3.numpy(), v3.numpy()
        t2v, v2t = model._loose_similarity(so, vo)
        out["eval_t2v"], out["eval_v2t"] = t2v.numpy(), v2t.numpy()
        t2v2, v2t2, _ = model.get_similarity_logits(so, vo, msk)
        out["sim_t2v"], out["sim_v2t"] = t2v2.numpy(), v2t2.numpy()
        t2v3, _ = model._loose_similarity(so, vo, logit_scale=torch.tensor(5.5))   # clamp(exp(5.5), max=100) = 100
        out["eval_t2v_clamped"] = t2v3.numpy()
    np.savez_compressed(os.path.join(HERE, "api_tiny.npz"), **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()})
