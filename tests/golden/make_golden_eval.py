"""Golden vectors for the inference subset, produced by the REAL reference on CPU in eval mode.

Run in the build container only:   python tests/golden/make_golden_eval.py
Output (committed): tests/golden/eval_tiny.npz
  * get_pos_embed(h, w) of the reference's VisualTransformer for several grids (bicubic resampling of the table)
  * clip.encode_image(image, return_hidden=True) at 2x the training resolution (4x the patches, main SegViT branch,
    no Gumbel noise): features, hidden states, soft / hard center assignment
  * clip.encode_text(ids, return_hidden=True)
Weights are the closed-form generators of segclip_amd/synth.py; the 2x-resolution image is stored in the file.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_harness as rh  # noqa: E402
from segclip_amd import synth  # noqa: E402

GRIDS = [(8, 8), (4, 8), (7, 5), (2, 2), (12, 12)]

if __name__ == "__main__":
    torch.manual_seed(0)
    spec = synth.SPECS["tiny"]
    model, _ = rh.build_reference_model(spec, {}, rank=0, world_size=1, cross_mode="t18")
    synth.apply_closed_form_weights(model)
    model.eval()
    out = {}
    with torch.no_grad():
        for h, w in GRIDS:
            out[f"pos_{h}x{w}"] = model.clip.visual.get_pos_embed(h, w).numpy()
        g = torch.Generator().manual_seed(321)
        image = torch.randn(3, 3, 2 * spec["image_res"], 2 * spec["image_res"], generator=g)
        feat, hidden, mid = model.clip.encode_image(image, return_hidden=True)
        out["image"], out["image_feat"], out["image_hidden"] = image.numpy(), feat.numpy(), hidden.numpy()
        out["soft_attn"] = mid["attns"][0]["soft_attn"].numpy()
        out["hard_idx"] = mid["attns"][0]["hard_attn"].argmax(dim=1).numpy()
        out["layers0_out"] = mid["hidden"].numpy()
        ids = synth.synthetic_batch(spec, 3, seed=55)["input_ids"][:, 0]
        tfeat, thidden = model.clip.encode_text(ids, return_hidden=True)
        out["input_ids"], out["text_feat"], out["text_hidden"] = ids.numpy(), tfeat.numpy(), thidden.numpy()
    np.savez_compressed(os.path.join(HERE, "eval_tiny.npz"), **out)
    print({k: v.shape for k, v in out.items()})
