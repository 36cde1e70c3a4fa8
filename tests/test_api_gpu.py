"""GPU: the remaining public entry points of the path's module API against vectors produced by the REAL reference
(tests/golden/api_tiny.npz <- tests/golden/make_golden_api.py): CLIP.forward (modules/module_clip.py:145-159),
SegCLIP.get_sequence_output / get_visual_output / get_sequence_visual_output (modules/modeling.py:258-320) and the
eval branch of _loose_similarity / get_similarity_logits (modules/modeling.py:338-368).
Tolerance: exact-f32 mode 1e-3 on features / logits (north_star), hard_idx bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import segclip_amd  # noqa: E402
from segclip_amd import synth  # noqa: E402
from tests.helpers import load_golden  # noqa: E402

DEV = "cuda"


def _close(got, ref, tol, what):
    err = float((got.detach().float().cpu() - torch.from_numpy(np.asarray(ref))).abs().max())
    assert err <= tol, (what, err)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2)])
def test_module_api_entry_points_match_reference(dtype, tol):
    g = load_golden("api_tiny.npz")
    spec = synth.SPECS["tiny"]
    segclip_amd.set_compute_dtype(dtype)
    try:
        model, _ = synth.build_model(spec, {}, device=DEV)
        model.eval()
        b = synth.synthetic_batch(spec, int(g["B"]), seed=int(g["seed"]), device=DEV, with_seg=False)
        ids, seg, msk, image = b["input_ids"], b["segment_ids"], b["input_mask"], b["image"]
        ltol = tol * 20 if dtype == torch.bfloat16 else tol     # logits carry the x14.3 logit scale
        with torch.no_grad():
            lpi, lpt = model.clip(image[:, 0], ids[:, 0])
            _close(lpi, g["clip_logits_per_image"], ltol, "CLIP.forward logits_per_image")
            _close(lpt, g["clip_logits_per_text"], ltol, "CLIP.forward logits_per_text")
            so = model.get_sequence_output(ids, seg, msk)
            assert tuple(so.shape) == g["sequence_output"].shape and so.dtype == torch.float32
            _close(so, g["sequence_output"], tol, "get_sequence_output")
            so2, sh2 = model.get_sequence_output(ids, seg, msk, return_hidden=True)
            _close(so2, g["sequence_output_h"], tol, "get_sequence_output(return_hidden) pooled")
            _close(sh2, g["sequence_hidden"], tol, "get_sequence_output(return_hidden) hidden")
            vo = model.get_visual_output(image)
            assert tuple(vo.shape) == g["visual_output"].shape
            _close(vo, g["visual_output"], tol, "get_visual_output")
            vo2, vh2, mid = model.get_visual_output(image, return_hidden=True)
            _close(vo2, g["visual_output_h"], tol, "get_visual_output(return_hidden) pooled")
            _close(vh2, g["visual_hidden"], tol, "get_visual_output(return_hidden) hidden")
            same = (mid["hard_idx"].cpu().long().numpy() == g["hard_idx"]).mean()
            assert same == 1.0 if dtype == torch.float32 else same >= 0.9, same
            s3, v3 = model.get_sequence_visual_output(ids, seg, msk, image)
            _close(s3, g["sv_sequence_output"], tol, "get_sequence_visual_output text")
            _close(v3, g["sv_visual_output"], tol, "get_sequence_visual_output image")
            # eval branch of the similarity: fed with the REFERENCE's embeddings so that only this function is tested
            rso = torch.from_numpy(g["sequence_output"]).to(DEV)
            rvo = torch.from_numpy(g["visual_output"]).to(DEV)
            t2v, v2t = model._loose_similarity(rso, rvo)
            _close(t2v, g["eval_t2v"], 1e-3, "_loose_similarity eval t2v")
            _close(v2t, g["eval_v2t"], 1e-3, "_loose_similarity eval v2t")
            assert torch.equal(v2t, t2v.T)
            a, bb, extra = model.get_similarity_logits(rso, rvo, msk)
            _close(a, g["sim_t2v"], 1e-3, "get_similarity_logits t2v")
            _close(bb, g["sim_v2t"], 1e-3, "get_similarity_logits v2t")
            assert extra == ()
            t3, _ = model._loose_similarity(rso, rvo, logit_scale=torch.tensor(5.5, device=DEV))
            _close(t3, g["eval_t2v_clamped"], 1e-3, "_loose_similarity clamp(exp(5.5), 100)")
    finally:
        segclip_amd.set_compute_dtype(torch.float32)
