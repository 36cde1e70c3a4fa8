"""CPU: the oracle's text-MAE branch against vectors produced by the REAL reference
(tests/golden/textmae_tiny.npz <- tests/golden/make_golden_textmae.py): modules/modeling.py:226-236,
modules/module_clip.py:113-124 (masked encode_text incl. the row-0 separator pinning of random_masking),
modules/module_mae.py:332-355 (forward_seq, key-padding mask, ignore_index cross entropy)."""
import numpy as np
import torch

from oracle import segclip_oracle as so
from segclip_amd import synth
from tests.helpers import load_golden, model_param_shapes, oracle_params

FLAGS = dict(use_text_mae_recon=True)


def test_oracle_text_mae_matches_reference():
    g = load_golden("textmae_tiny.npz")
    spec = synth.SPECS["tiny"]
    B, seed = int(g["B"]), int(g["seed"])
    P = oracle_params(spec, model_param_shapes(spec, FLAGS))
    np.testing.assert_allclose(P["seq_mae_decoder.decoder_pos_embed"].numpy(), g["decoder_pos_embed"], rtol=0, atol=1e-6)
    batch = synth.synthetic_batch(spec, B, seed=seed, with_seg=False)
    noise = synth.synthetic_noise(spec, B, seed=seed)
    loss, aux = so.segclip_forward(batch, P, spec, noise, FLAGS)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) <= 2e-5
    assert abs(float(aux["loss_text_mae"].detach()) - float(g["loss_text_mae"])) <= 2e-5
    assert np.array_equal(aux["text_mae_mask"].numpy(), g["text_mae_mask"])
    assert np.array_equal(aux["text_ids_restore"].numpy(), g["text_ids_restore"])
    np.testing.assert_allclose(aux["text_mae_hidden"].detach().numpy(), g["text_mae_hidden"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(aux["t2v"].detach().numpy(), g["t2v"], rtol=0, atol=1e-4)
    for n, ref in zip(g["grad_names"].tolist(), g["grad_norms"]):
        got = float(P[n].grad.double().norm())
        assert abs(got - ref) <= 3e-4 * max(1.0, ref), (n, got, ref)
    for n in g["none_grad"].tolist():
        assert P[n].grad is None or float(P[n].grad.abs().max()) == 0.0, n
    for k in g.files:
        if k.startswith("grad::"):
            np.testing.assert_allclose(P[k[6:]].grad.numpy(), g[k], rtol=1e-3, atol=1e-6, err_msg=k)
