"""CPU: the oracle (oracle/segclip_oracle.py) reproduces the golden vectors that were produced by
the REAL reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import segclip_oracle as so
from segclip_amd import synth
from tests.helpers import FULL_FLAGS, load_golden, model_param_shapes, oracle_params


def _run(spec_name, B, seed, mode):
    spec = synth.SPECS[spec_name]
    P = oracle_params(spec, model_param_shapes(spec, FULL_FLAGS))
    batch = synth.synthetic_batch(spec, B, seed=seed)
    noise = synth.synthetic_noise(spec, B, seed=seed)
    loss, aux = so.segclip_forward(batch, P, spec, noise, FULL_FLAGS, cross_mode=mode)
    loss.backward()
    return P, aux


@pytest.mark.parametrize("mode", ["t18", "intended"])
def test_tiny_full_tensors(mode):
    g = load_golden(f"tiny_{mode}.npz")
    P, aux = _run("tiny", int(g["B"]), int(g["seed"]), mode)
    for k in ("loss", "loss_contrastive", "loss_kl", "loss_mae"):
        assert abs(float(aux[k]) - float(g[k])) <= 1e-5, k
    for k in ("t2v", "v2t", "text_feat", "image_feat", "text_hidden", "image_hidden", "layers0_out", "soft"):
        np.testing.assert_allclose(aux[k].detach().numpy(), g[k], rtol=2e-4, atol=2e-5, err_msg=k)
    # integer paths: bit exact
    for k in ("eot", "hard_idx", "ids_restore", "mae_hard_idx"):
        assert np.array_equal(aux[k].numpy(), g[k]), k
    assert np.array_equal(aux["mae_mask"].numpy(), g["mae_mask"])
    names = g["grad_names"].tolist()
    for n, ref in zip(names, g["grad_norms"]):
        got = float(P[n].grad.double().norm())
        assert abs(got - ref) <= 2e-4 * max(1.0, ref), (n, got, ref)
    for n in g["none_grad"].tolist():
        assert P[n].grad is None or float(P[n].grad.abs().max()) == 0.0, n
    for k in g.files:
        if k.startswith("grad::"):
            np.testing.assert_allclose(P[k[6:]].grad.numpy(), g[k], rtol=2e-3, atol=2e-6, err_msg=k)


def test_vitb16_b4_scalars():
    """BASELINE.json config 1: ViT-B/16 + 77-token text, batch 4, full loss, one fwd+bwd."""
    g = load_golden("vitb16_b4_t18.npz")
    P, aux = _run("vitb16", int(g["B"]), int(g["seed"]), "t18")
    for k in ("loss", "loss_contrastive", "loss_kl", "loss_mae"):
        assert abs(float(aux[k]) - float(g[k])) <= 1e-4, (k, float(aux[k]), float(g[k]))
    np.testing.assert_allclose(aux["t2v"].detach().numpy(), g["t2v"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(aux["v2t"].detach().numpy(), g["v2t"], rtol=0, atol=1e-3)
    for k in ("eot", "hard_idx", "ids_restore", "mae_hard_idx"):
        assert np.array_equal(aux[k].numpy(), g[k]), k
    names = g["grad_names"].tolist()
    worst = 0.0
    for n, ref in zip(names, g["grad_norms"]):
        got = float(P[n].grad.double().norm())
        worst = max(worst, abs(got - ref) / max(1e-6, ref))
    assert worst <= 5e-3, worst


def test_zero_and_none_grad_contract():
    """SURVEY.md 'None-vs-zero gradient contract': contrastive-only -> layers_mae2 / reconstruct get no
    gradient; class_embedding gets an exactly-zero gradient."""
    spec = synth.SPECS["tiny"]
    flags = {}
    P = oracle_params(spec, model_param_shapes(spec, flags))
    batch = synth.synthetic_batch(spec, 2, seed=5)
    noise = synth.synthetic_noise(spec, 2, seed=5)
    loss, _ = so.segclip_forward(batch, P, spec, noise, flags)
    loss.backward()
    none = {n for n, p in P.items() if p.requires_grad and p.grad is None}
    assert all(("layers_mae2" in n or "reconstruct_layer2" in n) for n in none) and none
    assert float(P["clip.visual.class_embedding"].grad.abs().max()) == 0.0


def test_eval_mode_inference_subset_matches_reference():
    """eval-mode positional-table resampling (bicubic) and encode_image at 2x resolution / encode_text,
    against tests/golden/eval_tiny.npz (the real reference in eval mode)."""
    g = load_golden("eval_tiny.npz")
    spec = synth.SPECS["tiny"]
    P = oracle_params(spec, model_param_shapes(spec, {}), requires_grad=False)
    pos = P["clip.visual.positional_embedding"]
    for k in g.files:
        if k.startswith("pos_"):
            h, w = (int(v) for v in k[4:].split("x"))
            np.testing.assert_allclose(so.interp_pos_embed(pos, h, w).numpy(), g[k], rtol=1e-5, atol=2e-6, err_msg=k)
    feat, hidden, _, _, mid = so.encode_image(torch.from_numpy(g["image"]), P, spec, gumbel=None, eval_pos_interp=True)
    np.testing.assert_allclose(mid["hidden"].numpy(), g["layers0_out"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(feat.numpy(), g["image_feat"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(hidden.numpy(), g["image_hidden"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(mid["attns"][0]["soft_attn"].numpy(), g["soft_attn"], rtol=2e-4, atol=2e-6)
    assert np.array_equal(mid["hard_idx"].numpy(), g["hard_idx"])
    tfeat, thidden, _ = so.encode_text(torch.from_numpy(g["input_ids"]), P, spec)
    np.testing.assert_allclose(tfeat.numpy(), g["text_feat"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(thidden.numpy(), g["text_hidden"], rtol=2e-4, atol=2e-5)
