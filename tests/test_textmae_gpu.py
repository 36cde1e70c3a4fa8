"""GPU: the text-MAE branch of the product (modules/modeling.py:226-236, modules/module_clip.py:113-124,
modules/module_mae.py:332-355) against vectors produced by the REAL reference (tests/golden/textmae_tiny.npz) in exact-f32
mode, and at ViT-B/16 dimensions (vocabulary 49408, decoder width 256, 8 heads of 32) in bf16 mode against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import segclip_amd  # noqa: E402
from segclip_amd import synth  # noqa: E402
from tests.helpers import load_golden  # noqa: E402

DEV = "cuda"
FLAGS = dict(use_text_mae_recon=True)


def _run(spec_name, B, seed, dtype, flags):
    spec = synth.SPECS[spec_name]
    segclip_amd.set_compute_dtype(dtype)
    try:
        model, _ = synth.build_model(spec, flags, device=DEV)
        batch = synth.synthetic_batch(spec, B, seed=seed, device=DEV, with_seg=bool(flags.get("use_seglabel")))
        noise = synth.synthetic_noise(spec, B, seed=seed, device=DEV)
        items = [("gumbel", noise["gumbel_main"]), ("rand", noise["text_mask_noise"])]
        if flags.get("use_vision_mae_recon"):
            items += [("rand", noise["mask_noise"]), ("gumbel", noise["gumbel_mae"])]
        with segclip_amd.noise_injection(items):
            loss = model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"],
                         image_seg=batch.get("image_seg"))
        loss.backward()
        torch.cuda.synchronize()
        return model, loss
    finally:
        segclip_amd.set_compute_dtype(torch.float32)


def test_text_mae_tiny_f32_matches_reference_golden():
    g = load_golden("textmae_tiny.npz")
    model, loss = _run("tiny", int(g["B"]), int(g["seed"]), torch.float32, FLAGS)
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-4
    assert abs(float(model.last_losses["text_mae"]) - float(g["loss_text_mae"])) <= 1e-4
    mask, ids_restore, hidden = model.last_text_mae
    assert np.array_equal(mask.cpu().numpy(), g["text_mae_mask"])
    assert np.array_equal(ids_restore.cpu().numpy(), g["text_ids_restore"])
    np.testing.assert_allclose(hidden.detach().cpu().numpy(), g["text_mae_hidden"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(model.last_logits[0].cpu().numpy(), g["t2v"], rtol=0, atol=1e-3)
    P = dict(model.named_parameters())
    np.testing.assert_allclose(P["seq_mae_decoder.decoder_pos_embed"].detach().cpu().numpy(), g["decoder_pos_embed"], atol=1e-6)
    for n, ref in zip(g["grad_names"].tolist(), g["grad_norms"]):
        assert P[n].grad is not None, n
        got = float(P[n].grad.double().norm())
        assert abs(got - ref) <= 5e-3 * max(ref, 1e-4), (n, got, ref)
    for n in g["none_grad"].tolist():
        assert P[n].grad is None or float(P[n].grad.abs().max()) == 0.0, n
    for k in g.files:
        if k.startswith("grad::"):
            np.testing.assert_allclose(P[k[6:]].grad.cpu().numpy(), g[k], rtol=5e-3, atol=1e-5, err_msg=k)


def test_text_mae_vitb16_bf16_against_oracle():
    """Real dimensions (77 tokens -> 65 kept, 49408-way vocabulary loss), every loss switched on, bf16 kernels (the
    key-padding mask goes through the flash-attention kernels as per-sample key counts)."""
    from oracle import segclip_oracle as so
    from tests.helpers import model_param_shapes, oracle_params
    flags = dict(use_text_mae_recon=True, use_seglabel=True, use_vision_mae_recon=True)
    spec = synth.SPECS["vitb16"]
    B, seed = 3, 17
    model, loss = _run("vitb16", B, seed, torch.bfloat16, flags)
    P = oracle_params(spec, model_param_shapes(spec, flags), requires_grad=False)
    with torch.no_grad():
        lo, aux = so.segclip_forward(synth.synthetic_batch(spec, B, seed=seed), P, spec, synth.synthetic_noise(spec, B, seed=seed),
                                     flags)
    mask, ids_restore, hidden = model.last_text_mae
    assert hidden.shape == (B, 65, 512)
    assert torch.equal(mask.cpu(), aux["text_mae_mask"]) and torch.equal(ids_restore.cpu(), aux["text_ids_restore"])
    d_seq = abs(float(model.last_losses["text_mae"]) - float(aux["loss_text_mae"]))
    d_all = abs(float(loss.detach()) - float(lo))
    print(f"\n[text-MAE vitb16 bf16] loss {float(loss.detach()):.4f} vs oracle {float(lo):.4f}; text-MAE term "
          f"{float(model.last_losses['text_mae']):.4f} vs {float(aux['loss_text_mae']):.4f}")
    assert d_seq <= 0.05 and d_all <= 0.1
    g = [p.grad for n, p in model.named_parameters() if n.startswith("seq_mae_decoder.") and p.requires_grad]
    assert all(x is not None and torch.isfinite(x).all() for x in g)
