"""GPU: BASELINE.json configs[4] shape - "ViT-L/14" widths at 336^2 (conv1 (1024,3,14,14), 576 patch
tokens + CLS, 16 heads, text width 768; SegViT still builds 10+2 blocks) - through the product path:
 * exact-f32 mode against the CPU oracle on the same seeded inputs (loss / logits 1e-3, hard_idx bit-exact);
 * bf16 mode against the f32 run (bounded);
exercising the streaming attention backward (576 > 256 tokens), the zero-padded patch GEMM (3*14*14 = 588 -> 640) and
the 584-key center cross-attention."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import segclip_amd  # noqa: E402
from segclip_amd import synth  # noqa: E402

DEV = "cuda"


def _run(dtype, B, seed, attn_fp8=False):
    spec = synth.SPECS["vitl14_336"]
    segclip_amd.set_compute_dtype(dtype)
    segclip_amd.config.attn_fp8 = attn_fp8
    try:
        model, _ = synth.build_model(spec, {}, device=DEV)
        batch = synth.synthetic_batch(spec, B, seed=seed, device=DEV, with_seg=False)
        noise = synth.synthetic_noise(spec, B, seed=seed, device=DEV)
        with segclip_amd.noise_injection([("gumbel", noise["gumbel_main"])]):
            loss = model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"])
        loss.backward()
        torch.cuda.synchronize()
        out = dict(loss=float(loss.detach()), t2v=model.last_logits[0].float().cpu(),
                   hard_idx=model.last_mid_states["hard_idx"].cpu().long(),
                   gn={n: float(p.grad.double().norm()) for n, p in model.named_parameters() if p.grad is not None})
        del model
        torch.cuda.empty_cache()
        return out
    finally:
        segclip_amd.set_compute_dtype(torch.float32)
        segclip_amd.config.attn_fp8 = False


def test_vitl14_336_f32_matches_oracle_and_bf16_is_bounded():
    from oracle import segclip_oracle as so
    from tests.helpers import model_param_shapes, oracle_params
    spec = synth.SPECS["vitl14_336"]
    B, seed = 2, 5
    f = _run(torch.float32, B, seed)
    P = oracle_params(spec, model_param_shapes(spec, {}))
    lo, aux = so.segclip_forward(synth.synthetic_batch(spec, B, seed=seed, with_seg=False), P, spec,
                                 synth.synthetic_noise(spec, B, seed=seed), {})
    lo.backward()
    assert f["hard_idx"].shape == (B, 576)
    assert abs(f["loss"] - float(lo)) <= 1e-3, (f["loss"], float(lo))
    assert float((f["t2v"] - aux["t2v"].detach()).abs().max()) <= 1e-3
    assert torch.equal(f["hard_idx"], aux["hard_idx"])
    worst = 0.0
    for n, p in P.items():
        if p.grad is None or n not in f["gn"]:
            continue
        ref = float(p.grad.double().norm())
        if ref > 1e-7:
            worst = max(worst, abs(f["gn"][n] - ref) / ref)
            assert abs(f["gn"][n] - ref) <= 2e-2 * ref + 1e-8, (n, f["gn"][n], ref)
    b = _run(torch.bfloat16, B, seed)
    dl, dlog = abs(b["loss"] - f["loss"]), float((b["t2v"] - f["t2v"]).abs().max())
    agree = float((b["hard_idx"] == f["hard_idx"]).float().mean())
    rat = [b["gn"][n] / f["gn"][n] for n in f["gn"] if f["gn"][n] > 1e-6]
    print(f"\n[vitl14_336 B={B}] f32 vs oracle: worst grad-norm rel err {worst:.2e}; bf16 vs f32: d loss {dl:.2e}, "
          f"max |dlogit| {dlog:.4f}, hard_idx agreement {agree:.4f}, grad-norm ratio median {np.median(rat):.4f}")
    # B = 2: the loss is a 2 x 2 contrastive problem, and each of the ~1 % of patches whose 8-way argmax flips under bf16 moves
    # it.  WHICH patches flip depends on last-bit differences of the attention output: the same build gives d loss 3e-3 /
    # |dlogit| 0.017 or 3.2e-2 / 0.13 depending on the rounding order inside the softmax (SEGCLIP_ATTN_FWD_LEAN=0/1; the
    # kernel outputs agree to the last bf16 digit almost everywhere, tools/debug/attn_fwd_check.py) - bounds cover both
    assert dl <= 0.06 and dlog <= 0.3 and agree >= 0.97
    assert 0.95 <= float(np.median(rat)) <= 1.05
    # BASELINE configs[4] as first read: the e4m3 QK^T / PV forward in every self-attention block.  Round 5: slower than bf16 in
    # every round, so the kernel left the default build (configs[4] runs bf16) - exercised in experiments builds
    from segclip_amd import _lib
    try:
        e = _run(torch.bfloat16, B, seed, attn_fp8=True)
    except _lib.Unsupported:
        print("[vitl14_336] e4m3 attention forward: not in the default build")
        return
    dl8, dlog8 = abs(e["loss"] - f["loss"]), float((e["t2v"] - f["t2v"]).abs().max())
    agree8 = float((e["hard_idx"] == f["hard_idx"]).float().mean())
    rat8 = [e["gn"][n] / f["gn"][n] for n in f["gn"] if f["gn"][n] > 1e-6]
    print(f"[vitl14_336 B={B}] bf16 + fp8 attention forward vs f32: d loss {dl8:.2e}, max |dlogit| {dlog8:.4f}, hard_idx "
          f"agreement {agree8:.4f}, grad-norm ratio median {np.median(rat8):.4f}")
    assert dl8 <= 0.05 and dlog8 <= 0.5 and agree8 >= 0.9
    assert 0.9 <= float(np.median(rat8)) <= 1.1


def test_vitl14_336_b128_the_benchmarked_size_properties():
    """BASELINE configs[4] at its named per-GPU batch (128), bf16 - the configuration `bench.py --spec vitl14_336` times -
    through size-independent properties (the oracle needs minutes at this size): every patch assigned to exactly one center,
    logits bounded by the clamped scale and t2v == v2t^T on one rank, loss near ln B at random init, finite gradients for
    every trainable parameter, zero gradient for the dropped class embedding, and batch-permutation equivariance of the
    per-sample hard assignment in the 'intended' cross-attention mode."""
    spec = synth.SPECS["vitl14_336"]
    B = 128
    segclip_amd.set_compute_dtype(torch.bfloat16)
    segclip_amd.set_cross_mode("intended")
    try:
        model, _ = synth.build_model(spec, {}, device=DEV)
        batch = synth.synthetic_batch(spec, B, seed=9, device=DEV, with_seg=False)
        noise = synth.synthetic_noise(spec, B, seed=9, device=DEV)
        with segclip_amd.noise_injection([("gumbel", noise["gumbel_main"])]):
            loss = model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"])
        loss.backward()
        torch.cuda.synchronize()
        t2v, v2t = model.last_logits
        hard = model.last_mid_states["attns"][0]["hard_attn"]
        assert hard.shape == (B, 8, 576)
        assert torch.equal(hard.sum(1), torch.ones_like(hard.sum(1)))
        assert float((t2v - v2t.t()).abs().max()) <= 1e-3 * float(t2v.abs().max())
        assert float(t2v.abs().max()) <= 100.0
        assert abs(float(loss) - float(np.log(B))) < 1.0, float(loss)
        for n, p in model.named_parameters():
            if p.grad is not None:
                assert torch.isfinite(p.grad).all(), n
        assert float(model.clip.visual.class_embedding.grad.abs().max()) == 0.0
        hard_idx = model.last_mid_states["hard_idx"].clone()
        # permuted batch, same per-sample noise: every sample keeps its assignment map
        perm = torch.randperm(B, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
        with torch.no_grad(), segclip_amd.noise_injection([("gumbel", noise["gumbel_main"][perm])]):
            model(batch["input_ids"][perm], batch["segment_ids"][perm], batch["input_mask"][perm], batch["image"][perm])
        agree = float((model.last_mid_states["hard_idx"] == hard_idx[perm]).float().mean())
        print(f"\n[vitl14_336 B=128 bf16] loss {float(loss):.4f} (ln B = {np.log(B):.4f}); hard_idx agreement under a batch permutation {agree:.5f}")
        assert agree >= 0.999, agree
        del model
        torch.cuda.empty_cache()
    finally:
        segclip_amd.set_compute_dtype(torch.float32)
        segclip_amd.set_cross_mode("t18")
