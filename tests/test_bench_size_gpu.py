"""GPU: numeric parity AT THE SIZE THE BENCH RUNS (BASELINE.json configs[1]: ViT-B/16, per-GPU batch 256).

(i)  every large GEMM shape of the bf16 training step - M = 256*196 = 50176 rows, N/K in {768, 2304, 3072} - forward,
     dgrad (k-strided weight) and wgrad (both operands k-strided, split-K over the 50176-long contraction), incl. the
     fused epilogues the step uses (bias, QuickGELU + pre-activation, fp32 residual, act' and bias-gradient column
     sums), EVERY output element against a torch fp32 matmul of the same bf16 operands (so the >256-workgroup
     staggered start, the multi-round XCD remap and the split-K slabs are all numerically checked);
(ii) the whole model at B=256 in bf16 against the SAME model in exact-f32 mode (the mode gated to 1e-3 against the
     reference): loss, logits, per-parameter gradient-norm ratio, hard_idx agreement, with numeric bounds;
(iii) bf16 gradient norms at B=4 against the reference's golden gradient norms.
Bounds are ~3x the errors measured on MI355X (printed by the tests)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import segclip_amd  # noqa: E402
from segclip_amd import ops, synth  # noqa: E402
from tests.helpers import FULL_FLAGS, load_golden, noise_items  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16
M_BENCH = 256 * 196


def _rnd(*shape, seed, scale=1.0, dtype=BF):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=DEV) * scale).to(dtype)


def _check(got, ref, rtol, atol, what):
    err = (got.float() - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} mismatches, max err {float(err.max()):.3e} (ref max {float(ref.abs().max()):.3e})"
    return float(err.max())


@pytest.mark.parametrize("N,K", [(2304, 768), (768, 768), (3072, 768), (768, 3072)])
def test_bench_gemm_shapes_fwd_dgrad_wgrad(N, K):
    """Every element of every output is compared (torch fp32 matmul of the same bf16 operands as the reference)."""
    M = M_BENCH
    x = _rnd(M, K, seed=1)
    w = _rnd(N, K, seed=2, scale=K ** -0.5)
    b = _rnd(N, seed=3, dtype=torch.float32)
    xf, wf = x.float(), w.float()
    # ---- forward: plain (+bias) bf16 out; QuickGELU + pre-activation; fp32 residual + fp32 out
    ref = torch.addmm(b, xf, wf.t())
    y, _ = ops.p_linear(x, w, b)
    e1 = _check(y, ref, 2e-2, 2e-2, "fwd bias")
    h, u = ops.p_linear(x, w, b, act=ops.ACT_QUICK_GELU, want_aux=True)
    _check(u, ref, 2e-2, 2e-2, "fwd pre-activation")
    _check(h, ref * torch.sigmoid(1.702 * ref), 2e-2, 2e-2, "fwd QuickGELU")
    del h, u, y
    res = _rnd(M, N, seed=4, dtype=torch.float32)
    y32, _ = ops.p_linear(x, w, b, residual=res, out_dtype=torch.float32)
    _check(y32, ref + res, 1e-2, 1e-2, "fwd fp32 residual")
    del y32, res, ref
    # ---- dgrad: dx = dy w (w read k-strided through the LDS transpose read), with act' and fused column sums
    dy = _rnd(M, N, seed=5)
    refd = dy.float() @ wf
    dx = ops.p_dgrad(dy, w, BF)
    e2 = _check(dx, refd, 2e-2, 2e-2 * math.sqrt(N / 768), "dgrad")
    del dx
    aux = _rnd(M, K, seed=6)
    du, cs = ops.p_dgrad(dy, w, BF, aux=aux, act=ops.ACT_QUICK_GELU, want_colsum=True)
    s = torch.sigmoid(1.702 * aux.float())
    refd *= s * (1 + 1.702 * aux.float() * (1 - s))
    _check(du, refd, 2e-2, 2e-2 * math.sqrt(N / 768), "dgrad*act'")
    # the column sums are taken from the fp32 values BEFORE the bf16 rounding of the stored output
    _check(cs, refd.double().sum(0).float(), 2e-3, 2e-3 * float(refd.abs().mean()) * math.sqrt(M), "fused colsum")
    del du, aux, refd, s
    # ---- wgrad: dw = dy^T x, contraction over the 50176 rows, split-K; fp32 output
    dw = ops.p_wgrad(dy, x)
    refw = dy.float().t() @ xf
    e3 = _check(dw, refw, 1e-2, 1e-2 * math.sqrt(M / 768), "wgrad split-K")
    dw2 = ops.p_wgrad(dy, x)
    assert torch.equal(dw, dw2), "wgrad must be bit-reproducible run to run (deterministic split-K reduction)"
    print(f"\n[bench GEMM N={N} K={K}] max err fwd {e1:.3e} dgrad {e2:.3e} wgrad {e3:.3e}")


def _run(spec_name, B, seed, dtype, flags, mode="t18", keep_grads=False):
    spec = synth.SPECS[spec_name]
    segclip_amd.set_compute_dtype(dtype)
    segclip_amd.set_cross_mode(mode)
    try:
        model, _ = synth.build_model(spec, flags, device=DEV)
        batch = synth.synthetic_batch(spec, B, seed=seed, device=DEV, with_seg=bool(flags))
        noise = synth.synthetic_noise(spec, B, seed=seed, device=DEV)
        items = noise_items(noise, flags) if flags else [("gumbel", noise["gumbel_main"])]
        with segclip_amd.noise_injection(items):
            loss = model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"],
                         image_seg=batch.get("image_seg"))
        loss.backward()
        torch.cuda.synchronize()
        out = dict(loss=float(loss.detach()), t2v=model.last_logits[0].float().cpu(),
                   hard_idx=model.last_mid_states["hard_idx"].cpu(),
                   gn={n: float(p.grad.double().norm()) for n, p in model.named_parameters() if p.grad is not None},
                   dim={n: p.dim() for n, p in model.named_parameters()})
        if keep_grads:
            out["grads"] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        if getattr(model, "last_mae", None) is not None:
            out["ids_restore"] = model.last_mae[1].cpu()
        del model, loss
        torch.cuda.empty_cache()
        return out
    finally:
        segclip_amd.set_compute_dtype(torch.float32)
        segclip_amd.set_cross_mode("t18")


def test_b256_bf16_against_exact_f32_mode():
    """BASELINE configs[1] size.  'intended' cross-attention mode: each sample depends only on itself, so the
    comparison is not dominated by argmax flips propagating through the t18 cross-sample mixing.
    Every parameter gradient of the bf16 run is compared with the exact-f32 run by DIRECTION (cosine) and norm.
    At random init and B=256 the contrastive gradients are sums of nearly cancelling terms over 50176 rows, so the
    bf16 rounding of the activations shows up as 10-30 % norm differences in the earliest blocks; the GEMM kernels
    themselves are held to fp32 matmul of identical operands in test_bench_gemm_shapes_fwd_dgrad_wgrad."""
    f = _run("vitb16", 256, 3, torch.float32, {}, "intended", keep_grads=True)
    b = _run("vitb16", 256, 3, torch.bfloat16, {}, "intended", keep_grads=True)
    dl = abs(f["loss"] - b["loss"])
    dlog = float((f["t2v"] - b["t2v"]).abs().max())
    agree = float((f["hard_idx"] == b["hard_idx"]).float().mean())
    assert set(f["gn"]) == set(b["gn"])
    ratios = {n: b["gn"][n] / f["gn"][n] for n in f["gn"] if f["gn"][n] > 1e-6}
    cos = {n: float((f["grads"][n].double() * b["grads"][n].double()).sum()) / (f["gn"][n] * b["gn"][n])
           for n in ratios}
    mats = [n for n in ratios if b["dim"][n] >= 2]      # GEMM weights / embeddings / centers
    vecs = [n for n in ratios if b["dim"][n] < 2]       # biases, LayerNorm
    wm = max(mats, key=lambda n: abs(math.log(ratios[n])))
    wv = max(vecs, key=lambda n: abs(math.log(ratios[n])))
    cm, cv = min(mats, key=lambda n: cos[n]), min(vecs, key=lambda n: cos[n])
    med = float(np.median(list(ratios.values())))
    medcos = float(np.median(list(cos.values())))
    print(f"\n[B=256 bf16 vs f32] loss {b['loss']:.5f} vs {f['loss']:.5f} (d {dl:.2e}); max |dlogit| {dlog:.4f}; "
          f"hard_idx agreement {agree:.4f}; grad-norm ratio median {med:.4f}, worst matrix {wm} {ratios[wm]:.4f}, "
          f"worst vector {wv} {ratios[wv]:.4f}; cosine median {medcos:.4f}, worst matrix {cm} {cos[cm]:.4f}, "
          f"worst vector {cv} {cos[cv]:.4f}")
    for n in sorted(mats, key=lambda n: cos[n])[:8]:
        print(f"    {n}: cos {cos[n]:.4f} norm ratio {ratios[n]:.4f}")
    assert abs(f["loss"] - math.log(256)) < 1.0          # random-init loss sits near ln B
    assert dl <= 0.003, dl                               # measured 7e-4
    assert dlog <= 0.6, dlog                             # measured 0.22 (logits are x14.3-scaled cosines)
    assert agree >= 0.98, agree                          # measured 0.994
    assert 0.98 <= med <= 1.03, med                      # measured 1.006
    # bounds ~3x the error measured on MI355X with the shipped default (bf16 residual-gradient chain ON; numbers in
    # profiles/r03_accuracy_b256.txt: matrices cosine >= 0.8986, norm ratio 0.988-1.127; vectors cosine >= 0.70 - the
    # two ln_x biases of the center stage - and ratio 0.75-1.31)
    for n in mats:
        assert 0.85 <= ratios[n] <= 1.30 and cos[n] >= 0.85, (n, ratios[n], cos[n])
    for n in vecs:
        assert 0.6 <= ratios[n] <= 1.6 and cos[n] >= 0.55, (n, ratios[n], cos[n])
    # the chain itself: the same step with the fp32 residual gradient must give the same picture (measured: every cosine
    # equal to 3 decimals; if the chain ever cost more than 2 points of cosine it would have to be switched off)
    segclip_amd.config.bf16_resgrad = False
    try:
        b2 = _run("vitb16", 256, 3, torch.bfloat16, {}, "intended", keep_grads=True)
    finally:
        segclip_amd.config.bf16_resgrad = True
    worst = 0.0
    for n in ratios:
        c2 = float((f["grads"][n].double() * b2["grads"][n].double()).sum()) / (f["gn"][n] * b2["gn"][n])
        worst = max(worst, c2 - cos[n])
    print(f"    bf16 residual-gradient chain: largest cosine loss against the fp32 residual gradient {worst:.4f}")
    assert worst <= 0.02, worst


def test_b256_bf16_t18_mode_the_bench_runs_against_exact_f32_mode():
    """The configuration bench.py times - "t18" cross-attention mode, bf16 - at B = 256 against the exact-f32 mode of the SAME
    cross-attention mode.  In "t18" a sample's centers attend to tokens of other samples (SURVEY finding 0.4), so one flipped
    8-way argmax reaches other samples' gradients and the agreement is looser than in "intended" mode (measured on MI355X,
    profiles/r04_accuracy_b256.txt: d loss 4.3e-4, max |dlogit| 0.32, hard_idx agreement 0.9945, matrices cosine >= 0.716 with
    norm ratio 0.85-1.01, vectors cosine >= 0.57); bounds = the measured numbers with ~25 % slack - this pins the benchmarked
    mode against silent regressions, the kernel arithmetic itself is pinned in "intended" mode above."""
    f = _run("vitb16", 256, 3, torch.float32, {}, "t18", keep_grads=True)
    b = _run("vitb16", 256, 3, torch.bfloat16, {}, "t18", keep_grads=True)
    dl = abs(f["loss"] - b["loss"])
    dlog = float((f["t2v"] - b["t2v"]).abs().max())
    agree = float((f["hard_idx"] == b["hard_idx"]).float().mean())
    assert set(f["gn"]) == set(b["gn"])
    ratios = {n: b["gn"][n] / f["gn"][n] for n in f["gn"] if f["gn"][n] > 1e-6}
    cos = {n: float((f["grads"][n].double() * b["grads"][n].double()).sum()) / (f["gn"][n] * b["gn"][n]) for n in ratios}
    mats = [n for n in ratios if b["dim"][n] >= 2]
    vecs = [n for n in ratios if b["dim"][n] < 2]
    cm, cv = min(mats, key=lambda n: cos[n]), min(vecs, key=lambda n: cos[n])
    print(f"\n[B=256 t18 bf16 vs f32] loss {b['loss']:.5f} vs {f['loss']:.5f} (d {dl:.2e}); max |dlogit| {dlog:.4f}; hard_idx "
          f"agreement {agree:.4f}; cosine median {float(np.median(list(cos.values()))):.4f}, worst matrix {cm} {cos[cm]:.4f} "
          f"(ratio {ratios[cm]:.3f}), worst vector {cv} {cos[cv]:.4f}")
    assert dl <= 2e-3 and dlog <= 0.8 and agree >= 0.985, (dl, dlog, agree)
    assert float(np.median(list(cos.values()))) >= 0.95
    # floors = the measured minima (profiles/r05_accuracy_b256.txt, "t18": matrices 0.9125 / ratio 0.986-1.097, vectors 0.868 /
    # ratio 0.91-1.12) minus ~10 % (the r4 floors 0.64 / 0.51 were far below what is measured)
    for n in mats:
        assert cos[n] >= 0.82 and 0.85 <= ratios[n] <= 1.20, (n, cos[n], ratios[n])
    for n in vecs:
        assert cos[n] >= 0.78 and 0.75 <= ratios[n] <= 1.30, (n, cos[n], ratios[n])


def test_b256_full_loss_bf16_against_exact_f32_mode():
    """BASELINE configs[3] size: full SegCLIP loss (contrastive + superpixel-KL + MAE) at B = 256, bf16 (shipped
    defaults) against the exact-f32 mode: loss, logits, hard assignment and MAE index maps, every parameter gradient
    by cosine and norm.  Measured on MI355X (profiles/r03_accuracy_b256.txt): d loss 2.6e-4, hard_idx agreement 0.9939,
    ids_restore identical, matrices cosine >= 0.915 / ratio 0.989-1.10, vectors cosine >= 0.913 / ratio 0.975-1.14."""
    f = _run("vitb16", 256, 3, torch.float32, FULL_FLAGS, "intended", keep_grads=True)
    b = _run("vitb16", 256, 3, torch.bfloat16, FULL_FLAGS, "intended", keep_grads=True)
    dl = abs(f["loss"] - b["loss"])
    agree = float((f["hard_idx"] == b["hard_idx"]).float().mean())
    assert set(f["gn"]) == set(b["gn"])
    ratios = {n: b["gn"][n] / f["gn"][n] for n in f["gn"] if f["gn"][n] > 1e-6}
    cos = {n: float((f["grads"][n].double() * b["grads"][n].double()).sum()) / (f["gn"][n] * b["gn"][n]) for n in ratios}
    cm = min(ratios, key=lambda n: cos[n])
    print(f"\n[B=256 full loss bf16 vs f32] loss {b['loss']:.5f} vs {f['loss']:.5f} (d {dl:.2e}); hard_idx agreement {agree:.4f}; "
          f"worst cosine {cm} {cos[cm]:.4f}; norm ratio {min(ratios.values()):.4f} .. {max(ratios.values()):.4f}")
    assert dl <= 1e-3, dl
    assert agree >= 0.98, agree
    assert torch.equal(f["ids_restore"], b["ids_restore"])      # the MAE shuffle is an integer function of the injected noise
    for n in ratios:
        lo, hi, cmin = (0.85, 1.30, 0.80) if b["dim"][n] >= 2 else (0.8, 1.4, 0.75)
        assert lo <= ratios[n] <= hi and cos[n] >= cmin, (n, ratios[n], cos[n])


@pytest.mark.parametrize("split", [False, True], ids=["exact", "f32_split"])
@pytest.mark.parametrize("flags", [{}, FULL_FLAGS], ids=["contrastive", "full_loss"])
def test_b256_exact_f32_against_cpu_oracle(flags, split):
    """The benchmarked SIZE against the oracle itself (the B = 256 tests above compare the HIP bf16 mode
    with the HIP f32 mode - a self-comparison).  HIP exact-f32 mode, "t18" cross-attention (what bench.py times), ViT-B/16,
    B = 256, contrastive-only (BASELINE configs[1]) and full loss (configs[3]) against oracle.segclip_forward on the same
    seeded inputs under torch.no_grad() (forward only: ~1 min of CPU): loss to 1e-3 (measured 3e-6), the MAE shuffle
    array-equal, the 8-way hard assignment of all 256 x 196 patches equal except for verified exact near-ties (see below),
    every logit to 1e-3 when no patch flipped."""
    from oracle import segclip_oracle as so
    from tests.helpers import model_param_shapes, oracle_params
    spec = synth.SPECS["vitb16"]
    B, seed = 256, 3
    segclip_amd.set_compute_dtype(torch.float32)
    segclip_amd.set_cross_mode("t18")
    segclip_amd.config.f32_split = split     # the same bounds for the Linear layers as bf16 x 3 products (config.f32_split)
    try:
        model, _ = synth.build_model(spec, flags, device=DEV)
        batch = synth.synthetic_batch(spec, B, seed=seed, device=DEV, with_seg=bool(flags))
        noise = synth.synthetic_noise(spec, B, seed=seed, device=DEV)
        items = noise_items(noise, flags) if flags else [("gumbel", noise["gumbel_main"])]
        with torch.no_grad(), segclip_amd.noise_injection(items):
            loss = model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"],
                         image_seg=batch.get("image_seg"))
        torch.cuda.synchronize()
        got = dict(loss=float(loss), t2v=model.last_logits[0].float().cpu(), v2t=model.last_logits[1].float().cpu(),
                   hard_idx=model.last_mid_states["hard_idx"].cpu().long())
        if getattr(model, "last_mae", None) is not None:
            got["ids_restore"] = model.last_mae[1].cpu().long()
            got["mae_hard_idx"] = model.last_mae[2]["hard_idx"].cpu().long()
        del model, loss
        torch.cuda.empty_cache()
    finally:
        segclip_amd.config.f32_split = False
        segclip_amd.set_compute_dtype(torch.float32)
        segclip_amd.set_cross_mode("t18")
    P = oracle_params(spec, model_param_shapes(spec, flags), requires_grad=False)
    with torch.no_grad():
        lo, aux = so.segclip_forward(synth.synthetic_batch(spec, B, seed=seed, with_seg=bool(flags)), P, spec,
                                     synth.synthetic_noise(spec, B, seed=seed), flags, cross_mode="t18")
    dl = abs(got["loss"] - float(lo))
    dt = (got["t2v"] - aux["t2v"]).abs()
    d1 = float(dt.max())
    d2 = float((got["v2t"] - aux["v2t"]).abs().max())
    mism = (got["hard_idx"] != aux["hard_idx"]).nonzero()
    agree = 1.0 - mism.shape[0] / got["hard_idx"].numel()
    print(f"\n[B=256 f32{' (f32_split)' if split else ''} vs oracle, {'full loss' if flags else 'contrastive'}] loss {got['loss']:.6f} vs {float(lo):.6f} (d {dl:.2e}); "
          f"max |d t2v| {d1:.2e}, |d v2t| {d2:.2e} ({float((dt <= 1e-3).float().mean()):.5f} of the logits within 1e-3); "
          f"hard_idx: {mism.shape[0]} of {got['hard_idx'].numel()} patches differ (agreement {agree:.6f})")
    assert dl <= 1e-3, dl
    # The 8-way argmax of 50176 patches is an integer function of fp32 sums taken in different orders (the GPU's MFMA fmaf chain,
    # the CPU's blocked GEMM): it is bit-exact at B <= 5 (test_model_gpu.py) and may flip on an exact NEAR-TIE at this size
    # (measured round 5: 1 patch of 50176).  Every differing patch must BE such a tie - the oracle's own soft assignment of the
    # two candidates' noisy logits equal to 2e-4 - and there may be at most a handful; anything else is a real error.
    assert mism.shape[0] <= 5, mism.shape[0]
    # the decision is argmax_c (logit_c + gumbel_c); soft = softmax_c(logit_c) (no noise), so logit differences = log-soft differences
    soft, gmb = aux["soft"].double(), synth.synthetic_noise(spec, B, seed=seed)["gumbel_main"].double()
    for b_, t_ in mism.tolist():
        co, ch = int(aux["hard_idx"][b_, t_]), int(got["hard_idx"][b_, t_])
        yo = float(soft[b_, co, t_].log() + gmb[b_, co, t_])
        yh = float(soft[b_, ch, t_].log() + gmb[b_, ch, t_])
        print(f"    near-tie at sample {b_} patch {t_}: oracle center {co} (logit + noise {yo:.7f}) vs HIP center {ch} ({yh:.7f})")
        assert abs(yo - yh) <= 2e-4 * max(1.0, abs(yo)), (b_, t_, yo, yh)      # measured: 4.9e-5 (fp32 noise after 10 blocks)
    if mism.shape[0] == 0:
        assert d1 <= 1e-3 and d2 <= 1e-3, (d1, d2)
    else:   # in "t18" mode a flipped patch reaches other samples' features through the cross-sample key mixing (finding 0.4)
        assert d1 <= 2e-2 and d2 <= 2e-2, (d1, d2)
        assert float((dt <= 1e-3).float().mean()) >= 0.5
    if flags:
        assert torch.equal(got["ids_restore"], aux["ids_restore"].long())      # the MAE shuffle: integer function of the noise
        mm = (got["mae_hard_idx"] != aux["mae_hard_idx"]).nonzero()
        assert mm.shape[0] <= 5, mm.shape[0]


def test_b32_gradients_against_cpu_oracle():
    """GRADIENTS above B = 5 against the oracle (VERDICT r5 weak #1c: the B = 256 tests hold the forward to the oracle and the
    gradients only to the HIP f32 mode).  ViT-B/16, B = 32, full loss, "t18" mode: 6272 vision token rows = 24 full 256-row tiles +
    a 128-row remainder, i.e. the multi-tile forward / data-gradient kernels with their half-tile rows, the split-K and grouped weight
    gradients and the streaming attention backward that the B <= 5 tests never reach.  (i) HIP exact-f32: loss, logits, every index
    map and EVERY parameter gradient against `oracle.segclip_forward(...).backward()` on the same seeded inputs, bound as at B = 5
    (max |d| <= 5e-3 max |ref|); (ii) the bf16 mode of the same batch against those ORACLE gradients by cosine and norm ratio."""
    from oracle import segclip_oracle as so
    from tests.helpers import model_param_shapes, oracle_params
    spec = synth.SPECS["vitb16"]
    B, seed = 32, 5
    f = _run("vitb16", B, seed, torch.float32, FULL_FLAGS, "t18", keep_grads=True)
    P = oracle_params(spec, model_param_shapes(spec, FULL_FLAGS))
    lo, aux = so.segclip_forward(synth.synthetic_batch(spec, B, seed=seed, with_seg=True), P, spec,
                                 synth.synthetic_noise(spec, B, seed=seed), FULL_FLAGS, cross_mode="t18")
    lo.backward()
    ref = {n: P[n].grad for n in P if P[n].grad is not None}
    lo_f = float(lo.detach())
    assert abs(f["loss"] - lo_f) <= 1e-3, (f["loss"], lo_f)
    assert torch.equal(f["hard_idx"].long(), aux["hard_idx"]), "8-way hard assignment differs from the oracle at B = 32"
    assert torch.equal(f["ids_restore"].long(), aux["ids_restore"].long())
    assert float((f["t2v"] - aux["t2v"].detach()).abs().max()) <= 1e-3
    worst, wn = 0.0, None
    for n, g in f["grads"].items():
        assert n in ref, n
        r = ref[n]
        err = float((g.cpu() - r).abs().max())
        rel = err / (float(r.abs().max()) + 1e-12)
        if rel > worst:
            worst, wn = rel, n
        assert err <= 5e-3 * float(r.abs().max()) + 1e-6, (n, err, float(r.abs().max()))
    assert set(ref) <= set(f["grads"]) | {n for n in ref if float(ref[n].abs().max()) == 0.0}
    print(f"\n[B=32 f32 vs oracle, full loss] loss {f['loss']:.6f} vs {float(lo):.6f}; {len(f['grads'])} parameter gradients, "
          f"worst max|d| / max|ref| {worst:.2e} ({wn})")
    b = _run("vitb16", B, seed, torch.bfloat16, FULL_FLAGS, "t18", keep_grads=True)
    cos, rat = {}, {}
    for n, g in b["grads"].items():
        r = ref[n].double()
        rn = float(r.norm())
        if rn <= 1e-9:
            continue
        gd = g.cpu().double()
        cos[n] = float((gd * r).sum()) / (float(gd.norm()) * rn + 1e-30)
        rat[n] = float(gd.norm()) / rn
    mats = [n for n in cos if b["dim"][n] >= 2]
    vecs = [n for n in cos if b["dim"][n] < 2]
    cm, cv = min(mats, key=lambda n: cos[n]), min(vecs, key=lambda n: cos[n])
    agree = float((b["hard_idx"].long() == aux["hard_idx"]).float().mean())
    print(f"[B=32 bf16 vs ORACLE gradients] loss {b['loss']:.5f} vs {float(lo):.5f}; hard_idx agreement {agree:.4f}; cosine median "
          f"{float(np.median(list(cos.values()))):.4f}, worst matrix {cm} {cos[cm]:.4f} (ratio {rat[cm]:.3f}), worst vector {cv} "
          f"{cos[cv]:.4f} (ratio {rat[cv]:.3f}); norm ratio {min(rat.values()):.3f} .. {max(rat.values()):.3f}")
    assert abs(b["loss"] - float(lo)) <= 5e-3
    assert agree >= 0.98, agree
    assert float(np.median(list(cos.values()))) >= 0.97
    # floors: the minima measured on MI355X (round 6: worst matrix layers0.2.attn.out_proj.weight cosine 0.8752, worst vector
    # layers0.1.ln_1.bias 0.8584, norm ratio 0.932 .. 1.214, hard_idx agreement 0.9936, f32 gradients 4.9e-5 of max |ref|) minus ~10 %
    for n in mats:
        assert cos[n] >= 0.79 and 0.84 <= rat[n] <= 1.33, (n, cos[n], rat[n])
    for n in vecs:
        assert cos[n] >= 0.77 and 0.84 <= rat[n] <= 1.33, (n, cos[n], rat[n])


def test_b4_bf16_grad_norms_against_reference_golden():
    g = load_golden("vitb16_b4_t18.npz")
    b = _run("vitb16", int(g["B"]), int(g["seed"]), torch.bfloat16, FULL_FLAGS, "t18")
    dl = abs(b["loss"] - float(g["loss"]))
    dlog = float((b["t2v"] - torch.from_numpy(g["t2v"])).abs().max())
    rat = []
    for n, ref in zip(g["grad_names"].tolist(), g["grad_norms"]):
        if ref > 1e-6:
            rat.append((n, b["gn"][n] / ref))
    worst = max(rat, key=lambda kv: abs(math.log(max(kv[1], 1e-9))))
    med = float(np.median([r for _, r in rat]))
    print(f"\n[B=4 bf16 vs reference] d loss {dl:.2e}; max |dlogit| {dlog:.4f}; grad-norm ratio median {med:.4f}, "
          f"worst {worst[0]} {worst[1]:.4f}")
    assert dl <= 0.02 and dlog <= 0.1
    assert 0.97 <= med <= 1.03, med
    for n, r in rat:
        assert 0.6 <= r <= 1.6, (n, r)
