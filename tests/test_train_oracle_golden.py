"""CPU: the training-tail oracle (oracle/train_oracle.py) reproduces what the REAL reference's AdaptAdamW,
prep_optimizer, freeze block and train_epoch produced (tests/golden/make_golden_train.py)."""
import numpy as np
import torch

from oracle import train_oracle as to
from segclip_amd import synth
from tests.helpers import FULL_FLAGS, load_golden, model_param_shapes, oracle_params

ADAMW_GROUPS = [  # the param groups of make_golden_train.golden_adamw, defaults filled in
    dict(idx=[0, 1], lr=1e-2, weight_decay=0.05, schedule="warmup_cosine", t_total=8),
    dict(idx=[2, 3], lr=3e-3, weight_decay=0.0, schedule="warmup_linear", t_total=8),
    dict(idx=[4], lr=5e-2, weight_decay=0.2, schedule="warmup_constant", t_total=-1),
]
ADAMW_DEFAULTS = dict(warmup=0.25, b1=0.9, b2=0.98, e=1e-6, lr_start=0.1, lr_end=0.05)


def adamw_groups():
    return [dict(ADAMW_DEFAULTS, **{k: v for k, v in g.items() if k != "idx"}, names=[f"p{i}" for i in g["idx"]])
            for g in ADAMW_GROUPS]


def train_args(g):
    return {k[5:]: float(g[k]) if g[k].dtype.kind == "f" else int(g[k]) for k in g.files if k.startswith("arg::")}


def test_adamw_oracle_matches_reference_optimizer():
    g = load_golden("adamw_steps.npz")
    n = int(g["n_params"])
    params = {f"p{i}": g[f"p{i}_init"].copy() for i in range(n)}
    opt = to.AdamWState(adamw_groups())
    for step in range(4):
        grads = {f"p{i}": g[f"g{i}_s{step}"].copy() for i in range(n) if f"g{i}_s{step}" in g.files}
        lrs = sorted(set(opt.step(params, grads)))
        ref_lrs = g["lrs"][step]
        np.testing.assert_allclose(lrs, ref_lrs[~np.isnan(ref_lrs)], rtol=1e-12)
        for i in range(n):
            np.testing.assert_allclose(params[f"p{i}"], g[f"p{i}_s{step}"], rtol=1e-5, atol=1e-7, err_msg=f"p{i} step {step}")
    for i in range(n):
        st = opt.state[f"p{i}"]
        assert st["step"] == int(g[f"step{i}"])
        np.testing.assert_allclose(st["m"], g[f"m{i}"], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(st["v"], g[f"v{i}"], rtol=1e-5, atol=1e-12)


def test_group_routing_and_freeze_match_reference():
    g = load_golden("train_tiny_t18.npz")
    names = g["param_names"].tolist()
    args = train_args(g)
    groups = to.build_groups(names, args, int(g["t_total"]))
    for gi in range(8):
        ref = [n for n in g[f"group{gi}"].tolist() if n]
        assert groups[gi]["names"] == ref, gi
    built_frozen = ["vis_mae_decoder.decoder_pos_embed"]  # requires_grad=False by construction (module_mae.py)
    assert to.frozen_names(names, dict(args, pretrained_clip_name="ViT-B/16")) + built_frozen == g["frozen"].tolist()
    # the module mirror exposes the same parameter names in the same order
    model, _ = synth.build_model(synth.SPECS["tiny"], FULL_FLAGS, device="cpu", closed_form=False)
    assert [n for n, _ in model.named_parameters()] == names


def test_train_trajectory_oracle_matches_reference():
    g = load_golden("train_tiny_t18.npz")
    spec = synth.SPECS["tiny"]
    args = dict(train_args(g), pretrained_clip_name="ViT-B/16")
    P = oracle_params(spec, model_param_shapes(spec, FULL_FLAGS))
    for n in to.frozen_names(g["param_names"].tolist(), args):
        P[n].requires_grad_(False)
    B, steps = int(g["batch"]), int(g["steps"])
    batches = [synth.synthetic_batch(spec, B, seed=100 + s) for s in range(steps)]
    noises = [synth.synthetic_noise(spec, B, seed=100 + s) for s in range(steps)]
    # restrict the optimizer to named_parameters (state-dict buffers are not parameters)
    keep = set(g["param_names"].tolist())
    Pp = {k: v for k, v in P.items()}
    r = to.train_trajectory(spec, FULL_FLAGS, Pp, batches, noises, args, int(g["t_total"]), param_names=keep)
    np.testing.assert_allclose(r["losses"], g["losses"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(r["grad_norms"], g["grad_norms"], rtol=2e-4)
    for got, ref in zip(r["lrs"], g["lrs"]):
        np.testing.assert_allclose(got, ref[~np.isnan(ref)], rtol=1e-12)
    for n, s, a in zip(g["param_names"].tolist(), g["param_sum"], g["param_abssum"]):
        t = P[n].detach().double()
        assert abs(float(t.sum()) - s) <= 2e-4 * max(1.0, a), n
        assert abs(float(t.abs().sum()) - a) <= 2e-4 * max(1.0, a), n
    for k in g.files:
        if k.startswith("final::"):
            np.testing.assert_allclose(P[k[7:]].detach().numpy(), g[k], rtol=1e-3, atol=2e-5, err_msg=k)
