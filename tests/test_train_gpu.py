"""GPU: the training-step tail (fused multi-tensor AdaptAdamW + gradient clip + NaN skip, segclip_amd/train.py)
through the C-ABI, against (a) vectors produced by the REAL reference's AdaptAdamW / train_epoch
(tests/golden/adamw_steps.npz, train_tiny_t18.npz) and (b) the CPU oracle on seeded inputs.
Tolerances: optimizer arithmetic rtol 1e-5 (fp32 FMA contraction differs between hosts), trajectory loss 1e-3."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import segclip_amd  # noqa: E402
from oracle import train_oracle as to  # noqa: E402
from segclip_amd import synth, train  # noqa: E402
from segclip_amd.modules.optimization_adamw import AdaptAdamW  # noqa: E402
from tests.helpers import FULL_FLAGS, load_golden, noise_items  # noqa: E402
from tests.test_train_host import golden_args  # noqa: E402
from tests.test_train_oracle_golden import ADAMW_DEFAULTS, ADAMW_GROUPS, adamw_groups  # noqa: E402

DEV = "cuda"


def make_optimizer(params, **extra):
    groups = []
    for g in ADAMW_GROUPS:
        d = {k: v for k, v in g.items() if k != "idx"}
        d["params"] = [params[i] for i in g["idx"]]
        groups.append(d)
    return AdaptAdamW(groups, lr=5e-2, weight_decay=0.05, max_grad_norm=1.0, schedule="warmup_cosine", t_total=8,
                      **ADAMW_DEFAULTS, **extra)


def test_fused_adamw_matches_reference_optimizer():
    g = load_golden("adamw_steps.npz")
    n = int(g["n_params"])
    params = [torch.nn.Parameter(torch.from_numpy(g[f"p{i}_init"]).to(DEV)) for i in range(n)]
    opt = make_optimizer(params)
    for step in range(4):
        for i, p in enumerate(params):
            k = f"g{i}_s{step}"
            p.grad = torch.from_numpy(g[k]).to(DEV) if k in g.files else None
        opt.step()
        ref_lrs = g["lrs"][step]
        np.testing.assert_allclose(sorted(set(opt.get_lr())), ref_lrs[~np.isnan(ref_lrs)], rtol=1e-12)
        for i, p in enumerate(params):
            np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"p{i}_s{step}"], rtol=1e-5, atol=1e-7,
                                       err_msg=f"p{i} step {step}")
    for i, p in enumerate(params):
        st = opt.state[p]
        assert st["step"] == int(g[f"step{i}"])
        np.testing.assert_allclose(st["exp_avg"].cpu().numpy(), g[f"m{i}"], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(st["exp_avg_sq"].cpu().numpy(), g[f"v{i}"], rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize("max_norm", [1.0, 1e4, 0.0])
def test_grad_norm_clip_and_step_vs_oracle(max_norm):
    """Ragged sizes around the 16384-element chunk and the float4 tail; > 32 tensors (several launches)."""
    gen = torch.Generator().manual_seed(3)
    sizes = [1, 3, 4, 5, 255, 16383, 16384, 16385, 40000, 7, 65536 + 3] + [17 + i for i in range(30)]
    params = [torch.nn.Parameter(torch.randn(s, generator=gen).to(DEV)) for s in sizes]
    grads = [torch.randn(s, generator=gen) * (0.05 if i % 2 else 3.0) for i, s in enumerate(sizes)]
    hp = dict(lr=1e-2, weight_decay=0.05, schedule="warmup_cosine", warmup=0.2, t_total=10, b1=0.9, b2=0.98, e=1e-6,
              lr_start=0.0, lr_end=0.0)
    opt = AdaptAdamW(params, lr=hp["lr"], warmup=hp["warmup"], t_total=hp["t_total"], schedule=hp["schedule"],
                     b1=hp["b1"], b2=hp["b2"], e=hp["e"], weight_decay=hp["weight_decay"], shadow_bf16=True)
    tail = train.TrainTail(torch.nn.ParameterList(params), opt, clip_grad=max_norm)
    names = [f"p{i}" for i in range(len(sizes))]
    ref_p = {n: p.detach().cpu().numpy().copy() for n, p in zip(names, params)}
    ref_opt = to.AdamWState([dict(hp, names=names)])
    for step in range(3):
        ref_g = {n: (g * (step + 1)).numpy().copy() for n, g in zip(names, grads)}
        for p, g in zip(params, grads):
            p.grad = (g * (step + 1)).to(DEV)
        loss = torch.tensor(0.5 + step, device=DEV)
        tail.run(loss)
        total = to.clip_grad_norm(list(ref_g.values()), max_norm) if max_norm > 0 else \
            math.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in ref_g.values()))
        ref_opt.step(ref_p, ref_g)
        st = tail.read()
        assert st["grad_norm"] == pytest.approx(total, rel=1e-5)
        assert st["clip_coef"] == pytest.approx(min(1.0, max_norm / (total + 1e-6)) if max_norm > 0 else 1.0, rel=1e-5)
        assert st["steps"] == step + 1 and st["nan_skips"] == 0 and st["last_loss"] == 0.5 + step
        for n, p in zip(names, params):
            assert p.grad is None
            np.testing.assert_allclose(p.detach().cpu().numpy(), ref_p[n], rtol=2e-5, atol=2e-7, err_msg=f"{n} step {step}")
    assert st["loss_sum"] == pytest.approx(0.5 + 1.5 + 2.5)
    for p in params:  # 1-D tensors get no shadow; check the mechanism on a 2-D one below
        assert not hasattr(p, "_segclip_shadow")


def test_nan_loss_skips_on_device_and_shadow_tracks_param():
    gen = torch.Generator().manual_seed(5)
    w = torch.nn.Parameter(torch.randn(37, 129, generator=gen).to(DEV))
    b = torch.nn.Parameter(torch.randn(129, generator=gen).to(DEV))
    ls = torch.nn.Parameter(torch.tensor(5.0, device=DEV))

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.clip = torch.nn.Module()
            self.clip.logit_scale = ls
            self.w, self.b = w, b

    m = M()
    hp = dict(lr=1e-2, weight_decay=0.05, schedule="warmup_cosine", warmup=0.2, t_total=10, b1=0.9, b2=0.98, e=1e-6)
    opt = AdaptAdamW([w, b, ls], lr=hp["lr"], warmup=hp["warmup"], t_total=hp["t_total"], schedule=hp["schedule"],
                     b1=hp["b1"], b2=hp["b2"], e=hp["e"], weight_decay=hp["weight_decay"], shadow_bf16=True)
    tail = train.TrainTail(m, opt, clip_grad=1.0)
    names = ["w", "b", "ls"]
    ref_p = {"w": w.detach().cpu().numpy().copy(), "b": b.detach().cpu().numpy().copy(), "ls": ls.detach().cpu().numpy().copy()}
    ref_opt = to.AdamWState([dict(hp, names=names, lr_start=0.0, lr_end=0.0)])
    losses = [1.0, float("nan"), 2.0, float("nan"), 0.25]
    for it, lv in enumerate(losses):
        gs = {n: torch.randn(ref_p[n].shape, generator=gen) for n in names}
        w.grad, b.grad, ls.grad = gs["w"].to(DEV), gs["b"].to(DEV), gs["ls"].to(DEV)
        before = w.detach().clone()
        tail.run(torch.tensor(lv, device=DEV))
        ref_g = {n: gs[n].numpy().copy() for n in names}
        to.clip_grad_norm(list(ref_g.values()), 1.0)
        if math.isnan(lv):
            assert torch.equal(w.detach(), before)
        else:
            ref_opt.step(ref_p, ref_g)
            np.minimum(ref_p["ls"], np.float32(math.log(100)), out=ref_p["ls"])
        for n, p in zip(names, (w, b, ls)):
            np.testing.assert_allclose(p.detach().cpu().numpy(), ref_p[n], rtol=2e-5, atol=2e-7, err_msg=f"{n} it {it}")
        sh = w._segclip_shadow
        assert sh[1] == w._version and torch.equal(sh[0], w.detach().to(torch.bfloat16))
    st = tail.read()
    assert st["nan_skips"] == 2 and st["steps"] == 5 and st["loss_sum"] == pytest.approx(3.25)
    assert math.isclose(ls.item(), min(ls.item(), math.log(100)))
    assert opt.state[w]["step"] == 5 and opt.effective_step(w) == 3 == ref_opt.state["w"]["step"]
    sd = opt.state_dict()                                  # checkpoints carry the reference's meaning of 'step'
    assert all(v["step"] == 3 for v in sd["state"].values()) and set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    opt.load_state_dict(sd)
    assert opt.state[w]["step"] == 5 and opt.effective_step(w) == 3
    # logit_scale clamp
    with torch.no_grad():
        ls.fill_(9.0)
    w.grad, b.grad, ls.grad = torch.zeros_like(w), torch.zeros_like(b), torch.zeros_like(ls)
    tail.run(torch.tensor(float("nan"), device=DEV))
    assert ls.item() == pytest.approx(math.log(100), rel=1e-7)
    # a torch-side write to the parameter invalidates the shadow (version check in ops.wcast)
    from segclip_amd import ops
    assert ops.wcast(w, torch.bfloat16) is w._segclip_shadow[0]
    with torch.no_grad():
        w.mul_(2.0)
    fresh = ops.wcast(w, torch.bfloat16)
    assert w._segclip_shadow[1] == w._version and torch.equal(fresh, w.detach().to(torch.bfloat16))


def run_trajectory(dtype):
    g = load_golden("train_tiny_t18.npz")
    spec = synth.SPECS["tiny"]
    segclip_amd.set_compute_dtype(dtype)
    try:
        model, _ = synth.build_model(spec, FULL_FLAGS, device=DEV)
        args = golden_args(g, n_display=1, epochs=1)
        frozen = train.freeze_parameters(args, model)
        optimizer, scheduler, model, scaler = train.prep_optimizer(args, model, int(g["t_total"]))
        B, steps = int(g["batch"]), int(g["steps"])
        loader, inject = [], []
        for s in range(steps):
            b = synth.synthetic_batch(spec, B, seed=100 + s)
            nz = synth.synthetic_noise(spec, B, seed=100 + s, device=DEV)
            loader.append((b["input_ids"], b["input_mask"], b["segment_ids"], b["image"], torch.zeros(B, 4), b["image_seg"]))
            inject += noise_items(nz, FULL_FLAGS)
        losses = []
        hook = model.register_forward_hook(lambda m, i, o: losses.append(o.detach()))
        tail = train.TrainTail(model, optimizer, args.clip_grad)
        with segclip_amd.noise_injection(inject):
            total, gstep = train.train_epoch(0, args, model, loader, torch.device(DEV), 1, optimizer, scheduler, 0, scaler,
                                             tail=tail)
        hook.remove()
        torch.cuda.synchronize()
    finally:
        segclip_amd.set_compute_dtype(torch.float32)
        segclip_amd.config.trust_weight_shadows = False
    return g, model, optimizer, tail, [float(l) for l in losses], total, gstep, frozen


def test_train_trajectory_f32_matches_reference_driver():
    g, model, optimizer, tail, losses, total, gstep, frozen = run_trajectory(torch.float32)
    assert gstep == int(g["global_step"]) == 3
    np.testing.assert_allclose(losses, g["losses"], rtol=0, atol=1e-3)
    assert abs(total - float(g["total_loss"])) <= 1e-3
    st = tail.read()
    assert st["grad_norm"] == pytest.approx(float(g["grad_norms"][-1]), rel=2e-3)
    ref_lrs = g["lrs"][-1]
    np.testing.assert_allclose(sorted(set(optimizer.get_lr(with_grad_only=False))), ref_lrs[~np.isnan(ref_lrs)], rtol=1e-12)
    P = dict(model.named_parameters())
    for n, s, a in zip(g["param_names"].tolist(), g["param_sum"], g["param_abssum"]):
        t = P[n].detach().double()
        assert abs(float(t.sum()) - s) <= 1e-3 * max(1.0, a), n
        assert abs(float(t.abs().sum()) - a) <= 1e-3 * max(1.0, a), n
    for k in g.files:
        if k.startswith("final::"):
            np.testing.assert_allclose(P[k[7:]].detach().cpu().numpy(), g[k], rtol=2e-2, atol=2e-4, err_msg=k)
    for n in g["frozen"].tolist():
        assert P[n].grad is None and (P[n] not in optimizer.state or len(optimizer.state[P[n]]) == 0), n


def test_train_trajectory_bf16_tracks_reference():
    g, model, optimizer, tail, losses, total, gstep, frozen = run_trajectory(torch.bfloat16)
    np.testing.assert_allclose(losses, g["losses"], rtol=0, atol=5e-2)
    assert losses[2] < losses[0]
    n_shadow = 0
    for n, p in model.named_parameters():
        sh = getattr(p, "_segclip_shadow", None)
        if sh is not None:
            n_shadow += 1
            assert sh[1] == p._version and torch.equal(sh[0], p.detach().to(torch.bfloat16)), n
    assert n_shadow > 20


def test_checkpoint_round_trip_resumes_identically(tmp_path):
    """save_model / load_model (main_task_align.py:258-290 file names and dictionary keys): 2 iterations, checkpoint,
    fresh model + optimizer from the files, third iteration == the uninterrupted third iteration."""
    from segclip_amd.modules.module_clip import CLIP
    g = load_golden("train_tiny_t18.npz")
    spec = synth.SPECS["tiny"]
    B = int(g["batch"])

    def loader_and_noise(steps):
        loader, inject = [], []
        for s in steps:
            b = synth.synthetic_batch(spec, B, seed=100 + s)
            nz = synth.synthetic_noise(spec, B, seed=100 + s, device=DEV)
            loader.append((b["input_ids"], b["input_mask"], b["segment_ids"], b["image"], torch.zeros(B, 4), b["image_seg"]))
            inject += noise_items(nz, FULL_FLAGS)
        return loader, inject

    def fresh():
        model, margs = synth.build_model(spec, FULL_FLAGS, device=DEV)
        args = golden_args(g, n_display=1000, epochs=1, output_dir=str(tmp_path), cache_dir=None, **vars(margs))
        train.freeze_parameters(args, model)
        opt, sched, model, scaler = train.prep_optimizer(args, model, int(g["t_total"]), shadow_bf16=False)
        return model, args, opt, scaler

    def run(model, args, opt, scaler, steps, gstep):
        loader, inject = loader_and_noise(steps)
        with segclip_amd.noise_injection(inject):
            return train.train_epoch(0, args, model, loader, torch.device(DEV), 1, opt, None, gstep, scaler)

    try:
        model, args, opt, scaler = fresh()
        run(model, args, opt, scaler, [0, 1, 2], 0)
        want = {n: p.detach().clone() for n, p in model.named_parameters()}

        model, args, opt, scaler = fresh()
        loss, gstep = run(model, args, opt, scaler, [0, 1], 0)
        f = train.save_model(0, args, model, opt, loss, scaler)
        assert f.endswith("pytorch_model.bin.0") and (tmp_path / "pytorch_opt.bin.0").exists()
        orig = CLIP.get_config
        CLIP.get_config = staticmethod(lambda pretrained_clip_name="ViT-B/16": synth.synthetic_clip_state_dict(spec))
        try:
            model2 = train.load_model(0, args, 1, torch.device(DEV))
            assert train.load_model(7, args, 1, torch.device(DEV)) is None
        finally:
            CLIP.get_config = orig
        model2.train()
        train.freeze_parameters(args, model2)
        opt2, _, model2, scaler2 = train.prep_optimizer(args, model2, int(g["t_total"]), shadow_bf16=False)
        ck = torch.load(tmp_path / "pytorch_opt.bin.0", map_location=DEV)
        assert set(ck) == {"epoch", "optimizer_state_dict", "loss", "scaler"}
        opt2.load_state_dict(ck["optimizer_state_dict"])
        run(model2, args, opt2, scaler2, [2], gstep)
        for n, p in model2.named_parameters():
            assert torch.allclose(p.detach(), want[n], rtol=1e-6, atol=1e-7), n
    finally:
        segclip_amd.config.trust_weight_shadows = False


def test_fused_step_without_shadows_invalidates_cached_bf16_copies():
    """The fused step writes parameters through raw pointers; a bf16 copy cached by ops.wcast must be re-made."""
    from segclip_amd import ops
    w = torch.nn.Parameter(torch.randn(48, 40, generator=torch.Generator().manual_seed(1)).to(DEV))
    opt = AdaptAdamW([w], lr=1e-1, weight_decay=0.0, shadow_bf16=False)
    before = ops.wcast(w, torch.bfloat16).clone()
    w.grad = torch.ones_like(w)
    opt.step()
    after = ops.wcast(w, torch.bfloat16)
    assert torch.equal(after, w.detach().to(torch.bfloat16)) and not torch.equal(after, before)
