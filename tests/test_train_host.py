"""CPU: host logic of the training-step tail (segclip_amd/train.py, modules/optimization_adamw.py) against what the
REAL reference driver did (tests/golden/train_tiny_t18.npz), and the host-side contract of the optimizer ABI."""
import ctypes
import types

import pytest
import torch

from segclip_amd import _lib, synth, train
from segclip_amd.modules.optimization_adamw import SCHEDULES, AdaptAdamW
from tests.helpers import FULL_FLAGS, load_golden


def golden_args(g, **over):
    d = {k[5:]: float(g[k]) if g[k].dtype.kind == "f" else int(g[k]) for k in g.files if k.startswith("arg::")}
    d.update(pretrained_clip_name="ViT-B/16")
    d.update(over)
    return types.SimpleNamespace(**d)


def test_freeze_and_param_groups_match_reference_driver():
    g = load_golden("train_tiny_t18.npz")
    model, _ = synth.build_model(synth.SPECS["tiny"], FULL_FLAGS, device="cpu", closed_form=False)
    args = golden_args(g)
    frozen = train.freeze_parameters(args, model)
    assert frozen + ["vis_mae_decoder.decoder_pos_embed"] == g["frozen"].tolist()
    assert [n for n, p in model.named_parameters() if not p.requires_grad] == g["frozen"].tolist()
    opt, sched, wrapped, scaler = train.prep_optimizer(args, model, int(g["t_total"]), shadow_bf16=False)
    assert sched is None and wrapped is model and not scaler.is_enabled()
    pname = {id(p): n for n, p in model.named_parameters()}
    assert len(opt.param_groups) == 8
    for gi, grp in enumerate(opt.param_groups):
        assert [pname[id(p)] for p in grp["params"]] == [n for n in g[f"group{gi}"].tolist() if n], gi
        assert grp["weight_decay"] == (args.weight_decay if gi % 2 == 0 else 0.0)
        assert grp["lr"] == (args.lower_lr if gi < 4 else args.lr)
        assert grp["schedule"] == "warmup_cosine" and grp["t_total"] == int(g["t_total"])
        assert (grp["b1"], grp["b2"], grp["e"], grp["warmup"]) == (args.opt_b1, args.opt_b2, args.eps, args.warmup_proportion)


def test_freeze_thresholds():
    model, _ = synth.build_model(synth.SPECS["tiny"], FULL_FLAGS, device="cpu", closed_form=False)
    args = types.SimpleNamespace(freeze_layer_num=1, freeze_text_layer_num=1, first_stage_layer=1, pretrained_clip_name="x")
    frozen = set(train.freeze_parameters(args, model))
    assert any(n.startswith("clip.visual.transformer.layers0.0.") for n in frozen)
    assert not any(n.startswith("clip.visual.transformer.layers0.1.") for n in frozen)
    assert any(n.startswith("clip.transformer.resblocks.0.") for n in frozen)
    assert "clip.token_embedding.weight" in frozen and "clip.visual.proj" not in frozen
    assert not any("semantic_layer" in n or "layers2." in n for n in frozen)
    model2, _ = synth.build_model(synth.SPECS["tiny"], FULL_FLAGS, device="cpu", closed_form=False)
    none = train.freeze_parameters(types.SimpleNamespace(freeze_layer_num=-1, pretrained_clip_name="x"), model2)
    assert none == []


def test_schedules_and_constructor_errors():
    g = load_golden("adamw_steps.npz")
    # lrs of the golden run: group0 cosine (lr 1e-2, warmup .25, t_total 8, lr_start .1, lr_end .05)
    assert abs(1e-2 * SCHEDULES["warmup_cosine"](1 / 8, 0.25, 0.1, 0.05) - 0.0055) < 1e-15
    assert abs(1e-2 * SCHEDULES["warmup_cosine"](3 / 8, 0.25, 0.1, 0.05) - float(g["lrs"][2][2])) < 1e-15
    assert SCHEDULES["warmup_linear"](0.5, 0.25) == pytest.approx((0.5 - 1) / (0.25 - 1))
    assert SCHEDULES["warmup_constant"](0.5, 0.25) == 1.0
    p = torch.nn.Parameter(torch.zeros(3))
    for kw, msg in ((dict(lr=-1.0), "learning rate"), (dict(lr=1e-3, schedule="nope"), "schedule"),
                    (dict(lr=1e-3, warmup=1.5), "warmup"), (dict(lr=1e-3, b1=1.0), "b1"), (dict(lr=1e-3, b2=-0.1), "b2"),
                    (dict(lr=1e-3, e=-1.0), "epsilon"), (dict(lr=1e-3, lr_start=1.0), "lr_start"),
                    (dict(lr=1e-3, lr_end=2.0), "lr_end")):
        with pytest.raises(ValueError, match=msg):
            AdaptAdamW([p], **kw)
    opt = AdaptAdamW([p], lr=1e-3)
    assert opt.get_lr() == []          # no gradient anywhere
    p.grad = torch.zeros(3)
    assert opt.get_lr() == [0]         # reference behaviour before the first step
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step()


def test_optimizer_abi_host_contract():
    lib = _lib.load()
    assert ctypes.sizeof(_lib.TrainCtrl) == 32
    assert ctypes.sizeof(_lib.AdamWGroup) == 80
    assert ctypes.sizeof(_lib.AdamWTensor) == 56
    sizes = (ctypes.c_int64 * 4)(1, 16384, 16385, 0)
    assert lib.segclip_grad_sqnorm_ws_bytes(ctypes.cast(sizes, ctypes.c_void_p), 4) == 4 * 4
    groups = (_lib.AdamWGroup * 17)()
    tens = (_lib.AdamWTensor * 1)()
    rc = lib.segclip_adamw_step(ctypes.cast(tens, ctypes.c_void_p), 1, ctypes.cast(groups, ctypes.c_void_p), 17,
                                None, None, 0, None)
    assert rc == -1 and b"16 param groups" in lib.segclip_last_error_string()
    groups[0].schedule = 7
    rc = lib.segclip_adamw_step(ctypes.cast(tens, ctypes.c_void_p), 1, ctypes.cast(groups, ctypes.c_void_p), 1,
                                None, None, 0, None)
    assert rc == -1 and b"schedule" in lib.segclip_last_error_string()
    groups[0].schedule, groups[0].b1, groups[0].b2, groups[0].lr = 0, 0.9, 0.98, 1e-3
    tens[0].n, tens[0].param = 8, 16  # null grad / state pointers -> rejected before any launch
    rc = lib.segclip_adamw_step(ctypes.cast(tens, ctypes.c_void_p), 1, ctypes.cast(groups, ctypes.c_void_p), 1,
                                None, None, 0, None)
    assert rc == -1 and b"null pointer" in lib.segclip_last_error_string()
    assert lib.segclip_train_step_finish(None, None, None, 0.0, None) == -1
