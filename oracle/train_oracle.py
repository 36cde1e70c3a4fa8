"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- CPU restatement of the training-step tail that
surrounds the hot path.  numpy float32 arithmetic in the reference's op order.

Restates
  * modules/optimization_adamw.py:26-45   warmup_cosine / warmup_constant / warmup_linear
  * modules/optimization_adamw.py:111-174 AdaptAdamW.step  (bias-corrected Adam, decoupled decay applied
                                           BEFORE the update with the scheduled lr, per-parameter step)
  * torch.nn.utils.clip_grad_norm_ as called at main_task_align.py:326 (L2, all grads, coef clamped to 1)
  * main_task_align.py:175-241            prep_optimizer's 8 parameter groups (name-substring routing)
  * main_task_align.py:388-441            requires_grad freezing by name
  * main_task_align.py:292-359            train_epoch's per-iteration order incl. the NaN skip and the
                                           logit_scale clamp

Pinned against the real reference by tests/golden/train_*.npz (tests/golden/make_golden_train.py runs the
reference's own prep_optimizer / train_epoch / AdaptAdamW on CPU): tests/test_train_oracle_golden.py.
"""
import math

import numpy as np

F = np.float32
LN100 = float(np.log(100))


# ---------------------------------------------------------------------------------------------------
# schedules (optimization_adamw.py:26-45)
# ---------------------------------------------------------------------------------------------------
def schedule(kind, x, warmup, lr_start=0.0, lr_end=0.0):
    if kind == "warmup_cosine":
        if x < warmup:
            return x * (1.0 - lr_start) / warmup + lr_start
        y = (x - warmup) / (1.0 - warmup)
        return lr_end + 0.5 * (1.0 - lr_end) * (1.0 + math.cos(math.pi * y))
    if kind == "warmup_constant":
        return x / warmup if x < warmup else 1.0
    if kind == "warmup_linear":
        return x / warmup if x < warmup else max((x - 1.0) / (warmup - 1.0), 0.0)
    raise ValueError(kind)


def scheduled_lr(group, step):
    if group["t_total"] == -1:
        return group["lr"]
    return group["lr"] * schedule(group["schedule"], step / group["t_total"], group["warmup"],
                                  group.get("lr_start", 0.0), group.get("lr_end", 0.0))


# ---------------------------------------------------------------------------------------------------
# AdaptAdamW.step for one tensor (optimization_adamw.py:128-172); all arrays float32, updated in place
# ---------------------------------------------------------------------------------------------------
def adamw_update(p, g, m, v, step, group):
    b1, b2 = group["b1"], group["b2"]
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    m *= F(b1)
    m += F(1.0 - b1) * g
    v *= F(b2)
    v += F(1.0 - b2) * g * g
    denom = np.sqrt(v) / F(math.sqrt(bc2)) + F(group["e"])
    lr_t = scheduled_lr(group, step)
    p *= F(1.0 - lr_t * group["weight_decay"])
    p += F(-(lr_t / bc1)) * (m / denom)
    return lr_t


def clip_grad_norm(grads, max_norm):
    """grads: list of float32 arrays, scaled in place. Returns the total norm (before clipping)."""
    norms = np.array([np.sqrt(np.sum(g.astype(F) ** 2, dtype=F)) for g in grads], dtype=F)
    total = F(np.sqrt(np.sum(norms * norms, dtype=F)))
    coef = F(max_norm) / (total + F(1e-6))
    coef = F(min(float(coef), 1.0))
    for g in grads:
        g *= coef
    return float(total)


class AdamWState:
    """Optimizer over a dict name -> float32 array with reference-style param groups."""

    def __init__(self, groups):
        self.groups = groups  # list of dicts with 'names' + hyper-parameters (defaults filled in)
        self.state = {}

    def step(self, params, grads):
        lrs = []
        for g in self.groups:
            for n in g["names"]:
                if n not in grads or grads[n] is None:
                    continue
                st = self.state.setdefault(n, dict(step=0, m=np.zeros_like(params[n]), v=np.zeros_like(params[n])))
                st["step"] += 1
                lrs.append(adamw_update(params[n], grads[n], st["m"], st["v"], st["step"], g))
        return lrs


# ---------------------------------------------------------------------------------------------------
# name routing (main_task_align.py:175-241 and :388-441)
# ---------------------------------------------------------------------------------------------------
NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")
_SLOW_VIS = ("clip.visual.class_embedding", "clip.visual.positional_embedding", "clip.visual.conv1.",
             "clip.visual.ln_pre.", "clip.logit_scale", "clip.ln_final.", "clip.text_projection")
_SLOW_TXT = ("clip.positional_embedding", "clip.token_embedding.")
_SLOW_LAYERS = ("clip.visual.transformer.layers0.", "clip.transformer.resblocks.")


def group_index(name):
    """0/1 CLIP-initialised (lower_lr) decay/no-decay, 2/3 text embeddings (lower_text_lr), 4/5 new tensors
    inside clip.* (lr), 6/7 everything outside clip.* (default lr).  `prefix_ in n` is a substring test."""
    nd = int(any(s in name for s in NO_DECAY))
    if "clip." not in name:
        return 6 + nd
    if name.startswith(_SLOW_VIS):
        return 0 + nd
    if name.startswith(_SLOW_TXT):
        return 2 + nd
    if name.startswith(_SLOW_LAYERS):
        return 0 + nd
    return 4 + nd


def build_groups(names, args, t_total):
    """names: ordered named_parameters() names (frozen ones included, as the reference does)."""
    lower = args["lower_lr"] if args.get("lower_lr", 0.0) != 0.0 else args["lr"] * args.get("coef_lr", 1.0)
    lower_text = args["lower_text_lr"] if args.get("lower_text_lr", 0.0) != 0.0 else lower
    lrs = [lower, lower, lower_text, lower_text, args["lr"], args["lr"], args["lr"], args["lr"]]
    groups = []
    for gi in range(8):
        groups.append(dict(names=[n for n in names if group_index(n) == gi], lr=lrs[gi],
                           weight_decay=args["weight_decay"] if gi % 2 == 0 else 0.0,
                           schedule="warmup_cosine", warmup=args["warmup_proportion"], t_total=t_total,
                           b1=args["opt_b1"], b2=args["opt_b2"], e=args["eps"],
                           lr_start=args.get("lr_start", 0.0), lr_end=args.get("lr_end", 0.0)))
    return groups


def frozen_names(names, args):
    """Names (relative to the full model, 'clip.' prefixed) whose requires_grad main() switches off."""
    out = []
    fl, ftl, first = args.get("freeze_layer_num", 0), args.get("freeze_text_layer_num", 0), args.get("first_stage_layer", 10)

    def layer(n, key):
        return int(n.split(key)[1].split(".")[0])

    for full in names:
        if not full.startswith("clip."):
            continue
        n = full[len("clip."):]
        frozen = False
        if fl > -1:
            keep = (n.startswith(("ln_final.", "text_projection", "logit_scale", "visual.ln_post.", "visual.proj"))
                    or (n.startswith("visual.transformer.layers0.") and layer(n, ".layers0.") >= fl)
                    or (n.startswith("visual.transformer.layers2.") and layer(n, ".layers2.") >= fl - first)
                    or (n.startswith("transformer.resblocks.") and layer(n, ".resblocks.") >= fl)
                    or n.startswith(("visual.transformer.semantic_layer1", "visual.transformer.semantic_layer2",
                                     "visual.transformer.layers_mae", "visual.transformer.reconstruct_layer")))
            # a layers0/layers2/resblocks tensor below the threshold falls through to the freeze
            frozen = not keep
        if ftl > 0:
            if n.startswith("positional_embedding") or n.startswith("token_embedding.weight"):
                frozen = True
            elif n.startswith("transformer.resblocks.") and layer(n, ".resblocks.") < ftl:
                frozen = True
        if args.get("pretrained_clip_name", "ViT-B/16") in ("ViT-B/32", "ViT-B/16", "ViT-L/14"):
            if n.startswith("visual.positional_embedding") or n.startswith("visual.conv1.weight"):
                frozen = True
        if frozen:
            out.append(full)
    return out


# ---------------------------------------------------------------------------------------------------
# train_epoch restated on the oracle forward (main_task_align.py:292-359)
# ---------------------------------------------------------------------------------------------------
def train_trajectory(spec, flags, P, batches, noises, args, t_total, cross_mode="t18", param_names=None):
    """P: dict name -> torch fp32 leaf (requires_grad False for frozen / buffers).  Runs len(batches) iterations
    in place on P; returns dict(losses, lrs, grad_norms)."""
    import torch
    from oracle import segclip_oracle as so

    names = [n for n in P if P[n].is_floating_point() and (param_names is None or n in param_names)]
    opt = AdamWState(build_groups(names, args, t_total))
    losses, lrs, gnorms = [], [], []
    for batch, noise in zip(batches, noises):
        for t in P.values():
            t.grad = None
        loss, _ = so.segclip_forward(batch, P, spec, noise, flags, cross_mode=cross_mode)
        loss.backward()
        grads = {n: P[n].grad.numpy().copy() for n in names if P[n].grad is not None}
        gnorms.append(clip_grad_norm(list(grads.values()), args["clip_grad"]))
        lv = float(loss.detach())
        losses.append(lv)
        if not math.isnan(lv):
            arrs = {n: P[n].detach().numpy() for n in grads}  # views: updated in place
            lrs.append(sorted(set(opt.step(arrs, grads))))
        with torch.no_grad():
            P["clip.logit_scale"].clamp_(max=LN100)
    return dict(losses=losses, lrs=lrs, grad_norms=gnorms, opt=opt)
