"""Synthetic model specs, closed-form weights and synthetic batches.

Shared by bench.py, the tests and tests/golden/make_golden.py so that no weight file has to be
committed: every parameter is a pure function of (its reference name, its shape).
Batch contract follows the reference dataloader (dataloaders/dataloader_cc_retrieval.py:169-174,
SURVEY.md section 8b/8d).
"""
import math
import zlib

import torch

# Model specs.  Keys mirror what modules/modeling.py:86-109 derives from the CLIP state-dict.
SPECS = {
    # reduced-width model used for full-tensor golden vectors (SegViT depth is hard-wired to 10+2,
    # modules/module_seg_vit.py:356, so only widths / resolution / text depth shrink)
    "tiny": dict(embed_dim=64, image_res=64, patch=16, vision_width=128, context_length=16,
                 vocab_size=512, text_width=64, text_layers=2),
    # BASELINE.json configs 1-4
    "vitb16": dict(embed_dim=512, image_res=224, patch=16, vision_width=768, context_length=77,
                   vocab_size=49408, text_width=512, text_layers=12),
    # BASELINE.json config 5 ("ViT-L/14"-width at 336^2; still 10+2 blocks)
    "vitl14_336": dict(embed_dim=768, image_res=336, patch=14, vision_width=1024, context_length=77,
                       vocab_size=49408, text_width=768, text_layers=12),
}


def _seed(name):
    return zlib.crc32(name.encode()) & 0x7FFFFFFF


def closed_form_tensor(name, shape, kind=None):
    """Deterministic value for the parameter called `name` (reference state-dict key)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(_seed(name))
    shape = tuple(shape)
    leaf = name.split(".")[-1]
    if name.endswith("logit_scale"):
        return torch.tensor(math.log(1 / 0.07), dtype=torch.float32)
    r = torch.randn(shape, generator=g, dtype=torch.float32) if len(shape) else torch.zeros(())
    is_norm = any(t in name for t in (".ln_", "ln_pre", "ln_post", "ln_final", ".norm", "cross_ln", "k_ln",
                                      ".ln.", "decoder_norm"))
    if is_norm and leaf == "weight":
        return 1.0 + 0.1 * r
    if leaf in ("bias", "in_proj_bias"):
        return 0.02 * r
    if "semantic_center" in name:
        return 0.5 * r
    if "mask_token" in name:
        return 0.02 * r
    if "class_embedding" in name or "positional_embedding" in name:
        return 0.05 * r
    if "token_embedding" in name:
        return 0.05 * r
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        if leaf in ("proj", "text_projection"):  # stored (in, out)
            fan_in = shape[0]
        return r * (fan_in ** -0.5)
    return 0.02 * r


def apply_closed_form_weights(model):
    """Overwrite every parameter of a (reference-shaped) module tree in place.
    decoder_pos_embed (frozen sin-cos table, modules/module_mae.py:258-259) is left as built."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("decoder_pos_embed"):
                continue
            p.copy_(closed_form_tensor(name, p.shape).to(p.dtype))
    return model


def synthetic_batch(spec, batch, seed=0, device="cpu", max_body=None, with_seg=True, n_seg=6):
    """Image-text batch with the reference dataloader's tensor contract.
    image is fp32 here (the reference hands over fp64 and calls .float(), modeling.py:182)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(1000003 * seed + 17)
    L = spec["context_length"]
    V = spec["vocab_size"]
    res = spec["image_res"]
    grid = res // spec["patch"]
    image = torch.randn(batch, 1, 3, res, res, generator=g, dtype=torch.float32)
    ids = torch.zeros(batch, 1, L, dtype=torch.int64)
    mask = torch.zeros(batch, 1, L, dtype=torch.int64)
    hi = min(29, L - 3) if max_body is None else min(max_body, L - 2)
    lo = min(5, hi)
    sot, eot = V - 2, V - 1
    for b in range(batch):
        n = int(torch.randint(lo, hi + 1, (1,), generator=g))
        ids[b, 0, 0] = sot
        ids[b, 0, 1:n + 1] = torch.randint(1, V - 2, (n,), generator=g)
        ids[b, 0, n + 1] = eot
        mask[b, 0, :n + 2] = 1
    seg_ids = torch.zeros(batch, 1, L, dtype=torch.int64)
    out = dict(input_ids=ids, segment_ids=seg_ids, input_mask=mask, image=image)
    if with_seg:
        out["image_seg"] = torch.randint(0, n_seg, (batch, 1, grid, grid), generator=g, dtype=torch.int64)
    return {k: v.to(device) for k, v in out.items()}


def synthetic_noise(spec, batch, seed=0, device="cpu", groups=8):
    """The three RNG draws of one full-loss training forward, in reference order:
    Gumbel (B,G,N) main pass; rand (B,N+1) MAE masking; Gumbel (B,G,keep-1) MAE pass."""
    g = torch.Generator(device="cpu")
    g.manual_seed(7919 * seed + 3)
    n = (spec["image_res"] // spec["patch"]) ** 2

    def gumbel(shape):
        u = torch.rand(shape, generator=g, dtype=torch.float32).clamp_(1e-10, 1 - 1e-7)
        return -torch.log(-torch.log(u))

    keep = int((n + 1) * (1 - 0.75))
    out = dict(gumbel_main=gumbel((batch, groups, n)),
               mask_noise=torch.rand(batch, n + 1, generator=g, dtype=torch.float32),
               gumbel_mae=gumbel((batch, groups, keep - 1)))
    # text-MAE masking noise (drawn last so that the three tensors above keep their round-1 values)
    out["text_mask_noise"] = torch.rand(batch, spec["context_length"], generator=g, dtype=torch.float32)
    return {k: v.to(device) for k, v in out.items()}


def synthetic_clip_state_dict(spec):
    """Key/shape skeleton of an OpenAI-CLIP ViT state-dict - what CLIP.get_config() returns for a real
    ViT-B-16.pt (modules/module_clip_util.py:174-197); all model dimensions are derived from these shapes
    (modules/modeling.py:86-109).  Values are placeholders; callers overwrite every parameter."""
    W, Wt, E = spec["vision_width"], spec["text_width"], spec["embed_dim"]
    p, res = spec["patch"], spec["image_res"]
    n = (res // p) ** 2 + 1
    sd = {
        "visual.conv1.weight": torch.zeros(W, 3, p, p),
        "visual.class_embedding": torch.zeros(W),
        "visual.positional_embedding": torch.zeros(n, W),
        "visual.proj": torch.zeros(W, E),
        "visual.ln_pre.weight": torch.ones(W), "visual.ln_pre.bias": torch.zeros(W),
        "visual.ln_post.weight": torch.ones(W), "visual.ln_post.bias": torch.zeros(W),
        "text_projection": torch.zeros(Wt, E),
        "positional_embedding": torch.zeros(spec["context_length"], Wt),
        "token_embedding.weight": torch.zeros(spec["vocab_size"], Wt),
        "ln_final.weight": torch.ones(Wt), "ln_final.bias": torch.zeros(Wt),
        "logit_scale": torch.tensor(math.log(1 / 0.07)),
        "input_resolution": torch.tensor(res), "context_length": torch.tensor(spec["context_length"]),
        "vocab_size": torch.tensor(spec["vocab_size"]),
    }

    def block(prefix, d):
        sd[prefix + "attn.in_proj_weight"] = torch.zeros(3 * d, d)
        sd[prefix + "attn.in_proj_bias"] = torch.zeros(3 * d)
        sd[prefix + "attn.out_proj.weight"] = torch.zeros(d, d)
        sd[prefix + "attn.out_proj.bias"] = torch.zeros(d)
        for ln in ("ln_1", "ln_2"):
            sd[prefix + ln + ".weight"] = torch.ones(d)
            sd[prefix + ln + ".bias"] = torch.zeros(d)
        sd[prefix + "mlp.c_fc.weight"] = torch.zeros(4 * d, d)
        sd[prefix + "mlp.c_fc.bias"] = torch.zeros(4 * d)
        sd[prefix + "mlp.c_proj.weight"] = torch.zeros(d, 4 * d)
        sd[prefix + "mlp.c_proj.bias"] = torch.zeros(d)

    for i in range(12):
        block(f"visual.transformer.resblocks.{i}.", W)
    for i in range(spec["text_layers"]):
        block(f"transformer.resblocks.{i}.", Wt)
    return sd


def build_model(spec, flags=None, rank=0, world_size=1, device="cuda", closed_form=True):
    """segclip_amd.modules.modeling.SegCLIP built through its own from_pretrained() from a synthetic CLIP
    state-dict (there is no network for ViT-B-16.pt), in train mode, closed-form weights."""
    import argparse

    from .modules.modeling import SegCLIP
    from .modules.module_clip import CLIP
    flags = flags or {}
    args = argparse.Namespace(local_rank=0, rank=rank, world_size=world_size, pretrained_clip_name="ViT-B/16",
                              first_stage_layer=10, use_vision_mae_recon=flags.get("use_vision_mae_recon", False),
                              use_text_mae_recon=flags.get("use_text_mae_recon", False),
                              use_seglabel=flags.get("use_seglabel", False),
                              mae_vis_mask_ratio=0.75, max_words=spec["context_length"])
    import logging
    lg = logging.getLogger("seg")
    lvl = lg.level
    lg.setLevel(logging.ERROR)
    orig = CLIP.get_config
    CLIP.get_config = staticmethod(lambda pretrained_clip_name="ViT-B/16": synthetic_clip_state_dict(spec))
    try:
        model = SegCLIP.from_pretrained(cache_dir=None, state_dict=None, task_config=args)
    finally:
        CLIP.get_config = orig
        lg.setLevel(lvl)
    if closed_form:
        apply_closed_form_weights(model)
    return model.to(device).train(), args
