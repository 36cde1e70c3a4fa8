"""CPU: CLIP weight import.  modules/module_clip_util.py:174-197 loads the OpenAI archive with
torch.jit.load(...).state_dict() and falls back to torch.load for a plain state-dict file; modules/modeling.py:46-68
renames `visual.transformer.resblocks.i` to layers0.i / layers2.(i-first_stage).  There is no network for ViT-B-16.pt,
so a synthetic archive with the real key layout is written, scripted and saved with torch.jit, and round-tripped."""
import argparse
import logging
import os

import pytest
import torch
from torch import nn

from segclip_amd import synth


class _Box(nn.Module):
    pass


def _module_tree(sd):
    root = _Box()
    for key, val in sd.items():
        node = root
        parts = key.split(".")
        for p in parts[:-1]:
            if not hasattr(node, p):
                node.add_module(p, _Box())
            node = getattr(node, p)
        if val.is_floating_point() and val.dim() > 0 or key == "logit_scale":
            node.register_parameter(parts[-1], nn.Parameter(val.clone(), requires_grad=False))
        else:
            node.register_buffer(parts[-1], val.clone())
    return root


def _archive_state(spec):
    sd = synth.synthetic_clip_state_dict(spec)
    return {k: (synth.closed_form_tensor("clip." + k, v.shape) if v.is_floating_point() and v.dim() > 0 else v)
            for k, v in sd.items()}


@pytest.mark.parametrize("kind", ["jit", "plain"])
def test_clip_archive_round_trip_and_key_renaming(kind, tmp_path):
    from segclip_amd.modules.module_clip import CLIP
    from segclip_amd.modules.modeling import SegCLIP
    spec = synth.SPECS["tiny"]
    sd = _archive_state(spec)
    path = str(tmp_path / ("ViT-tiny.pt"))
    if kind == "jit":
        torch.jit.save(torch.jit.script(_module_tree(sd)), path)
    else:
        torch.save(sd, path)
    got = CLIP.get_config(pretrained_clip_name=path)          # explicit file path (no downloader in this build)
    assert set(got.keys()) == set(sd.keys())
    for k in sd:
        assert torch.equal(got[k].float(), sd[k].float()), k
    # a name that is neither a known model file nor a path fails loudly
    with pytest.raises(RuntimeError):
        CLIP.get_config(pretrained_clip_name="ViT-B/16")
    # the archive drives from_pretrained: dimensions derived from shapes, resblocks renamed, values carried over
    logging.getLogger("seg").setLevel(logging.ERROR)
    orig = CLIP.get_config
    CLIP.get_config = staticmethod(lambda pretrained_clip_name="ViT-B/16": orig(pretrained_clip_name=path))
    try:
        args = argparse.Namespace(local_rank=0, rank=0, world_size=1, pretrained_clip_name="ViT-B/16",
                                  first_stage_layer=10, max_words=spec["context_length"])
        model = SegCLIP.from_pretrained(cache_dir=None, state_dict=None, task_config=args)
    finally:
        CLIP.get_config = orig
    msd = model.state_dict()
    assert model.clip.visual.conv1.weight.shape == sd["visual.conv1.weight"].shape
    for k, v in sd.items():
        if k in ("input_resolution", "context_length", "vocab_size"):
            continue
        nk = "clip." + k
        if k.startswith("visual.transformer.resblocks."):
            i = int(k.split(".")[3])
            rest = ".".join(k.split(".")[4:])
            nk = f"clip.visual.transformer.layers0.{i}.{rest}" if i < 10 else f"clip.visual.transformer.layers2.{i - 10}.{rest}"
        assert nk in msd, nk
        assert torch.equal(msd[nk].float(), v.float()), nk
    # parts that have no CLIP counterpart keep their own initialisation (present, finite)
    assert any(k.startswith("clip.visual.transformer.semantic_layer2.") for k in msd)
    assert all(torch.isfinite(v).all() for v in msd.values() if v.is_floating_point())


def test_init_preweight_tolerates_shape_mismatches_like_the_reference(caplog):
    """modules/util_module.py:118-145: the reference's name-based load collects size mismatches in `error_msgs`, logs
    "Weights from pretrained model cause errors ..." and CONTINUES (its "reset ViT but keep Text Encoder" branch,
    modules/modeling.py:41-43, relies on it: a ViT-B/32-shaped archive goes into a ViT-B/16 model).  Same here: a state dict
    whose vision tower has another patch size (conv1 kernel 32 instead of 16 -> another positional table) loads without raising,
    the mismatching tensors keep their initial values, every matching tensor is taken over, and the three log lines appear."""
    from segclip_amd.modules.util_module import PreTrainedModel
    spec = synth.SPECS["tiny"]
    model, _ = synth.build_model(spec, {}, device="cpu", closed_form=False)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    other = {k: torch.full_like(v, 0.125) if v.is_floating_point() else v.clone() for k, v in before.items()}
    # the "ViT-B/32 into ViT-B/16" situation: twice the patch size -> conv1 kernel and positional table of other shapes
    w = before["clip.visual.conv1.weight"]
    other["clip.visual.conv1.weight"] = torch.zeros(w.shape[0], w.shape[1], 2 * w.shape[2], 2 * w.shape[3])
    pos = before["clip.visual.positional_embedding"]
    other["clip.visual.positional_embedding"] = torch.zeros((pos.shape[0] - 1) // 4 + 1, pos.shape[1])
    other["clip.some_tensor_the_model_does_not_have"] = torch.zeros(3)
    dropped = "clip.text_projection"
    del other[dropped]
    with caplog.at_level(logging.WARNING, logger="seg"):
        logging.getLogger("seg").setLevel(logging.WARNING)
        out = PreTrainedModel.init_preweight(model, other)          # must not raise
    assert out is model
    rep = model.last_load_report
    assert rep["missing_keys"] == [dropped]
    assert rep["unexpected_keys"] == ["clip.some_tensor_the_model_does_not_have"]
    assert len(rep["error_msgs"]) == 2 and all("size mismatch for clip.visual." in m for m in rep["error_msgs"])
    text = caplog.text
    assert "not initialized from pretrained model" in text and "not used in" in text and "cause errors in" in text
    after = model.state_dict()
    for k in ("clip.visual.conv1.weight", "clip.visual.positional_embedding", dropped):
        assert torch.equal(after[k], before[k]), k                    # untouched
    for k, v in after.items():
        if k in ("clip.visual.conv1.weight", "clip.visual.positional_embedding", dropped) or not v.is_floating_point():
            continue
        assert torch.equal(v, other[k]), k                            # taken over
    # and the prefix form (used for partial loads) prepends the prefix and stays silent
    sub = {k[len("clip."):]: torch.full_like(v, 0.25) for k, v in before.items() if k.startswith("clip.ln_final.")}
    PreTrainedModel.init_preweight(model, sub, prefix="clip.")
    assert float(model.clip.ln_final.weight.detach().mean()) == 0.25


def test_mean_pooling_helpers_match_their_definition():
    """modules/modeling.py:322-336 (dead API of the reference's training path, carried for importers)."""
    spec = synth.SPECS["tiny"]
    model, _ = synth.build_model(spec, {}, device="cpu", closed_form=False)
    g = torch.Generator().manual_seed(0)
    seq = torch.randn(3, 7, 5, generator=g)
    vis = torch.randn(3, 4, 5, generator=g)
    mask = torch.tensor([[1, 1, 1, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1], [1, 1, 0, 0, 0, 0, 0]])
    t, v = model._mean_pooling_for_similarity(seq, vis, mask)
    for b in range(3):
        n = int(mask[b].sum())
        assert torch.allclose(t[b], seq[b, 1:n].mean(0), atol=1e-6)    # token 0 (start of text) excluded
    assert torch.allclose(v, vis.mean(1))
    assert torch.equal(mask, torch.tensor([[1, 1, 1, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1], [1, 1, 0, 0, 0, 0, 0]]))   # input untouched
