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
