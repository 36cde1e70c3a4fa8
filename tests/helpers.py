"""Shared test utilities (oracle-side parameter dicts, golden loading)."""
import os

import numpy as np
import torch

from segclip_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FULL_FLAGS = dict(use_seglabel=True, use_vision_mae_recon=True)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def oracle_params(spec, names_shapes, requires_grad=True):
    """Closed-form parameter dict keyed by reference state-dict names (same values build_model loads)."""
    from oracle import segclip_oracle as so
    P = {}
    for name, shape in names_shapes:
        if name.endswith("decoder_pos_embed"):
            if name.startswith("seq_mae_decoder."):
                t = so.position_encoding_init(shape[-2], shape[-1]).reshape(shape)
            else:
                n = (spec["image_res"] // spec["patch"])
                t = so.sincos_pos_embed_2d(shape[-1], n).reshape(shape)
            P[name] = t
            continue
        t = synth.closed_form_tensor(name, shape).float()
        P[name] = t.requires_grad_(requires_grad)
    return P


def model_param_shapes(spec, flags):
    """(name, shape) of every state-dict entry, taken from the module mirror built on CPU."""
    model, _ = synth.build_model(spec, flags, device="cpu", closed_form=False)
    return [(k, tuple(v.shape)) for k, v in model.state_dict().items()]


def noise_items(noise, flags):
    items = [("gumbel", noise["gumbel_main"])]
    if flags.get("use_vision_mae_recon"):
        items += [("rand", noise["mask_noise"]), ("gumbel", noise["gumbel_mae"])]
    return items
