"""Mirror of modules/module_clip_util.py (hot-path parts only): LayerNorm, QuickGELU,
random_masking, build_attention_mask, CLIP weight locator."""
import os

import torch
from torch import nn

from .. import config, ops

_PT_NAME = {"ViT-B/32": "ViT-B-32.pt", "ViT-B/16": "ViT-B-16.pt", "ViT-L/14": "ViT-L-14.pt"}
_MODELS = dict.fromkeys(["RN50", "RN101", "RN50x4", "RN50x16", "RN50x64", "ViT-B/32", "ViT-B/16", "ViT-L/14"])


class LayerNorm(nn.LayerNorm):
    """modules/module_clip_util.py:126-132: fp32 LayerNorm; here a parameter container whose forward
    launches the HIP kernel (output in the current compute dtype unless out_dtype is given)."""

    def forward(self, x, out_dtype=None):
        return ops.layer_norm(x, self.weight, self.bias, self.eps, out_dtype or config.compute_dtype)


class QuickGELU(nn.Module):
    """modules/module_clip_util.py:134-136.  Normally fused into the producing GEMM's epilogue."""

    def forward(self, x):
        return ops.ActFn.apply(x, ops.ACT_QUICK_GELU)


def random_masking(x, mask_ratio, keep_cls=False, keep_sep=False, cls_pos=None, sep_pos=None):
    """modules/module_clip_util.py:91-124: x (N, L, D) -> x_masked, mask, ids_restore, ids_keep.
    keep_cls pins position 0; keep_sep (text-MAE) is reproduced AS WRITTEN in the reference:
    `noise.scatter_(dim=1, index=sep_pos.unsqueeze(0), value=-1)` with a (1, N) index pins the separator positions
    of every sample in ROW 0 only.  The sort is segclip_mask_sort (stable rank sort = torch's CPU argsort, ties keep
    index order)."""
    if cls_pos is not None:
        raise NotImplementedError("random_masking: cls_pos is never used by the model")
    N, Lq, D = x.shape
    len_keep = int(Lq * (1 - mask_ratio))
    noise = config.rand((N, Lq), x.device)
    if keep_cls or keep_sep:
        noise = noise.clone()
    if keep_cls:
        noise[:, 0] = -1.
    if keep_sep:
        assert sep_pos is not None
        noise[0].index_fill_(0, sep_pos.reshape(-1), -1.)
    if not keep_cls:
        raise NotImplementedError("random_masking: the model only calls it with keep_cls=True")
    ids_shuffle, ids_restore, mask = ops.mask_sort(noise, len_keep)   # the kernel pins position 0 itself as well
    ids_keep = ids_shuffle[:, :len_keep].contiguous()
    x_masked = ops.GatherRowsFn.apply(x, ids_keep)
    return x_masked, mask, ids_restore, ids_keep


class CLIP_Module(nn.Module):
    @staticmethod
    def get_config(pretrained_clip_name="ViT-B/32"):
        """modules/module_clip_util.py:174-197 without the downloader (no network): the OpenAI CLIP
        archive must sit next to this file or be given as a path."""
        here = os.path.dirname(os.path.abspath(__file__))
        model_path = os.path.join(here, _PT_NAME.get(pretrained_clip_name, "ViT-B-32.pt"))
        if not os.path.exists(model_path):
            if os.path.isfile(pretrained_clip_name):
                model_path = pretrained_clip_name
            else:
                raise RuntimeError(f"Model {pretrained_clip_name} not found at {model_path} (downloads are "
                                   f"disabled); available models = {list(_MODELS.keys())}")
        try:
            model = torch.jit.load(model_path, map_location="cpu").eval()
            return model.state_dict()
        except RuntimeError:
            return torch.load(model_path, map_location="cpu")

    def build_attention_mask(self, context_length):
        """modules/module_clip_util.py:199-205.  Kept for API parity; the text blocks generate the causal
        mask inside the attention kernel and never read this tensor."""
        mask = torch.zeros(context_length, context_length)
        mask.fill_(float("-inf"))
        mask.triu_(1)
        return mask
