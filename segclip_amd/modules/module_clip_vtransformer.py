"""Mirror of modules/module_clip_vtransformer.py: ViT front end (patch embed, positional table,
ln_pre, optional MAE masking) in front of SegViT."""
import torch
from torch import nn

from .. import config, ops
from .module_clip_util import LayerNorm
from .module_seg_vit import SegViT


class VisualTransformer(nn.Module):
    def __init__(self, input_resolution: int, patch_size: int, width: int, layers: int, heads: int, output_dim: int,
                 first_stage_layer: int = 10):
        super().__init__()
        self.input_resolution = input_resolution
        self.output_dim = output_dim
        self.patch_size = patch_size
        self.conv1 = nn.Conv2d(in_channels=3, out_channels=width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = SegViT(width, patch_size=patch_size, input_resolution=input_resolution,
                                  first_stage_layer=first_stage_layer)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))

    def train(self, mode=True):
        # the fused optimizer rewrites parameters through raw pointers (no autograd version bump), so a resampled
        # table cached before a training phase may be stale: drop it at every train()/eval() switch
        self.__dict__.pop("_pos_cache", None)
        return super().train(mode)

    def get_pos_embed(self, h_, w_):
        """modules/module_clip_vtransformer.py:35-53.  Training: the raw table.  Eval at a different grid: the
        patch rows are resampled bicubically (F.interpolate(..., mode='bicubic', align_corners=False)) by
        segclip_interp_bicubic; the result is cached per (grid, table version)."""
        n = self.positional_embedding.shape[0] - 1
        if self.training or (h_ * w_ == n and h_ == w_):
            return self.positional_embedding
        key = (h_, w_, self.positional_embedding._version, self.positional_embedding.data_ptr())
        cache = self.__dict__.setdefault("_pos_cache", {})
        if key not in cache:
            cache.clear()
            cache[key] = ops.interp_pos_table(self.positional_embedding, h_, w_)
        return cache[key]

    def forward(self, x: torch.Tensor, video_frame=-1, mask_ratio=0., pooled_only=False):
        """pooled_only (mask_ratio == 0 only): returns (cls (B, D), None, None, mid_states) - see SegViT.forward_patches"""
        B, _, H, W = x.shape
        h_, w_ = H // self.patch_size, W // self.patch_size
        pos = self.get_pos_embed(h_, w_)
        # patch embed + positional add fused; the CLS row is never materialised (SegViT discards it)
        xp = ops.PatchEmbedFn.apply(x.float(), self.conv1.weight, self.class_embedding, pos, self.patch_size,
                                    config.compute_dtype)
        xp = self.ln_pre(xp, out_dtype=torch.float32)
        mae_mask, mae_ids_restore = None, None
        if mask_ratio > 0.:
            # random_masking(keep_cls=True) on the (B, 1+T) token axis: CLS is kept first and dropped again
            # by SegViT, so only ids_keep[:, 1:] - 1 index the patch tensor
            Lq = xp.shape[1] + 1
            len_keep = int(Lq * (1 - mask_ratio))
            noise = config.rand((B, Lq), xp.device)
            ids_shuffle, mae_ids_restore, mae_mask = ops.mask_sort(noise, len_keep)
            keep = (ids_shuffle[:, 1:len_keep] - 1).contiguous()
            xp = ops.GatherRowsFn.apply(xp, keep)
        if pooled_only and mask_ratio > 0.:
            raise ValueError("pooled_only is the main (unmasked) branch's fast path")
        x, mid_states = self.transformer.forward_patches(xp, pooled_only=pooled_only)
        if len(mid_states["attns"]) == 0:
            assert mask_ratio > 0., "Must pass the semantic layer~"
        return x, mae_mask, mae_ids_restore, mid_states
