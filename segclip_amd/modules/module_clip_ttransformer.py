"""Mirror of modules/module_clip_ttransformer.py: 12 pre-LN residual blocks with a causal mask."""
from collections import OrderedDict

import torch
from torch import nn

from .. import config, ops
from .module_clip_util import LayerNorm, QuickGELU


class ResidualAttentionBlock(nn.Module):
    """modules/module_clip_ttransformer.py:20-52.  The reference block consumes an (x, attn_mask,
    video_frame) tuple with x in LND; this mirror keeps that calling convention."""

    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        self.n_head = n_head
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)

    def block_params(self):
        """The 12 parameters in ops.ResBlockFn / ops.ResStackFn order."""
        return (self.ln_1.weight, self.ln_1.bias, self.attn.in_proj_weight, self.attn.in_proj_bias,
                self.attn.out_proj.weight, self.attn.out_proj.bias, self.ln_2.weight, self.ln_2.bias,
                self.mlp.c_fc.weight, self.mlp.c_fc.bias, self.mlp.c_proj.weight, self.mlp.c_proj.bias)

    def forward_nld(self, x, causal):
        return ops.ResBlockFn.apply(x, self.ln_1.weight, self.ln_1.bias, self.attn.in_proj_weight,
                                    self.attn.in_proj_bias, self.attn.out_proj.weight, self.attn.out_proj.bias,
                                    self.ln_2.weight, self.ln_2.bias, self.mlp.c_fc.weight, self.mlp.c_fc.bias,
                                    self.mlp.c_proj.weight, self.mlp.c_proj.bias, self.n_head, causal,
                                    ops.ACT_QUICK_GELU, self.ln_1.eps, config.compute_dtype)

    def forward(self, x_tuple: tuple):
        x, attn_mask, video_frame = x_tuple
        if attn_mask is not None and not callable(attn_mask):
            raise NotImplementedError("padding masks are only reachable from the text-MAE branch (out of scope)")
        y = self.forward_nld(x.permute(1, 0, 2).contiguous().float(), attn_mask is not None)
        return (y.permute(1, 0, 2), attn_mask, video_frame)


class TextTransformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int):
        super().__init__()
        self.width = width
        self.layers = layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])

    def forward_nld(self, x, causal=True):
        blocks = list(self.resblocks)
        if config.fuse_res_stack and len(blocks) > 1:   # the whole tower as one autograd node (ops.ResStackFn)
            b0 = blocks[0]
            return ops.res_stack(x, [b.block_params() for b in blocks], b0.n_head, causal, ops.ACT_QUICK_GELU,
                                 b0.ln_1.eps, config.compute_dtype)
        for blk in blocks:
            x = blk.forward_nld(x, causal)
        return x

    def forward(self, x: torch.Tensor, attn_mask=None, video_frame=-1):
        """x LND like the reference; a callable attn_mask (CLIP.build_attention_mask) selects the causal mask."""
        if attn_mask is not None and not callable(attn_mask):
            raise NotImplementedError("padding masks are only reachable from the text-MAE branch (out of scope)")
        y = self.forward_nld(x.permute(1, 0, 2).contiguous().float(), attn_mask is not None)
        return y.permute(1, 0, 2)
