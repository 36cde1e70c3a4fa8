"""Mirror of modules/module_clip.py: the CLIP container (vision tower, text tower, embeddings,
ln_final, text_projection, logit_scale)."""
from typing import Tuple, Union

import numpy as np
import weakref

import torch
from torch import nn

from .. import config, ops
from .module_clip_util import CLIP_Module, LayerNorm, _MODELS, random_masking
from .module_clip_ttransformer import TextTransformer
from .module_clip_vtransformer import VisualTransformer


class CLIP(CLIP_Module):
    def __init__(self, embed_dim: int, image_resolution: int, vision_layers: Union[Tuple[int, int, int, int], int],
                 vision_width: int, vision_patch_size: int, context_length: int, vocab_size: int,
                 transformer_width: int, transformer_heads: int, transformer_layers: int,
                 first_stage_layer: int = 10):
        super().__init__()
        self.context_length = context_length
        vision_heads = vision_width // 64
        self.visual = VisualTransformer(input_resolution=image_resolution, patch_size=vision_patch_size,
                                        width=vision_width, layers=vision_layers, heads=vision_heads,
                                        output_dim=embed_dim, first_stage_layer=first_stage_layer)
        self.transformer = TextTransformer(width=transformer_width, layers=transformer_layers, heads=transformer_heads)
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(self.context_length, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.initialize_parameters()

    @property
    def dtype(self):
        return self.visual.proj.dtype

    def encode_image_hidden_ln(self, image, video_frame=-1, mask_ratio=0.):
        hidden, mae_mask, mae_ids_restore, mid_states = self.visual(image.type(self.dtype), video_frame=video_frame,
                                                                    mask_ratio=mask_ratio)
        hidden_ln = self.visual.ln_post(hidden)
        return hidden_ln, mae_mask, mae_ids_restore, mid_states

    def encode_image(self, image, return_hidden=False, video_frame=-1, mask_ratio=0.):
        """modules/module_clip.py:89-103."""
        hidden_ln, mae_mask, mae_ids_restore, mid_states = self.encode_image_hidden_ln(image, video_frame=video_frame,
                                                                                       mask_ratio=mask_ratio)
        hidden = ops.linear(hidden_ln, self.visual.proj, None, out_dtype=torch.float32,
                            act_dtype=config.compute_dtype, w_kn=True)
        x = hidden[:, 0, :]
        if mask_ratio > 0.:
            assert return_hidden is True
        if return_hidden:
            if mask_ratio > 0.:
                return x, hidden, mae_mask, mae_ids_restore, mid_states
            return x, hidden, mid_states
        return x

    def encode_image_pooled(self, image):
        """Training fast path of encode_image (modules/module_clip.py:89-103): only the pooled feature the contrastive loss
        uses.  ln_post and the projection are per-row operations, so applying them to the max-pooled row alone gives the
        same values as hidden[:, 0, :] of the full path, without normalising / projecting (and back-propagating zeros
        through) the other 196 rows.  -> (feature (B, embed_dim) fp32, mid_states)"""
        cls, _, _, mid_states = self.visual(image.type(self.dtype), pooled_only=True)
        h = self.visual.ln_post(cls)
        x = ops.linear(h, self.visual.proj, None, out_dtype=torch.float32, act_dtype=config.compute_dtype, w_kn=True)
        return x, mid_states

    def _text_trunk(self, text):
        x = ops.EmbedFn.apply(text, self.token_embedding.weight, self.positional_embedding)
        return self.transformer.forward_nld(x, causal=True)

    def encode_text(self, text, attn_mask=None, return_hidden=False, mask_ratio=0.):
        """modules/module_clip.py:105-143.  mask_ratio > 0 (text-MAE): the kept tokens run through the causal tower in
        the SHUFFLED order random_masking leaves them in, exactly as the reference does."""
        if attn_mask is not None and not callable(attn_mask):
            raise NotImplementedError("encode_text: only the causal (callable) attention mask is used by the model")
        if mask_ratio > 0.:
            assert return_hidden is True
            x = ops.EmbedFn.apply(text, self.token_embedding.weight, self.positional_embedding)
            x, mae_mask, mae_ids_restore, ids_keep = random_masking(x, mask_ratio, keep_cls=True, keep_sep=True,
                                                                    sep_pos=text.argmax(dim=-1))
            text = torch.gather(text, dim=1, index=ids_keep)
            x = self.transformer.forward_nld(x, causal=True)
            hidden_ln = self.ln_final(x)
            hidden = ops.linear(hidden_ln, self.text_projection, None, out_dtype=torch.float32,
                                act_dtype=config.compute_dtype, w_kn=True)
            pooled = ops.GatherRowsFn.apply(hidden, text.argmax(dim=-1).view(-1, 1)).squeeze(1)
            return pooled, hidden, mae_mask, mae_ids_restore
        x = self._text_trunk(text)
        eot = text.argmax(dim=-1)
        if not return_hidden:
            return self._project_eot(x, eot)
        hidden_ln = self.ln_final(x)
        hidden = ops.linear(hidden_ln, self.text_projection, None, out_dtype=torch.float32,
                            act_dtype=config.compute_dtype, w_kn=True)
        x = ops.GatherRowsFn.apply(hidden, eot.view(-1, 1)).squeeze(1)
        return x, hidden

    def _project_eot(self, x, eot):
        """EOT row first, then ln_final + projection on B rows instead of B*L (same values for the row the
        loss uses: LayerNorm and the projection are per-row; modules/module_clip.py:129-136)."""
        xe = ops.GatherRowsFn.apply(x, eot.view(-1, 1))
        h = self.ln_final(xe)
        return ops.linear(h, self.text_projection, None, out_dtype=torch.float32, act_dtype=config.compute_dtype,
                          w_kn=True).squeeze(1)

    def _trim_len(self, text):
        """config.text_trim: number of leading token positions that can reach the loss = largest EOT position + 1."""
        hint = config.text_trim_hint
        if hint is not None:
            return max(1, min(int(hint), text.shape[1]))
        # cached per id TENSOR OBJECT (weak reference + version counter), never per address: a data loader hands over a fresh
        # tensor every step, and the caching allocator very often gives it the address the previous batch just freed - an
        # (address, version, shape) key then returned the OLD batch's length, cut the ids in front of a later EOT and argmax
        # picked a wrong position, silently.  One host synchronisation per new id tensor.
        c = getattr(self, "_trim_cache", None)
        if c is None or c[0]() is not text or c[1] != text._version:
            c = (weakref.ref(text), text._version, int(text.argmax(dim=-1).max()) + 1)
            self._trim_cache = c
        return max(1, min(c[2], text.shape[1]))

    def encode_text_eot(self, text):
        """Training fast path: only the pooled (EOT) feature.  config.text_trim: the causal tower runs on the positions up to
        the batch's last EOT only (the rest cannot reach the EOT rows; see config.py)."""
        if config.text_trim:
            keep = self._trim_len(text)
            if keep < text.shape[1]:
                text = text[:, :keep].contiguous()
        return self._project_eot(self._text_trunk(text), text.argmax(dim=-1))

    def forward(self, image, text):
        """modules/module_clip.py:145-159 (plain CLIP logits; API parity, unused by training)."""
        image_features = ops.L2NormFn.apply(self.encode_image(image).float())
        text_features = ops.L2NormFn.apply(self.encode_text(text, attn_mask=self.build_attention_mask).float())
        logit_scale = self.logit_scale.exp()
        raw = ops.bmm(image_features.unsqueeze(0), text_features.unsqueeze(0), transB=True)[0]
        logits_per_image = logit_scale * raw
        return logits_per_image, logits_per_image.t()

    def initialize_parameters(self):
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        proj_std = (self.transformer.width ** -0.5) * ((2 * self.transformer.layers) ** -0.5)
        attn_std = self.transformer.width ** -0.5
        fc_std = (2 * self.transformer.width) ** -0.5
        for block in self.transformer.resblocks:
            nn.init.normal_(block.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(block.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(block.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(block.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=self.transformer.width ** -0.5)


def available_models():
    return list(_MODELS.keys())
