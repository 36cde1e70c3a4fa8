"""Mirror of modules/module_mae.py: MAE decoder (vision: forward_vis; text: forward_seq), patchify, position tables."""
import numpy as np
import torch
from torch import nn

from .. import config, ops


def patchify(imgs, patch_size):
    """modules/module_mae.py:18-29."""
    return ops.patchify_target(imgs.float(), patch_size)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    """modules/module_mae.py:63-108 (float64 numpy, 'w goes first')."""
    grid_h = np.arange(grid_size, dtype=np.float32)
    grid_w = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, grid_size, grid_size])

    def one(d, pos):
        omega = np.arange(d // 2, dtype=np.float64) / (d / 2.)
        omega = 1. / 10000 ** omega
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    emb = np.concatenate([one(embed_dim // 2, grid[0]), one(embed_dim // 2, grid[1])], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


def position_encoding_init(n_position, d_pos_vec):
    """modules/module_mae.py:44-54: sinusoid table, row 0 zero, sin on even / cos on odd columns."""
    position_enc = np.array([[pos / np.power(10000, 2 * i / d_pos_vec) for i in range(d_pos_vec)]
                             if pos != 0 else np.zeros(d_pos_vec) for pos in range(n_position)])
    position_enc[1:, 0::2] = np.sin(position_enc[1:, 0::2])
    position_enc[1:, 1::2] = np.cos(position_enc[1:, 1::2])
    return position_enc


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, in_features)


class Block(nn.Module):
    """timm-style block, modules/module_mae.py:185-201 (qkv bias, erf GELU, LN eps from norm_layer)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        qb = self.attn.qkv.bias
        if qb is None:
            qb = torch.zeros(self.attn.qkv.weight.shape[0], device=x.device, dtype=torch.float32)
        return ops.ResBlockFn.apply(x.float(), self.norm1.weight, self.norm1.bias, self.attn.qkv.weight, qb,
                                    self.attn.proj.weight, self.attn.proj.bias, self.norm2.weight, self.norm2.bias,
                                    self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias,
                                    self.attn.num_heads, False, ops.ACT_GELU_ERF, self.norm1.eps, config.compute_dtype)


class ResidualAttentionBlock(nn.Module):
    """Text-decoder block, modules/module_mae.py:203-232: nn.MultiheadAttention with the additive key-padding mask
    (1 - attention_mask) * -1e6, erf-GELU Mlp, LayerNorm.  forward takes (x (B,L,D), key_len (B,) int32 or None):
    the mask of the caption batches is a PREFIX mask, so it is handed to the attention kernel as a per-sample key count
    (exp(-1e6) is exactly 0 in fp32: excluding the keys is the same arithmetic)."""

    def __init__(self, d_model, n_head, mlp_ratio=4., norm_layer=nn.LayerNorm):
        super().__init__()
        self.n_head = n_head
        self.norm1 = norm_layer(d_model)
        self.attn = nn.MultiheadAttention(d_model, n_head)  # parameter container
        self.norm2 = norm_layer(d_model)
        self.mlp = Mlp(d_model, int(d_model * mlp_ratio))

    def forward(self, x, key_len=None):
        return ops.ResBlockFn.apply(x.float(), self.norm1.weight, self.norm1.bias, self.attn.in_proj_weight,
                                    self.attn.in_proj_bias, self.attn.out_proj.weight, self.attn.out_proj.bias,
                                    self.norm2.weight, self.norm2.bias, self.mlp.fc1.weight, self.mlp.fc1.bias,
                                    self.mlp.fc2.weight, self.mlp.fc2.bias, self.n_head, False, ops.ACT_GELU_ERF,
                                    self.norm1.eps, config.compute_dtype, key_len)


class MAEDecoder(nn.Module):
    """modules/module_mae.py:235-355."""

    def __init__(self, embed_dim, decoder_embed_dim, image_resolution, patch_size, decoder_depth=8,
                 decoder_num_heads=16, mlp_ratio=4., norm_layer=nn.LayerNorm, in_chans=3, choice_seq=False,
                 pred_len=None, seq_len=None):
        super().__init__()
        if choice_seq:
            assert pred_len is not None and seq_len is not None
        else:
            pred_len = patch_size ** 2 * in_chans
        self.choice_seq, self.pred_len, self.seq_len = choice_seq, pred_len, seq_len
        self.patch_size = patch_size
        self.num_patches = (image_resolution // patch_size) ** 2
        self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, self.num_patches + 1 if not choice_seq else seq_len,
                                                          decoder_embed_dim), requires_grad=False)
        if not choice_seq:
            self.decoder_blocks = nn.ModuleList([Block(decoder_embed_dim, decoder_num_heads, mlp_ratio, qkv_bias=True,
                                                       norm_layer=norm_layer) for _ in range(decoder_depth)])
        else:
            self.decoder_blocks = nn.Sequential(*[ResidualAttentionBlock(decoder_embed_dim, decoder_num_heads, mlp_ratio,
                                                                         norm_layer=norm_layer)
                                                  for _ in range(decoder_depth)])
        self.decoder_norm = norm_layer(decoder_embed_dim)
        self.decoder_pred = nn.Linear(decoder_embed_dim, self.pred_len, bias=True)
        self.initialize_weights()

    def initialize_weights(self):
        if self.choice_seq:
            pe = position_encoding_init(self.seq_len, self.decoder_pos_embed.shape[-1])
        else:
            pe = get_2d_sincos_pos_embed(self.decoder_pos_embed.shape[-1], int(self.num_patches ** .5), cls_token=True)
        self.decoder_pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        torch.nn.init.normal_(self.mask_token, std=.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _unshuffle(self, x, ids_restore):
        """modules/module_mae.py:310-314 / 338-342: append mask tokens, un-shuffle, add the positional table - one kernel
        (ops.MaeUnshuffleFn) where its preconditions hold, the op-by-op form otherwise."""
        B, Kk, Dd = x.shape
        Lq = ids_restore.shape[1]
        if (x.dtype == torch.float32 and Dd % 4 == 0 and ids_restore.dtype == torch.int64 and Kk <= Lq
                and self.decoder_pos_embed.shape[-2] == Lq):
            return ops.MaeUnshuffleFn.apply(x, self.mask_token, ids_restore, self.decoder_pos_embed)
        x_ = torch.cat([x, self.mask_token.float().expand(B, Lq - Kk, Dd)], dim=1)
        return ops.GatherRowsFn.apply(x_, ids_restore) + self.decoder_pos_embed.float()

    def forward_vis(self, image, vis_hidden, vis_mae_mask, vis_mae_ids_restore, loss_allpatch=False):
        if loss_allpatch:
            raise NotImplementedError("loss_allpatch=True is never used by the reference forward")
        ad = config.compute_dtype
        B, Kk, _ = vis_hidden.shape
        Lq = vis_mae_ids_restore.shape[1]
        x = ops.linear(vis_hidden.float(), self.decoder_embed.weight, self.decoder_embed.bias, out_dtype=torch.float32,
                       act_dtype=ad)
        x = self._unshuffle(x, vis_mae_ids_restore)
        for blk in self.decoder_blocks:
            x = blk(x)
        x = ops.layer_norm(x, self.decoder_norm.weight, self.decoder_norm.bias, self.decoder_norm.eps, ad)
        pred = ops.linear(x, self.decoder_pred.weight, self.decoder_pred.bias, out_dtype=torch.float32, act_dtype=ad)
        target = patchify(image, self.patch_size)
        return ops.MaskedMSEFn.apply(pred, target, vis_mae_mask.float().contiguous())

    def forward_seq(self, input_ids, seq_hidden, seq_mae_mask, seq_mae_ids_restore, attention_mask):
        """modules/module_mae.py:332-355: embed -> append mask tokens -> un-shuffle -> + sinusoid table -> 3 blocks with
        the key-padding mask -> LayerNorm -> vocabulary projection -> cross entropy on the masked, non-padding tokens
        (labels -1 elsewhere, ignore_index=-1)."""
        ad = config.compute_dtype
        B, Kk, _ = seq_hidden.shape
        Lq = seq_mae_ids_restore.shape[1]
        x = ops.linear(seq_hidden.float(), self.decoder_embed.weight, self.decoder_embed.bias, out_dtype=torch.float32,
                       act_dtype=ad)
        x = self._unshuffle(x, seq_mae_ids_restore)
        key_len = ops.prefix_mask_lengths(attention_mask)
        for blk in self.decoder_blocks:
            x = blk(x, key_len)
        x = ops.layer_norm(x, self.decoder_norm.weight, self.decoder_norm.bias, self.decoder_norm.eps, ad)
        pred = ops.linear(x, self.decoder_pred.weight, self.decoder_pred.bias, out_dtype=torch.float32, act_dtype=ad)
        m = seq_mae_mask.reshape(-1).to(input_ids.dtype)
        labels = input_ids.reshape(-1) * m - (1 - m)
        return ops.CrossEntropyLabelsFn.apply(pred.reshape(-1, self.pred_len), labels, -1)
