"""Mirror of modules/module_seg_vit.py: ResidualAttentionBlock, CrossAttentionBlock,
SemanticLearnerModule (learnable-center aggregation), ReconstructLayer, SegViT."""
from collections import OrderedDict

import os

import torch
from torch import nn
from torch.nn.parameter import Parameter

from .. import _lib as L
from .. import config, ops
from .module_clip_util import LayerNorm, QuickGELU


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class Mlp(nn.Module):
    """modules/module_seg_vit.py:127-143 (fc1 -> erf GELU -> fc2)."""

    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x, final_act=ops.ACT_NONE, out_dtype=None):
        h = ops.linear(x, self.fc1.weight, self.fc1.bias, act=ops.ACT_GELU_ERF, act_dtype=config.compute_dtype)
        return ops.linear(h, self.fc2.weight, self.fc2.bias, act=final_act, out_dtype=out_dtype,
                          act_dtype=config.compute_dtype)


def _mlp_sequential(d_model):
    return nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                      ("c_proj", nn.Linear(d_model * 4, d_model))]))


class ResidualAttentionBlock(nn.Module):
    """modules/module_seg_vit.py:162-196.  forward takes/returns NLD (B,T,D) like the reference."""

    def __init__(self, d_model: int, n_head: int, drop_path: float = 0.):
        super().__init__()
        self.n_head = n_head
        self.attn = nn.MultiheadAttention(d_model, n_head)  # parameter container (in_proj_*, out_proj.*)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = _mlp_sequential(d_model)
        self.ln_2 = LayerNorm(d_model)
        self.causal = False

    def block_params(self):
        """The 12 parameters in ops.ResBlockFn / ops.ResStackFn order."""
        return (self.ln_1.weight, self.ln_1.bias, self.attn.in_proj_weight, self.attn.in_proj_bias,
                self.attn.out_proj.weight, self.attn.out_proj.bias, self.ln_2.weight, self.ln_2.bias,
                self.mlp.c_fc.weight, self.mlp.c_fc.bias, self.mlp.c_proj.weight, self.mlp.c_proj.bias)

    def forward(self, x, attn_mask=None, video_frame=-1):
        if attn_mask is not None and not callable(attn_mask):
            raise NotImplementedError("padding masks are only reachable from the text-MAE branch (out of scope)")
        return ops.ResBlockFn.apply(x.float(), self.ln_1.weight, self.ln_1.bias, self.attn.in_proj_weight,
                                    self.attn.in_proj_bias, self.attn.out_proj.weight, self.attn.out_proj.bias,
                                    self.ln_2.weight, self.ln_2.bias, self.mlp.c_fc.weight, self.mlp.c_fc.bias,
                                    self.mlp.c_proj.weight, self.mlp.c_proj.bias, self.n_head,
                                    self.causal or callable(attn_mask), ops.ACT_QUICK_GELU, self.ln_1.eps,
                                    config.compute_dtype)


# one autograd node per cross-attention block (ops.CrossBlockFn); SEGCLIP_CROSS_FUSED=0: the op-by-op composition
_CROSS_FUSED = os.environ.get("SEGCLIP_CROSS_FUSED", "1") != "0"


class CrossAttentionBlock(nn.Module):
    """modules/module_seg_vit.py:199-218: q += MHA(ln_x q, ln_k kv, ln_k kv); q += mlp(ln_2 q)."""

    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        self.n_head = n_head
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_x = LayerNorm(d_model)
        self.ln_k = LayerNorm(d_model)
        self.mlp = _mlp_sequential(d_model)
        self.ln_2 = LayerNorm(d_model)

    def forward(self, q, k, k_tail=None, kn_buf=None):
        """The key/value input is k, or cat([k, k_tail], dim=1) when k_tail is given - never materialised: ln_k writes
        both parts into one buffer (ops.LayerNormCatFn), or, with kn_buf (B, S, D) whose tail rows already hold
        ln_k(k_tail) (ops.LayerNormMultiFn), only the head rows ln_k(k)."""
        B, G, D = q.shape
        ad = config.compute_dtype
        w, b = self.attn.in_proj_weight, self.attn.in_proj_bias
        if kn_buf is not None and _CROSS_FUSED:
            return ops.CrossBlockFn.apply(q, kn_buf, self.ln_x.weight, self.ln_x.bias, self.ln_k.weight, self.ln_k.bias, w, b,
                                          self.attn.out_proj.weight, self.attn.out_proj.bias, self.ln_2.weight, self.ln_2.bias,
                                          self.mlp.c_fc.weight, self.mlp.c_fc.bias, self.mlp.c_proj.weight, self.mlp.c_proj.bias,
                                          self.n_head, config.cross_mode, (self.ln_x.eps, self.ln_k.eps, self.ln_2.eps), ad)[0]
        qn = self.ln_x(q)
        if kn_buf is not None:
            kn = ops.layer_norm_into(kn_buf, k, self.ln_k.weight, self.ln_k.bias, self.ln_k.eps, 0)
        elif k_tail is None:
            kn = self.ln_k(k)
        else:
            kn = ops.layer_norm_cat(k, k_tail, self.ln_k.weight, self.ln_k.bias, self.ln_k.eps, ad)
        S = kn.shape[1]
        o = ops.CrossInProjAttnFn.apply(qn, kn, w, b, B, G, S, self.n_head, config.cross_mode, ad).view(B, G, D)
        q = ops.linear(o, self.attn.out_proj.weight, self.attn.out_proj.bias, residual=q, out_dtype=torch.float32,
                       act_dtype=ad)
        z = self.ln_2(q)
        h = ops.linear(z, self.mlp.c_fc.weight, self.mlp.c_fc.bias, act=ops.ACT_QUICK_GELU, act_dtype=ad)
        return ops.linear(h, self.mlp.c_proj.weight, self.mlp.c_proj.bias, residual=q, out_dtype=torch.float32,
                          act_dtype=ad)


def gumbel_softmax(logits, tau=1, hard=False, dim=-1, is_training=True):
    """modules/module_seg_vit.py:221-242 for (hard=True, dim=1) on (B,G,T) logits."""
    if not hard or dim not in (1, -2) or logits.dim() != 3:
        raise NotImplementedError("only the hard, dim=1 use of the hot path is implemented")
    g = config.gumbel(tuple(logits.shape), logits.device) if is_training else None
    return ops.AssignFn.apply(logits.float(), g, float(tau))[0]


def _group_linear(n2d, w, groups):
    """Conv1d(D, D, kernel 1, groups, bias=False) on channel-last rows: block-diagonal linear
    (modules/module_seg_vit.py:266,269,299,302).  n2d (M, D); w (D, D/groups, 1)."""
    return GroupLinearFn.apply(n2d, w, groups)


class GroupLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, groups):
        M, D = x.shape
        hd = D // groups
        wc = ops.wcast(w.reshape(D, hd), x.dtype)
        y = torch.empty_like(x)
        ops.p_gemm(x, wc, y, M, hd, hd, (D, 1), (hd, 1), D, nb1=groups, bsA=(hd, 0), bsB=(hd * hd, 0), bsC=(hd, 0))
        ctx.save_for_backward(x, wc)
        ctx.groups, ctx.wshape = groups, tuple(w.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wc = ctx.saved_tensors
        M, D = x.shape
        G = ctx.groups
        hd = D // G
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            ops.p_gemm(dy, wc, dx, M, hd, hd, (D, 1), (1, hd), D, nb1=G, bsA=(hd, 0), bsB=(hd * hd, 0), bsC=(hd, 0))
        if ctx.needs_input_grad[1]:
            dw = torch.empty((D, hd), dtype=torch.float32, device=x.device)
            ops.p_gemm(dy, x, dw, hd, hd, M, (1, D), (1, D), hd, nb1=G, bsA=(hd, 0), bsB=(hd, 0), bsC=(hd * hd, 0))
            dw = dw.view(ctx.wshape)
        return dx, dw, None


def _gl64(ins, outs, ws, M, groups):
    """segclip_group_linear64: out_o = sum_i in_i x ws[i * len(outs) + o] per 64-channel group (bf16 rows)."""
    import ctypes as C
    L.require_cuda(*ins, *outs, *ws)
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    lds = lambda ts: (C.c_int64 * len(ts))(*[t.stride(0) for t in ts])
    L.check(L.load().segclip_group_linear64(C.cast(arr(ins), C.c_void_p), C.cast(lds(ins), C.c_void_p), len(ins),
                                            C.cast(arr(outs), C.c_void_p), C.cast(lds(outs), C.c_void_p), len(outs),
                                            C.cast(arr(ws), C.c_void_p), M, groups, L.stream()), "group_linear64")


class GroupLinearPairFn(torch.autograd.Function):
    """k_conv and v_conv of the learnable-center stage (two grouped kernel-1 Conv1d's of the SAME input,
    modules/module_seg_vit.py:299,302) as ONE pass over the rows, and their data gradient dn = dk Wk + dv Wv as one
    pass (segclip_group_linear64; bf16 mode, 64 channels per group).  Weight gradients: the batched GEMM path."""

    @staticmethod
    def forward(ctx, x, wk, wv, groups):
        M, D = x.shape
        hd = D // groups
        wkc, wvc = ops.wcast(wk.reshape(D, hd), x.dtype), ops.wcast(wv.reshape(D, hd), x.dtype)
        k, v = torch.empty_like(x), torch.empty_like(x)
        _gl64((x,), (k, v), (wkc, wvc), M, groups)
        ctx.save_for_backward(x, wkc, wvc)
        ctx.groups, ctx.wshape = groups, tuple(wk.shape)
        return k, v

    @staticmethod
    def backward(ctx, dk, dv):
        x, wkc, wvc = ctx.saved_tensors
        M, D = x.shape
        G = ctx.groups
        hd = D // G
        dk, dv = dk.contiguous(), dv.contiguous()
        dx = dwk = dwv = None
        if ctx.needs_input_grad[0]:
            # per-group transposed weights (96 KiB each): dn(m, g*64 + k) = sum_n dk(m, g*64 + n) Wk[g*64 + n][k] + (same, v)
            wkt = wkc.view(G, hd, hd).transpose(1, 2).contiguous().view(D, hd)
            wvt = wvc.view(G, hd, hd).transpose(1, 2).contiguous().view(D, hd)
            dx = torch.empty_like(x)
            _gl64((dk, dv), (dx,), (wkt, wvt), M, G)
        for need, dy, name in ((ctx.needs_input_grad[1], dk, "k"), (ctx.needs_input_grad[2], dv, "v")):
            if need:
                dw = torch.empty((D, hd), dtype=torch.float32, device=x.device)
                ops.p_gemm(dy, x, dw, hd, hd, M, (1, D), (1, D), hd, nb1=G, bsA=(hd, 0), bsB=(hd, 0), bsC=(hd * hd, 0))
                if name == "k":
                    dwk = dw.view(ctx.wshape)
                else:
                    dwv = dw.view(ctx.wshape)
        return dx, dwk, dwv, None


_GL64 = __import__("os").environ.get("SEGCLIP_GL64", "1") != "0"   # 0: the two batched-GEMM launches (A/B tests)


def _group_linear_pair(n2d, wk, wv, groups):
    """(k_conv(n), v_conv(n)) on channel-last rows; one fused pass in bf16 mode with 64-channel groups."""
    M, D = n2d.shape
    if (_GL64 and n2d.dtype == torch.bfloat16 and D // groups == 64 and n2d.stride(1) == 1 and n2d.stride(0) % 8 == 0
            and n2d.data_ptr() % 16 == 0):
        return GroupLinearPairFn.apply(n2d, wk, wv, groups)
    return _group_linear(n2d, wk, groups), _group_linear(n2d, wv, groups)


class SemanticLearnerModule(nn.Module):
    """modules/module_seg_vit.py:244-314: learnable centers -> cross attention x2 -> hard (Gumbel)
    assignment of every patch to one center -> segment mean -> proj_o."""

    def __init__(self, in_channels, num_tokens, num_heads, cross_layer=1):
        super().__init__()
        self.in_channels = in_channels
        self.num_heads = num_heads
        self.norm = nn.LayerNorm(in_channels)
        self.semantic_center = Parameter(torch.Tensor(num_tokens, in_channels))
        trunc_normal_(self.semantic_center, std=.02)
        self.cross_att = nn.Sequential(OrderedDict(
            [(str(i), CrossAttentionBlock(in_channels, n_head=num_heads)) for i in range(cross_layer)]))
        self.cross_ln = nn.LayerNorm(in_channels)
        self.k_conv = nn.Conv1d(in_channels, in_channels, kernel_size=(1,), stride=(1,), padding=(0,),
                                groups=num_heads, bias=False)
        self.k_ln = nn.LayerNorm(in_channels)
        self.v_conv = nn.Conv1d(in_channels, in_channels, kernel_size=(1,), stride=(1,), padding=(0,),
                                groups=num_heads, bias=False)
        self.proj_o = nn.Sequential(OrderedDict([("ln", nn.LayerNorm(in_channels)),
                                                 ("mlp", Mlp(in_channels, 4 * in_channels, in_channels)),
                                                 ("act", QuickGELU())]))

    def forward(self, inputs):
        B, T, D = inputs.shape
        G = self.semantic_center.shape[0]
        ad = config.compute_dtype
        inputs = inputs.float()
        q = self.semantic_center.float().unsqueeze(0).expand(B, G, D).contiguous()
        # `self.norm(inputs)` and the token part of `ln_1(kv)`, kv = torch.cat([q, inputs], dim=1), of every cross layer
        # normalise the same rows: one kernel, three affine outputs (and one dx in the backward); kv is never built
        multi = None
        blocks = list(self.cross_att)
        if len(blocks) == 2 and all(blk.ln_k.eps == self.norm.eps for blk in blocks):
            multi = ops.layer_norm_multi(inputs.reshape(B * T, D),
                                         [(self.norm.weight, self.norm.bias)] + [(blk.ln_k.weight, blk.ln_k.bias) for blk in blocks],
                                         [None, (T, G + T, G), (T, G + T, G)], self.norm.eps, ad)
        if multi is not None:
            n = multi[0].view(B, T, D)
            for blk, buf in zip(blocks, multi[1:]):
                q = blk(q, q, kn_buf=buf)
        else:
            n = ops.layer_norm(inputs, self.norm.weight, self.norm.bias, self.norm.eps, ad)
            for blk in blocks:
                q = blk(q, q, inputs)
        q = ops.layer_norm(q, self.cross_ln.weight, self.cross_ln.bias, self.cross_ln.eps, torch.float32)
        n2 = n.view(B * T, D)
        k, v = _group_linear_pair(n2, self.k_conv.weight, self.v_conv.weight, self.num_heads)
        k = ops.layer_norm(k, self.k_ln.weight, self.k_ln.bias, self.k_ln.eps, torch.float32).view(B, T, D)
        v = v.view(B, T, D)
        # assignment logits always in exact fp32 (bit-exact argmax), un-scaled (module_seg_vit.py:304)
        attn = ops.center_logits(q, k, exact=ad != torch.bfloat16)
        g = config.gumbel((B, G, T), inputs.device) if self.training else None
        hard_attn, soft_attn, idx, counts = ops.AssignFn.apply(attn, g, 0.9)
        if G <= 8 and D <= 1024:      # segment mean by center index: one launch each way (csrc/center.hip)
            outputs = ops.SegMeanFn.apply(hard_attn, idx, counts, v)
        else:
            cnt = torch.clamp_min(hard_attn.sum(dim=-1, keepdim=True), 1.0)
            outputs = ops.bmm(hard_attn.to(ad), v, transB=False, out_dtype=torch.float32) / cnt
        z = ops.layer_norm(q + outputs, self.proj_o.ln.weight, self.proj_o.ln.bias, self.proj_o.ln.eps, ad)
        outputs = self.proj_o.mlp(z, final_act=ops.ACT_QUICK_GELU, out_dtype=torch.float32)
        self.last_hard_idx = idx
        return outputs, hard_attn, soft_attn, q


class ReconstructLayer(nn.Module):
    """modules/module_seg_vit.py:316-345: (B,G,D) centers + hard (B,G,M) -> (B,M,D)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.rec_proj_a = nn.Sequential(OrderedDict([("a_fc", nn.Linear(in_channels, in_channels))]))
        self.proj_o = nn.Sequential(OrderedDict([("act_a", QuickGELU())]))

    def forward(self, inputs, attn):
        B, G, D = inputs.shape
        M = attn.shape[2]
        # Linear(8,8) over the center axis: K = 8 is below the MFMA K granule -> exact f32 path always
        a = ops.linear(attn.permute(0, 2, 1).contiguous().float(), self.rec_proj_a.a_fc.weight,
                       self.rec_proj_a.a_fc.bias, act_dtype=torch.float32)          # (B,M,G)
        out = ops.recon_mix(a, inputs.float())                                       # (B,M,D) = a @ inputs, K = 8
        return ops.ActFn.apply(out, ops.ACT_QUICK_GELU)


class SegViT(nn.Module):
    """modules/module_seg_vit.py:348-452."""

    def __init__(self, dim_in, patch_size=32, input_resolution=224, first_stage_layer=10, cross_layer=2, group_num=8):
        super().__init__()
        self.dim_in = dim_in
        self.patch_len = input_resolution // patch_size
        depths = [first_stage_layer, 12 - first_stage_layer]
        heads = dim_in // 64
        self.layers0 = nn.Sequential(OrderedDict(
            [(str(i), ResidualAttentionBlock(dim_in, heads)) for i in range(depths[0])]))
        self.semantic_layer2 = SemanticLearnerModule(in_channels=dim_in, num_tokens=group_num, num_heads=heads,
                                                     cross_layer=cross_layer)
        if depths[1] > 0:
            self.layers2 = nn.Sequential(OrderedDict(
                [(str(i), ResidualAttentionBlock(dim_in, heads)) for i in range(depths[1])]))
            self.layers_mae2 = nn.Sequential(OrderedDict(
                [(str(i), ResidualAttentionBlock(dim_in, heads)) for i in range(depths[1])]))
        else:
            self.layers2 = nn.Identity()
            self.layers_mae2 = nn.Identity()
        self.reconstruct_layer2 = ReconstructLayer(in_channels=group_num, out_channels=dim_in)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @staticmethod
    def _run_blocks(seq, x):
        """nn.Sequential of ResidualAttentionBlocks as ONE autograd node (ops.ResStackFn) when there are several.
        A pending config.set_stack_hook(k, fn) cuts the stack after k blocks and runs fn() in between."""
        blocks = list(seq) if isinstance(seq, nn.Sequential) else None
        hook = config.take_stack_hook()
        if (config.fuse_res_stack and blocks and len(blocks) > 1
                and all(isinstance(b, ResidualAttentionBlock) and not b.causal for b in blocks)):
            b0 = blocks[0]

            def stack(bs, t, keep16=False):
                if len(bs) == 1:
                    return bs[0](t.float())
                if t.dtype != torch.bfloat16:      # (a bf16 stream handed over by the previous stack stays as it is)
                    t = t.float()
                return ops.res_stack(t, [b.block_params() for b in bs], b0.n_head, False, ops.ACT_QUICK_GELU,
                                     b0.ln_1.eps, config.compute_dtype, keep16=keep16)
            if hook is not None and 0 < hook[0] < len(blocks):
                # config.bf16_resid: the two parts exchange the bf16 residual stream directly (no fp32 round trip)
                x = stack(blocks[:hook[0]], x, keep16=len(blocks) - hook[0] > 1)
                hook[1]()
                return stack(blocks[hook[0]:], x)
            if hook is not None:
                hook[1]()
            return stack(blocks, x)
        if hook is not None:
            hook[1]()
        return seq(x)

    def forward_patches(self, x_, pooled_only=False):
        """Body of forward() on the patch tokens only: x_ (B,T,D) NLD without the CLS row.
        Returns (x (B,1+T',D) NLD with the pooled CLS prepended, mid_states); pooled_only (the training step's fast path, main
        branch only): (cls (B,D), mid_states) - the max over the tokens alone, without building the (B,1+T',D) tensor."""
        mid_states = {"hidden": None, "attns": []}
        x_ = self._run_blocks(self.layers0, x_)
        if self.patch_len ** 2 != x_.size(1) and 4 * (self.patch_len ** 2) != x_.size(1):  # MAE branch
            sx_, hard_attn_2, soft_attn_2, _ = self.semantic_layer2(x_)
            x_ = self.reconstruct_layer2(sx_, hard_attn_2)
            x_ = self._run_blocks(self.layers_mae2, x_)
            mid_states["hidden"] = x_
            x = ops.mean_cat(x_)
        else:
            mid_states["hidden"] = x_
            x_, hard_attn_2, soft_attn_2, _ = self.semantic_layer2(x_)
            x_ = self._run_blocks(self.layers2, x_)
            mid_states["attns"].append({"soft_attn": soft_attn_2, "hard_attn": hard_attn_2})
            if pooled_only:
                mid_states["hard_idx"] = self.semantic_layer2.last_hard_idx
                return ops.MaxTokensFn.apply(x_.float(), config.compute_dtype == torch.bfloat16), mid_states
            cls = torch.max(x_, dim=1, keepdim=True)[0]
            x = torch.cat([cls, x_], dim=1)
        mid_states["hard_idx"] = self.semantic_layer2.last_hard_idx
        return x, mid_states

    def forward(self, x, attn_mask=None, video_frame=-1):
        """Reference signature: x is LND including the CLS row, which is split off and discarded
        (modules/module_seg_vit.py:414-419)."""
        if attn_mask is not None:
            raise NotImplementedError
        x = x.permute(1, 0, 2)
        x, mid_states = self.forward_patches(x[:, 1:].contiguous())
        return x.permute(1, 0, 2), mid_states
