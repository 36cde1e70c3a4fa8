"""ORACLE - TEST INFRASTRUCTURE ONLY (not product code, never imported by segclip_amd/).

CPU restatement (plain torch-CPU fp32 eager ops, functional style) of the SegCLIP image-text
contrastive forward path: ViT front end, residual attention blocks, learnable-center aggregation,
text transformer, contrastive / superpixel-KL / MAE losses.  Gradients come from torch autograd
over these plain ops.  Every function cites the reference file:line it restates (paths relative to
/root/reference).

Pinning: validated against the *real* reference imported in the build container
(oracle/ref_harness.py, tests/golden/make_golden.py) and against the committed golden vectors
tests/golden/*.npz (tests/test_oracle_golden.py).  Two third-party behaviours on the path are not
under /root/reference and are "parity unpinned" (SURVEY.md 8c): diffdist.all_gather (restated as
all-gather fwd / reduce-scatter-sum bwd) and torch-1.8's MultiheadAttention key reshape
(cross_mode="t18", restated as a buffer reinterpretation).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Parameters are passed as a dict P keyed by the reference's state_dict names.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------------
def layer_norm(x, w, b, eps=1e-5):
    """modules/module_clip_util.py:126-132 (fp32 LayerNorm, eps 1e-5; MAE decoder uses 1e-6)."""
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def quick_gelu(x):
    """modules/module_clip_util.py:134-136."""
    return x * torch.sigmoid(1.702 * x)


def gelu_erf(x):
    """nn.GELU() default (erf form) used by Mlp, modules/module_seg_vit.py:128."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def mha_core(q, k, v, n_head, causal=False, key_mask=None):
    """softmax(q k^T / sqrt(hd)) v per head.  q (B,Tq,D), k/v (B,Tk,D) already projected.
    Restates nn.MultiheadAttention's core as used at modules/module_seg_vit.py:189 and
    modules/module_clip_ttransformer.py:46 (additive -inf upper-triangular mask from
    modules/module_clip_util.py:199-205 when causal)."""
    B, Tq, D = q.shape
    Tk = k.shape[1]
    hd = D // n_head
    qh = q.reshape(B, Tq, n_head, hd).permute(0, 2, 1, 3)
    kh = k.reshape(B, Tk, n_head, hd).permute(0, 2, 1, 3)
    vh = v.reshape(B, Tk, n_head, hd).permute(0, 2, 1, 3)
    s = (qh @ kh.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    if causal:
        mask = torch.full((Tq, Tk), float("-inf"), dtype=s.dtype, device=s.device).triu_(1)
        s = s + mask
    if key_mask is not None:   # (B, Tk) 1 = attend: additive (1 - mask) * -1e6 (modules/module_mae.py:216-219)
        s = s + ((1.0 - key_mask.to(s.dtype)) * -1000000.0)[:, None, None, :]
    p = torch.softmax(s, dim=-1)
    o = p @ vh
    return o.permute(0, 2, 1, 3).reshape(B, Tq, D)


def residual_attention_block(x, P, pre, n_head, causal=False):
    """modules/module_seg_vit.py:162-196 (vision) / modules/module_clip_ttransformer.py:20-52 (text).
    x (B,T,D) -> (B,T,D)."""
    D = x.shape[-1]
    y = layer_norm(x, P[pre + "ln_1.weight"], P[pre + "ln_1.bias"])
    qkv = y @ P[pre + "attn.in_proj_weight"].t() + P[pre + "attn.in_proj_bias"]
    q, k, v = qkv.split(D, dim=-1)
    o = mha_core(q, k, v, n_head, causal)
    x = x + (o @ P[pre + "attn.out_proj.weight"].t() + P[pre + "attn.out_proj.bias"])
    z = layer_norm(x, P[pre + "ln_2.weight"], P[pre + "ln_2.bias"])
    h = quick_gelu(z @ P[pre + "mlp.c_fc.weight"].t() + P[pre + "mlp.c_fc.bias"])
    x = x + (h @ P[pre + "mlp.c_proj.weight"].t() + P[pre + "mlp.c_proj.bias"])
    return x


def cross_attention_block(q, kv, P, pre, n_head, cross_mode="t18"):
    """modules/module_seg_vit.py:199-218.  q (B,G,D), kv (B,S,D).
    cross_mode "t18": torch-1.8 reshapes the un-permuted (B,S,D) key buffer as (S,B,D)
    (SURVEY.md finding 0.4) -> sample b attends to flat tokens {r*B+b}.  "intended": own tokens."""
    B, S, D = kv.shape
    w, bias = P[pre + "attn.in_proj_weight"], P[pre + "attn.in_proj_bias"]
    qn = layer_norm(q, P[pre + "ln_x.weight"], P[pre + "ln_x.bias"])
    kn = layer_norm(kv, P[pre + "ln_k.weight"], P[pre + "ln_k.bias"])
    qp = qn @ w[:D].t() + bias[:D]
    kp = kn @ w[D:2 * D].t() + bias[D:2 * D]
    vp = kn @ w[2 * D:].t() + bias[2 * D:]
    if cross_mode == "t18":
        kp = kp.reshape(S, B, D).permute(1, 0, 2)
        vp = vp.reshape(S, B, D).permute(1, 0, 2)
    o = mha_core(qp, kp, vp, n_head, causal=False)
    q = q + (o @ P[pre + "attn.out_proj.weight"].t() + P[pre + "attn.out_proj.bias"])
    z = layer_norm(q, P[pre + "ln_2.weight"], P[pre + "ln_2.bias"])
    h = quick_gelu(z @ P[pre + "mlp.c_fc.weight"].t() + P[pre + "mlp.c_fc.bias"])
    return q + (h @ P[pre + "mlp.c_proj.weight"].t() + P[pre + "mlp.c_proj.bias"])


def gumbel_softmax_hard(logits, gumbel, tau, dim):
    """modules/module_seg_vit.py:221-242 with hard=True.  gumbel=None <=> eval (no noise, no tau)."""
    if gumbel is not None:
        y_soft = ((logits + gumbel) / tau).softmax(dim)
    else:
        y_soft = logits.softmax(dim)
    index = y_soft.max(dim, keepdim=True)[1]
    y_hard = torch.zeros_like(logits).scatter_(dim, index, 1.0)
    return y_hard - y_soft.detach() + y_soft, index


def semantic_learner(x, P, pre, n_head, gumbel, cross_mode="t18"):
    """SemanticLearnerModule.forward, modules/module_seg_vit.py:277-314.
    x (B,T,D) -> out (B,G,D), hard (B,G,T), soft (B,G,T), q (B,G,D), hard_idx (B,T) int64."""
    B, T, D = x.shape
    hd = D // n_head
    n = layer_norm(x, P[pre + "norm.weight"], P[pre + "norm.bias"])
    q = P[pre + "semantic_center"].unsqueeze(0).repeat(B, 1, 1)
    i = 0
    while (pre + f"cross_att.{i}.ln_x.weight") in P:
        kv = torch.cat([q, x], dim=1)
        q = cross_attention_block(q, kv, P, pre + f"cross_att.{i}.", n_head, cross_mode)
        i += 1
    q = layer_norm(q, P[pre + "cross_ln.weight"], P[pre + "cross_ln.bias"])
    # grouped 1x1 Conv1d(D, D, groups=n_head, bias=False): weight (D, hd, 1); block-diagonal linear
    wk = P[pre + "k_conv.weight"].reshape(n_head, hd, hd)
    wv = P[pre + "v_conv.weight"].reshape(n_head, hd, hd)
    ng = n.reshape(B, T, n_head, hd)
    k = torch.einsum("btgi,goi->btgo", ng, wk).reshape(B, T, D)
    v = torch.einsum("btgi,goi->btgo", ng, wv).reshape(B, T, D)
    k = layer_norm(k, P[pre + "k_ln.weight"], P[pre + "k_ln.bias"])
    attn = q @ k.transpose(1, 2)  # (B,G,T), un-scaled (module_seg_vit.py:304)
    hard, index = gumbel_softmax_hard(attn, gumbel, 0.9, dim=1)
    soft = attn.softmax(dim=1)
    out = hard @ v
    out = out / torch.clamp_min(hard.sum(dim=-1, keepdim=True), 1.0)
    z = layer_norm(q + out, P[pre + "proj_o.ln.weight"], P[pre + "proj_o.ln.bias"])
    h = gelu_erf(z @ P[pre + "proj_o.mlp.fc1.weight"].t() + P[pre + "proj_o.mlp.fc1.bias"])
    o = quick_gelu(h @ P[pre + "proj_o.mlp.fc2.weight"].t() + P[pre + "proj_o.mlp.fc2.bias"])
    return o, hard, soft, q, index.squeeze(1)


def reconstruct_layer(sx, hard, P, pre):
    """ReconstructLayer.forward, modules/module_seg_vit.py:333-345.  sx (B,G,D), hard (B,G,M) -> (B,M,D)."""
    a = hard.permute(0, 2, 1) @ P[pre + "rec_proj_a.a_fc.weight"].t() + P[pre + "rec_proj_a.a_fc.bias"]
    return quick_gelu(a @ sx)


def random_masking(x, noise, mask_ratio):
    """modules/module_clip_util.py:91-124 with keep_cls=True; `noise` is the injected rand(N,L)."""
    N, L, D = x.shape
    len_keep = int(L * (1 - mask_ratio))
    noise = noise.clone()
    noise[:, 0] = -1.0
    ids_shuffle = torch.argsort(noise, dim=1)
    ids_restore = torch.argsort(ids_shuffle, dim=1)
    ids_keep = ids_shuffle[:, :len_keep]
    x_masked = torch.gather(x, 1, ids_keep.unsqueeze(-1).repeat(1, 1, D))
    mask = torch.ones(N, L, dtype=x.dtype, device=x.device)
    mask[:, :len_keep] = 0
    mask = torch.gather(mask, 1, ids_restore)
    return x_masked, mask, ids_restore, ids_keep


def patch_embed(image, P, patch):
    """conv1 16x16/16 no-bias as im2col GEMM, modules/module_clip_vtransformer.py:21,56-61."""
    B, C, H, W = image.shape
    gh, gw = H // patch, W // patch
    cols = image.reshape(B, C, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * patch * patch)
    w = P["clip.visual.conv1.weight"].reshape(P["clip.visual.conv1.weight"].shape[0], -1)
    return cols @ w.t()


def interp_pos_embed(pos, h, w):
    """VisualTransformer.get_pos_embed in eval mode, modules/module_clip_vtransformer.py:35-53: class row kept, the
    n x n patch rows resampled to h x w with F.interpolate(mode='bicubic', align_corners=False).  The bicubic
    resampling is restated from its definition (cubic convolution, A = -0.75, source coordinate
    (dst + 0.5) * n/out - 0.5, taps clamped to the border) in float32 loops over the (small) output grid."""
    import math
    import numpy as np
    tab = pos.detach().numpy().astype(np.float32)
    n = int(round(math.sqrt(tab.shape[0] - 1)))
    if h * w == n * n and h == w:
        return pos
    D = tab.shape[1]
    grid = tab[1:].reshape(n, n, D)
    A = np.float32(-0.75)

    def weights(t):
        t = np.float32(t)
        x0, x1, x2, x3 = t + np.float32(1), t, np.float32(1) - t, np.float32(2) - t
        return [((A * x0 - 5 * A) * x0 + 8 * A) * x0 - 4 * A, ((A + 2) * x1 - (A + 3)) * x1 * x1 + 1,
                ((A + 2) * x2 - (A + 3)) * x2 * x2 + 1, ((A * x3 - 5 * A) * x3 + 8 * A) * x3 - 4 * A]

    out = np.zeros((h, w, D), np.float32)
    sy, sx = np.float32(n) / np.float32(h), np.float32(n) / np.float32(w)
    for oy in range(h):
        fy = sy * np.float32(oy + 0.5) - np.float32(0.5)
        iy = int(math.floor(fy))
        wy = weights(fy - iy)
        for ox in range(w):
            fx = sx * np.float32(ox + 0.5) - np.float32(0.5)
            ix = int(math.floor(fx))
            wx = weights(fx - ix)
            acc = np.zeros(D, np.float32)
            for a in range(4):
                yy = min(max(iy - 1 + a, 0), n - 1)
                row = np.zeros(D, np.float32)
                for b in range(4):
                    xx = min(max(ix - 1 + b, 0), n - 1)
                    row += np.float32(wx[b]) * grid[yy, xx]
                acc += np.float32(wy[a]) * row
            out[oy, ox] = acc
    return torch.from_numpy(np.concatenate([tab[:1], out.reshape(h * w, D)], axis=0))


# ----------------------------------------------------------------------------------------------
# towers
# ----------------------------------------------------------------------------------------------
def encode_image(image, P, spec, gumbel=None, mask_noise=None, mask_ratio=0.0, cross_mode="t18",
                 first_stage_layer=10, eval_pos_interp=False):
    """CLIP.encode_image(return_hidden=True), modules/module_clip.py:81-103 ->
    VisualTransformer.forward, modules/module_clip_vtransformer.py:55-80 ->
    SegViT.forward, modules/module_seg_vit.py:403-452.  Training-mode positional table (raw) unless
    eval_pos_interp: then the eval-mode table of get_pos_embed (resampled when the patch grid differs)."""
    V = "clip.visual."
    T_ = V + "transformer."
    D = spec["vision_width"]
    n_head = D // 64
    x = patch_embed(image, P, spec["patch"])
    B = x.shape[0]
    cls = P[V + "class_embedding"].reshape(1, 1, D).expand(B, 1, D)
    pos = P[V + "positional_embedding"]
    if eval_pos_interp:
        pos = interp_pos_embed(pos, image.shape[2] // spec["patch"], image.shape[3] // spec["patch"])
    x = torch.cat([cls, x], dim=1) + pos
    x = layer_norm(x, P[V + "ln_pre.weight"], P[V + "ln_pre.bias"])
    mae_mask = ids_restore = ids_keep = None
    if mask_ratio > 0:
        x, mae_mask, ids_restore, ids_keep = random_masking(x, mask_noise, mask_ratio)
    x_ = x[:, 1:]  # CLS row split off and discarded (module_seg_vit.py:419)
    for i in range(first_stage_layer):
        x_ = residual_attention_block(x_, P, T_ + f"layers0.{i}.", n_head)
    mid = {"hidden": None, "attns": []}
    n_patch = (spec["image_res"] // spec["patch"]) ** 2
    if x_.shape[1] != n_patch and x_.shape[1] != 4 * n_patch:  # MAE branch
        sx, hard, soft, _, idx = semantic_learner(x_, P, T_ + "semantic_layer2.", n_head, gumbel, cross_mode)
        x_ = reconstruct_layer(sx, hard, P, T_ + "reconstruct_layer2.")
        for i in range(12 - first_stage_layer):
            x_ = residual_attention_block(x_, P, T_ + f"layers_mae2.{i}.", n_head)
        mid["hidden"] = x_
        mid["hard_idx"] = idx
        x = torch.cat([x_.mean(dim=1, keepdim=True), x_], dim=1)
    else:
        mid["hidden"] = x_
        x_, hard, soft, _, idx = semantic_learner(x_, P, T_ + "semantic_layer2.", n_head, gumbel, cross_mode)
        for i in range(12 - first_stage_layer):
            x_ = residual_attention_block(x_, P, T_ + f"layers2.{i}.", n_head)
        x = torch.cat([x_.max(dim=1, keepdim=True)[0], x_], dim=1)
        mid["attns"].append({"soft_attn": soft, "hard_attn": hard})
        mid["hard_idx"] = idx
    hidden_ln = layer_norm(x, P[V + "ln_post.weight"], P[V + "ln_post.bias"])
    hidden = hidden_ln @ P[V + "proj"]
    return hidden[:, 0, :], hidden, mae_mask, ids_restore, mid


def encode_text(ids, P, spec):
    """CLIP.encode_text(return_hidden=True), modules/module_clip.py:105-143."""
    Wt = spec["text_width"]
    n_head = Wt // 64
    x = P["clip.token_embedding.weight"][ids] + P["clip.positional_embedding"][: ids.shape[1]]
    for i in range(spec["text_layers"]):
        x = residual_attention_block(x, P, f"clip.transformer.resblocks.{i}.", n_head, causal=True)
    hidden = layer_norm(x, P["clip.ln_final.weight"], P["clip.ln_final.bias"]) @ P["clip.text_projection"]
    eot = ids.argmax(dim=-1)
    return hidden[torch.arange(hidden.shape[0]), eot], hidden, eot


def random_masking_text(x, noise, mask_ratio, sep_pos):
    """modules/module_clip_util.py:91-124 with keep_cls=True, keep_sep=True as CLIP.encode_text calls it
    (modules/module_clip.py:118-120).  `noise.scatter_(dim=1, index=sep_pos.unsqueeze(0), value=-1)` receives a
    (1, N) index, so it pins the separator positions OF EVERY SAMPLE in ROW 0 only - restated as written."""
    N, L, D = x.shape
    len_keep = int(L * (1 - mask_ratio))
    noise = noise.clone()
    noise[:, 0] = -1.0
    noise[0, sep_pos] = -1.0
    ids_shuffle = torch.argsort(noise, dim=1, stable=True)   # torch's CPU sort is stable: ties keep index order
    ids_restore = torch.argsort(ids_shuffle, dim=1, stable=True)
    ids_keep = ids_shuffle[:, :len_keep]
    x_masked = torch.gather(x, 1, ids_keep.unsqueeze(-1).repeat(1, 1, D))
    mask = torch.ones(N, L, dtype=x.dtype, device=x.device)
    mask[:, :len_keep] = 0
    mask = torch.gather(mask, 1, ids_restore)
    return x_masked, mask, ids_restore, ids_keep


def encode_text_masked(ids, P, spec, noise, mask_ratio):
    """CLIP.encode_text(return_hidden=True, mask_ratio>0), modules/module_clip.py:105-143: the kept tokens run through
    the causal text tower IN SHUFFLED ORDER (ids_keep is not sorted), the pooled row is the argmax of the gathered ids."""
    Wt = spec["text_width"]
    n_head = Wt // 64
    x = P["clip.token_embedding.weight"][ids] + P["clip.positional_embedding"][: ids.shape[1]]
    x, mask, ids_restore, ids_keep = random_masking_text(x, noise, mask_ratio, ids.argmax(dim=-1))
    text = torch.gather(ids, 1, ids_keep)
    for i in range(spec["text_layers"]):
        x = residual_attention_block(x, P, f"clip.transformer.resblocks.{i}.", n_head, causal=True)
    hidden = layer_norm(x, P["clip.ln_final.weight"], P["clip.ln_final.bias"]) @ P["clip.text_projection"]
    return hidden[torch.arange(hidden.shape[0]), text.argmax(dim=-1)], hidden, mask, ids_restore


def position_encoding_init(n_position, d):
    """modules/module_mae.py:44-54 (row 0 all zeros, sin on even / cos on odd columns, exponent 2*i/d per COLUMN i)."""
    pos = torch.arange(n_position, dtype=torch.float64)[:, None]
    i = torch.arange(d, dtype=torch.float64)[None, :]
    enc = pos / torch.pow(torch.tensor(10000.0, dtype=torch.float64), 2 * i / d)
    out = enc.clone()
    out[1:, 0::2] = torch.sin(enc[1:, 0::2])
    out[1:, 1::2] = torch.cos(enc[1:, 1::2])
    out[0] = 0
    return out.float()


def text_mae_loss(ids, seq_hidden, mae_mask, ids_restore, attention_mask, P, n_head=8):
    """modules/modeling.py:226-236 + MAEDecoder.forward_seq, modules/module_mae.py:332-355; decoder blocks =
    modules/module_mae.py:203-232 (nn.MultiheadAttention with the additive key-padding mask, erf-GELU Mlp, LN 1e-5)."""
    M = "seq_mae_decoder."
    x = seq_hidden @ P[M + "decoder_embed.weight"].t() + P[M + "decoder_embed.bias"]
    B, K, Dd = x.shape
    L = ids_restore.shape[1]
    mtok = P[M + "mask_token"].reshape(1, 1, Dd).expand(B, L - K, Dd)
    x = torch.gather(torch.cat([x, mtok], dim=1), 1, ids_restore.unsqueeze(-1).repeat(1, 1, Dd))
    x = x + P[M + "decoder_pos_embed"].reshape(1, L, Dd)
    i = 0
    while (M + f"decoder_blocks.{i}.norm1.weight") in P:
        pre = M + f"decoder_blocks.{i}."
        y = layer_norm(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"])
        qkv = y @ P[pre + "attn.in_proj_weight"].t() + P[pre + "attn.in_proj_bias"]
        q, k, v = qkv.split(Dd, dim=-1)
        o = mha_core(q, k, v, n_head, key_mask=attention_mask)
        x = x + (o @ P[pre + "attn.out_proj.weight"].t() + P[pre + "attn.out_proj.bias"])
        z = layer_norm(x, P[pre + "norm2.weight"], P[pre + "norm2.bias"])
        h = gelu_erf(z @ P[pre + "mlp.fc1.weight"].t() + P[pre + "mlp.fc1.bias"])
        x = x + (h @ P[pre + "mlp.fc2.weight"].t() + P[pre + "mlp.fc2.bias"])
        i += 1
    x = layer_norm(x, P[M + "decoder_norm.weight"], P[M + "decoder_norm.bias"])
    pred = x @ P[M + "decoder_pred.weight"].t() + P[M + "decoder_pred.bias"]
    m = ((mae_mask + attention_mask.to(mae_mask.dtype)) > 1).reshape(-1).to(ids.dtype)   # masked AND not padding
    labels = ids.reshape(-1) * m - (1 - m)
    return torch.nn.functional.cross_entropy(pred.reshape(-1, pred.shape[-1]), labels, ignore_index=-1)


# ----------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------
def loose_similarity(text_feat, img_feat, logit_scale, gather=None):
    """SegCLIP._loose_similarity (training branch), modules/modeling.py:338-362.
    gather: callable (B,E)->(B*W,E) rank-ordered differentiable all-gather (dist_collect,
    modules/util_module.py:180-190); None <=> world size 1."""
    v = img_feat / img_feat.norm(dim=-1, keepdim=True)
    t = text_feat / text_feat.norm(dim=-1, keepdim=True)
    s = torch.clamp(logit_scale.exp(), max=100)
    v_all = gather(v) if gather is not None else v
    t_all = gather(t) if gather is not None else t
    return s * (t @ v_all.t()), s * (v @ t_all.t())


def contrastive_loss(t2v, v2t, rank=0):
    """modules/modeling.py:204-210."""
    B = t2v.shape[0]
    labels = torch.arange(B, dtype=torch.long, device=t2v.device) + B * rank
    return (F.cross_entropy(t2v, labels) + F.cross_entropy(v2t, labels)) / 2.0


def superpixel_kl(hard, image_seg):
    """modules/modeling.py:212-224.  hard (B,G,T); image_seg (B,gh,gw) int64."""
    B = hard.shape[0]
    h = hard.permute(0, 2, 1)  # (B,T,G)
    seg = image_seg.reshape(B, -1)
    eq = ((seg.unsqueeze(-1) - seg.unsqueeze(-2)) == 0).to(h.dtype)
    cm = (eq @ h) / torch.clamp_min(eq.sum(dim=-1, keepdim=True), 1.0)
    coef = float(h.shape[0] * h.shape[1] * h.shape[2])
    kl1 = F.kl_div(F.log_softmax(h, dim=-1), F.softmax(cm, dim=-1), reduction="sum") / coef
    kl2 = F.kl_div(F.log_softmax(cm, dim=-1), F.softmax(h, dim=-1), reduction="sum") / coef
    return (kl1 + kl2) / 2.0


def sincos_pos_embed_2d(embed_dim, grid_size):
    """get_2d_sincos_pos_embed(cls_token=True), modules/module_mae.py:63-108 (float64 numpy there)."""
    gw = torch.arange(grid_size, dtype=torch.float64)
    gh = torch.arange(grid_size, dtype=torch.float64)
    # np.meshgrid(grid_w, grid_h): grid[0][i,j] = w_j, grid[1][i,j] = h_i   ("w goes first")
    g0 = gw.unsqueeze(0).expand(grid_size, grid_size).reshape(-1)
    g1 = gh.unsqueeze(1).expand(grid_size, grid_size).reshape(-1)

    def one(d, pos):
        omega = torch.arange(d // 2, dtype=torch.float64) / (d / 2.0)
        omega = 1.0 / 10000 ** omega
        out = pos.unsqueeze(1) * omega.unsqueeze(0)
        return torch.cat([out.sin(), out.cos()], dim=1)

    emb = torch.cat([one(embed_dim // 2, g0), one(embed_dim // 2, g1)], dim=1)
    emb = torch.cat([torch.zeros(1, embed_dim, dtype=torch.float64), emb], dim=0)
    return emb.float()


def patchify(imgs, p):
    """modules/module_mae.py:18-29: (N,3,H,W) -> (N, L, p*p*3) with layout nchpwq->nhwpqc."""
    N = imgs.shape[0]
    h = w = imgs.shape[2] // p
    x = imgs.reshape(N, 3, h, p, w, p)
    x = torch.einsum("nchpwq->nhwpqc", x)
    return x.reshape(N, h * w, p * p * 3)


def mae_decoder_loss(image, vis_hidden, mask, ids_restore, P, spec, n_head=8):
    """MAEDecoder.forward_vis(loss_allpatch=False), modules/module_mae.py:304-330; decoder Block
    (timm style: qkv-bias, LN eps 1e-6, erf GELU), modules/module_mae.py:110-134,185-201."""
    M = "vis_mae_decoder."
    x = vis_hidden @ P[M + "decoder_embed.weight"].t() + P[M + "decoder_embed.bias"]
    B, K, Dd = x.shape
    L = ids_restore.shape[1]
    mtok = P[M + "mask_token"].reshape(1, 1, Dd).expand(B, L - K, Dd)
    x_ = torch.cat([x, mtok], dim=1)
    x = torch.gather(x_, 1, ids_restore.unsqueeze(-1).repeat(1, 1, Dd))
    x = x + P[M + "decoder_pos_embed"].reshape(1, L, Dd)
    i = 0
    while (M + f"decoder_blocks.{i}.norm1.weight") in P:
        pre = M + f"decoder_blocks.{i}."
        y = layer_norm(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"], 1e-6)
        qkv = y @ P[pre + "attn.qkv.weight"].t() + P[pre + "attn.qkv.bias"]
        q, k, v = qkv.split(Dd, dim=-1)
        o = mha_core(q, k, v, n_head)
        x = x + (o @ P[pre + "attn.proj.weight"].t() + P[pre + "attn.proj.bias"])
        z = layer_norm(x, P[pre + "norm2.weight"], P[pre + "norm2.bias"], 1e-6)
        h = gelu_erf(z @ P[pre + "mlp.fc1.weight"].t() + P[pre + "mlp.fc1.bias"])
        x = x + (h @ P[pre + "mlp.fc2.weight"].t() + P[pre + "mlp.fc2.bias"])
        i += 1
    x = layer_norm(x, P[M + "decoder_norm.weight"], P[M + "decoder_norm.bias"], 1e-6)
    pred = (x @ P[M + "decoder_pred.weight"].t() + P[M + "decoder_pred.bias"])[:, 1:, :]
    target = patchify(image, spec["patch"])
    loss = ((pred - target) ** 2).mean(dim=-1)
    m = mask[:, 1:]
    return (loss * m).sum() / m.sum()


# ----------------------------------------------------------------------------------------------
# SegCLIP.forward (training)
# ----------------------------------------------------------------------------------------------
def segclip_forward(batch, P, spec, noise, flags, rank=0, gather=None, cross_mode="t18"):
    """SegCLIP.forward in training mode, modules/modeling.py:174-256 (incl. the text-MAE branch :226-236).
    Returns (loss, aux) where aux holds every intermediate the parity tests compare."""
    ids = batch["input_ids"].reshape(-1, batch["input_ids"].shape[-1])
    image = batch["image"].float()[:, 0]
    aux = {}
    t_feat, t_hidden, eot = encode_text(ids, P, spec)
    v_feat, v_hidden, _, _, mid = encode_image(image, P, spec, gumbel=noise.get("gumbel_main"),
                                               cross_mode=cross_mode)
    t2v, v2t = loose_similarity(t_feat, v_feat, P["clip.logit_scale"], gather)
    l_con = contrastive_loss(t2v, v2t, rank)
    loss = l_con
    aux.update(text_feat=t_feat, image_feat=v_feat, text_hidden=t_hidden, image_hidden=v_hidden,
               eot=eot, t2v=t2v, v2t=v2t, loss_contrastive=l_con, hard_idx=mid["hard_idx"],
               hard=mid["attns"][0]["hard_attn"], soft=mid["attns"][0]["soft_attn"],
               layers0_out=mid["hidden"])
    if flags.get("use_seglabel", False):
        l_kl = superpixel_kl(mid["attns"][0]["hard_attn"], batch["image_seg"][:, 0])
        loss = loss + l_kl
        aux["loss_kl"] = l_kl
    if flags.get("use_text_mae_recon", False):
        amask = batch["input_mask"].reshape(-1, batch["input_mask"].shape[-1])
        _, s_hidden, s_mask, s_restore = encode_text_masked(ids, P, spec, noise["text_mask_noise"],
                                                            flags.get("mae_seq_mask_ratio", 0.15))
        l_seq = text_mae_loss(ids, s_hidden, s_mask, s_restore, amask, P)
        loss = loss + l_seq
        aux.update(loss_text_mae=l_seq, text_mae_mask=s_mask, text_ids_restore=s_restore, text_mae_hidden=s_hidden)
    if flags.get("use_vision_mae_recon", False):
        _, _, m_mask, m_restore, m_mid = encode_image(image, P, spec, gumbel=noise.get("gumbel_mae"),
                                                      mask_noise=noise["mask_noise"], mask_ratio=0.75,
                                                      cross_mode=cross_mode)
        vh = m_mid["hidden"]
        vh = torch.cat([vh.mean(dim=1, keepdim=True), vh], dim=1)
        l_mae = mae_decoder_loss(image, vh, m_mask, m_restore, P, spec)
        loss = loss + l_mae
        aux.update(loss_mae=l_mae, mae_mask=m_mask, ids_restore=m_restore, mae_hard_idx=m_mid["hard_idx"],
                   mae_hidden=m_mid["hidden"])
    aux["loss"] = loss
    return loss, aux


def params_from_module(model, requires_grad=True):
    """state_dict-keyed leaf tensors (detached clones) of a reference-shaped module tree."""
    frozen = {n for n, p in model.named_parameters() if not p.requires_grad}
    P = {}
    for k, v in model.state_dict().items():
        t = v.detach().clone().float()
        P[k] = t.requires_grad_(requires_grad and t.is_floating_point() and k not in frozen)
    return P
