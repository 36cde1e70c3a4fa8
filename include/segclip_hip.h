/* libsegclip_hip.so - C ABI of the MI355X (gfx950) SegCLIP hot-path kernels.
 *
 * Boundary contract:
 *   - plain pointers + sizes, no torch / C++ types; every entry point is extern "C".
 *   - the caller owns every buffer (inputs, outputs, workspaces); the library never allocates,
 *     frees or retains device pointers and never synchronises the device.
 *   - kernels are enqueued on the hipStream_t passed as `stream` (void*).
 *   - return 0 on success, non-zero otherwise; segclip_last_error_string() describes the failure.
 *   - stateless and re-entrant (forward thread + autograd thread may call concurrently).
 *
 * Each entry point replaces an *implicit* torch op group of the reference (the reference is pure
 * Python and has no native interface of its own); the reference call site is cited per function
 * (paths relative to the ArrowLuo/SegCLIP tree).
 *
 * dtypes: SEGCLIP_F32 activations run the exact-f32 MFMA path (v_mfma_f32_32x32x2_f32, parity
 * gate 1e-3); SEGCLIP_BF16 activations run the bf16 MFMA path (v_mfma_f32_32x32x16_bf16,
 * fp32 accumulate).  Parameters / gradients of parameters are always fp32.
 */
#ifndef SEGCLIP_HIP_H
#define SEGCLIP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEGCLIP_ABI_VERSION 1

#define SEGCLIP_F32 0
#define SEGCLIP_BF16 1

#define SEGCLIP_ACT_NONE 0
#define SEGCLIP_ACT_QUICK_GELU 1 /* x*sigmoid(1.702x), modules/module_clip_util.py:134-136 */
#define SEGCLIP_ACT_GELU_ERF 2   /* nn.GELU(), modules/module_seg_vit.py:128 */

#define SEGCLIP_ERR_INVALID (-1)
#define SEGCLIP_ERR_UNSUPPORTED (-2)

int segclip_version(void);
const char* segclip_last_error_string(void);

/* ------------------------------------------------------------------------------------------
 * GEMM with fused epilogue.   C[z](m,n) = epi( alpha * sum_k A[z](m,k) * B[z](n,k) )
 *   epi(v): v += bias[n];  if act: (aux ? aux(m,n) = v : 0), v = act(v);   v += residual(m,n)
 *   mul_dact: v = v * act'(aux(m,n))   (aux is an INPUT: the saved pre-activation; bias/residual unused)
 *   aux_kind 1: aux holds act'(pre-activation) instead of the pre-activation - the forward epilogue stores the
 *               derivative (its exponential is already computed there) and mul_dact multiplies by aux as it is
 *   aux_kind 2: as 1, but aux is ONE BYTE per element, q = rint((act'(v) + 0.125) * 204) (QuickGELU': [-0.10, 1.10],
 *               absolute error <= 0.0025; erf-GELU': [-0.129, 1.129], the two extremes saturate: <= 0.0065); ldaux in bytes; bf16 operands and output, full 256 x 256 tiles (M, N
 *               multiples of 256), 16-byte aligned operands, ldaux % 8 == 0, no split-K - otherwise
 *               SEGCLIP_ERR_UNSUPPORTED (callers fall back to aux_kind 1)
 * Replaces nn.Linear / MHA in-proj / out-proj / the einsum + Conv1d contractions:
 *   modules/module_seg_vit.py:166-172,189,266-269,304,309 ; modules/module_clip_ttransformer.py:24-30 ;
 *   modules/module_clip.py:91-94,131-134 ; modules/modeling.py:356-357 and their autograd backward
 *   (dgrad: A = dY, B(n,k) = W^T via strides; wgrad: A = dY^T, B = X^T via strides).
 * Operand strides are in elements.  f32 operands: any strides.  bf16 operands: each of A, B must
 * have unit stride along k (sak == 1) or along its row index (sam == 1).  a_dtype may be F32 with
 * b_dtype BF16 (fp32 residual-stream gradients are rounded to bf16 while staging).
 * Batch index z = z1*nb2 + z2 with independent strides for both levels (heads / groups).
 * ------------------------------------------------------------------------------------------ */
typedef struct segclip_gemm_desc {
  const void* A;
  const void* B;
  void* C;
  const float* bias;    /* [N] fp32 or NULL */
  const void* residual; /* (M,N) leading dim ldr, dtype r_dtype, or NULL */
  void* aux;            /* (M,N) leading dim ldaux, dtype c_dtype, or NULL */
  int64_t M, N, K;
  int64_t sam, sak;
  int64_t sbn, sbk;
  int64_t ldc, ldr, ldaux;
  int64_t nb1, nb2;
  int64_t bsA1, bsA2, bsB1, bsB2, bsC1, bsC2; /* aux uses the C batch strides */
  int64_t bsR1, bsR2;                         /* residual batch strides (0 = broadcast over the batch) */
  int32_t a_dtype, b_dtype, c_dtype, r_dtype;
  int32_t act;
  int32_t mul_dact;
  float alpha;
  int32_t aux_kind; /* 0: aux = pre-activation; 1: aux = act'(pre-activation); 2: the same as one byte per element */
  void* ws;         /* optional split-K scratch (bf16 path, plain epilogue only); NULL = no split-K */
  int64_t ws_bytes; /* size of ws; segclip_gemm_ws_bytes(d) is the amount that enables split-K */
  float* colsum;    /* optional [N] fp32: column sums of the stored C (bias gradient fused into the epilogue).
                       Needs colsum_ws = (M/64)*N floats, M % 256 == 0, N % 256 == 0 and the LDS-DMA bf16 path;
                       otherwise segclip_gemm returns SEGCLIP_ERR_UNSUPPORTED without launching. */
  float* colsum_ws;
  int32_t flags;    /* SEGCLIP_GEMM_DEFER_*: leave the trailing reduction launches to the caller (segclip_reduce_multi):
                       the split-K slabs stay in ws, the column-sum partials in colsum_ws */
  int32_t res_row_mod; /* > 0: the residual of output row m is residual row (m % res_row_mod) - a (T, N) table broadcast over
                          the B samples of an (B*T, N) output (the positional table added to the patch embedding,
                          modules/module_clip_vtransformer.py:56-64) without a batched launch.  Honoured for fp32 outputs
                          with an fp32 residual on full 256 x 256 bf16 tiles; otherwise SEGCLIP_ERR_UNSUPPORTED. */
} segclip_gemm_desc;

#define SEGCLIP_GEMM_DEFER_SPLITK 1 /* do not launch the split-K combine: C is NOT written; combine ws later */
#define SEGCLIP_GEMM_DEFER_COLSUM 2 /* do not launch the column-sum reduction of colsum_ws ((M/64) x N partials) */

size_t segclip_gemm_ws_bytes(const segclip_gemm_desc* d);
/* number of K splits segclip_gemm(d) uses with the ws / ws_bytes given in d: ws then holds that many (nb1*nb2*M*N) fp32
 * slabs before the combine (SEGCLIP_GEMM_DEFER_SPLITK leaves them there); 1 = no split-K */
int segclip_gemm_splits(const segclip_gemm_desc* d);
int segclip_gemm(const segclip_gemm_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Deferred reductions, many per launch.  The backward of a residual block ends in ~8 small reductions (4 split-K
 * combines of the weight gradients, the dgamma|dbeta|column-sum partials of 2 LayerNorm backwards, fused bias-gradient
 * column sums): their producers can leave the partials in their workspaces (SEGCLIP_GEMM_DEFER_*,
 * segclip_layernorm_bwd with dgamma == NULL) and the caller combines them with ONE launch per kind.
 *   kind SEGCLIP_REDUCE_SLABS: out[i] = scale * sum_s src[s*width + i], i < width (width % 4 == 0; fp32 or bf16 out)
 *   kind SEGCLIP_REDUCE_ROWS : out_k[c] = sum_r src[r*ld + k*seg + c] for k < nseg, c < seg (fp32 out, seg % 4 == 0):
 *                              column sums of a (rows x nseg*seg) partial matrix, split into up to 3 output arrays
 * Deterministic (fixed summation order), n <= SEGCLIP_REDUCE_MAX entries per call, all of one kind.
 * Replaces nothing in the reference (torch sums these inside its autograd kernels); it is the second stage of
 * modules/module_clip_util.py:126-132 (LayerNorm backward) and of every nn.Linear weight / bias gradient.
 * ------------------------------------------------------------------------------------------ */
#define SEGCLIP_REDUCE_SLABS 0
#define SEGCLIP_REDUCE_ROWS 1
#define SEGCLIP_REDUCE_MAX 16
typedef struct segclip_reduce_entry {
  const float* src;
  void* out0;
  float* out1;
  float* out2;
  int64_t rows;   /* SLABS: number of slabs; ROWS: number of partial rows */
  int64_t width;  /* SLABS: elements per slab; ROWS: nseg * seg */
  int64_t ld;     /* ROWS: leading dimension of src (floats) */
  int64_t seg;    /* ROWS: columns per output array */
  float scale;    /* SLABS only */
  int32_t out_dtype; /* SLABS: SEGCLIP_F32 / SEGCLIP_BF16 */
} segclip_reduce_entry;
int segclip_reduce_multi(const segclip_reduce_entry* entries, int n, int kind, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last axis (fp32 statistics).  modules/module_clip_util.py:126-132,
 * modules/module_seg_vit.py:150-156 (eps 1e-5), modules/modeling.py:152 (eps 1e-6).
 * bwd: dx = LN'(dy) (+ dres if given);  dgamma/dbeta (fp32, [cols]) are fully reduced.
 *      dx_bf16 (optional): a bf16 copy of dx for the GEMMs that consume it next;
 *      dres_colsum (optional, needs dres): column sums of dres = the bias gradient of the Linear whose
 *      output gradient dres is (fused here because this kernel streams dres anyway).
 * ws: segclip_layernorm_bwd_ws_bytes(rows, cols) bytes of scratch.
 * dgamma == NULL: the final reduction is left to the caller - ws then holds segclip_layernorm_bwd_ws_bytes / (3*cols*4)
 *      partial rows of [dgamma | dbeta | dres column sums] (3*cols floats each), see segclip_reduce_multi.
 * ------------------------------------------------------------------------------------------ */
int segclip_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                          float* rstd, int64_t rows, int64_t cols, float eps, int x_dtype, int y_dtype,
                          void* stream);
size_t segclip_layernorm_bwd_ws_bytes(int64_t rows, int64_t cols);
int segclip_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                          const float* rstd, const void* dres, void* dx, void* dx_bf16, float* dgamma,
                          float* dbeta, float* dres_colsum, void* ws, int64_t rows, int64_t cols, int dy_dtype,
                          int x_dtype, int dx_dtype, void* stream);
/* Row-mapped variants: row r of the kernel's row space is row (r / seg_in) * seg_out + seg_off + r % seg_in of y
 * (forward) / of dy (backward); x, mean, rstd, dres, dx keep plain rows.  LayerNorm(cat([centers, tokens], dim=1))
 * without the concatenated copy - modules/module_seg_vit.py:294-296 (`kv = torch.cat([q, inputs], dim=1)`) with
 * :211 (`self.ln_1(k)`): one call per part writes its token slice of every sample of the (B, seg_out, cols) buffer. */
int segclip_layernorm_fwd_seg(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                              float* rstd, int64_t rows, int64_t cols, float eps, int x_dtype, int y_dtype,
                              int64_t seg_in, int64_t seg_out, int64_t seg_off, void* stream);
int segclip_layernorm_bwd_seg(const void* dy, const void* x, const float* gamma, const float* mean,
                              const float* rstd, const void* dres, void* dx, void* dx_bf16, float* dgamma,
                              float* dbeta, float* dres_colsum, void* ws, int64_t rows, int64_t cols, int dy_dtype,
                              int x_dtype, int dx_dtype, int64_t seg_in, int64_t seg_out, int64_t seg_off,
                              void* stream);

/* Three affine outputs of ONE normalisation (n must be 3; fp32 x; cols 768 or 1024; y / dy all fp32 or all bf16 - otherwise
 * SEGCLIP_ERR_UNSUPPORTED and nothing is launched).  The learnable-center stage normalises the same token rows with
 * `self.norm` (modules/module_seg_vit.py:289) and with `ln_1` of both cross-attention layers (:211 over :294-296): one
 * read of x in the forward, and ONE dx = sum of the three LayerNorm backwards in the backward.
 * gamma / beta / y / dy: arrays of n device pointers; maps: n x {seg_in, seg_out, seg_off} (seg_in = 0: identity) - the row
 * mapping of output k / gradient k as in segclip_layernorm_fwd_seg.
 * dgb: (2n, cols) fp32 <- [dgamma_0, dbeta_0, dgamma_1, dbeta_1, ...];  ws: segclip_layernorm_bwd_multi_ws_bytes. */
int segclip_layernorm_fwd_multi(const void* x, int n, const float* const* gamma, const float* const* beta,
                                void* const* y, const int64_t* maps, float* mean, float* rstd, int64_t rows,
                                int64_t cols, float eps, int x_dtype, int y_dtype, void* stream);
size_t segclip_layernorm_bwd_multi_ws_bytes(int64_t rows, int64_t cols, int n);
int segclip_layernorm_bwd_multi(const void* const* dy, const void* x, int n, const float* const* gamma,
                                const int64_t* maps, const float* mean, const float* rstd, void* dx, float* dgb,
                                void* ws, int64_t rows, int64_t cols, int dy_dtype, int x_dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-head attention core  O = softmax(scale * Q K^T [+ causal mask]) V,  head_dim <= 64.
 * Replaces the inside of nn.MultiheadAttention: modules/module_seg_vit.py:189 (self, 12x64),
 * :215 (center cross-attention; K/V addressed with explicit (batch, token) strides so that both
 * the torch-1.8 "t18" buffer reinterpretation and the intended layout run on the same kernel,
 * SURVEY.md finding 0.4), modules/module_clip_ttransformer.py:46 (causal, mask of
 * modules/module_clip_util.py:199-205 generated in-kernel), modules/module_mae.py:124-130 (8x48).
 * Element strides: *_sb batch, *_st token; head h lives at offset h*hd.
 * stats: fp32 scratch kept for backward, segclip_attn_stats_bytes() bytes
 *        (bf16: log-sum-exp per row; f32: the full probability matrix).
 * bwd ws: segclip_attn_bwd_ws_bytes() bytes (f32: dP; bf16 with more than 256 tokens: the streaming kernels' D and
 *         column-sum vectors; 0 otherwise).
 * ------------------------------------------------------------------------------------------ */
#define SEGCLIP_ATTN_FP8 1
typedef struct segclip_attn_desc {
  const void* Q;
  const void* K;
  const void* V;
  void* O;
  void* stats;
  const void* dO;
  void* dQ;
  void* dK;
  void* dV;
  void* ws;
  int64_t B, H, Tq, Tk, hd;
  int64_t q_sb, q_st, k_sb, k_st, v_sb, v_st, o_sb, o_st;
  int64_t dq_sb, dq_st, dk_sb, dk_st, dv_sb, dv_st, do_sb, do_st;
  float scale;
  int32_t causal;
  int32_t dtype;
  /* SEGCLIP_ATTN_FP8 (bf16 dtype, forward only): Q K^T and P V on the e4m3 MFMA (per-token scales for Q and K, one
   * scale per 256-key chunk for V, P x 256); the statistics it leaves serve the bf16 backward.  BASELINE configs[4]. */
  int32_t flags;
  /* bwd, bf16 only, nullable: fp32 [B][3][H*hd] receives, per sample, the token sums of dQ | dK | dV
   * (= that sample's contribution to the in_proj bias gradient of nn.MultiheadAttention), computed from the
   * tiles already in LDS instead of a separate column-sum pass over dQ/dK/dV. */
  void* colsum_part;
  /* nullable: int32 [B] number of valid keys per sample (keys >= klen[b] are masked out).  The key-padding mask of the
   * text-MAE decoder blocks, modules/module_mae.py:213-219: (1 - attention_mask) * -1e6 added to the logits, where
   * attention_mask is a prefix mask (captions are padded at the end) and exp(-1e6) == 0 in fp32.  Sequences of at most
   * 256 keys. */
  const int32_t* klen;
} segclip_attn_desc;

size_t segclip_attn_stats_bytes(const segclip_attn_desc* d);
size_t segclip_attn_bwd_ws_bytes(const segclip_attn_desc* d);
int segclip_attn_fwd(const segclip_attn_desc* d, void* stream);
int segclip_attn_bwd(const segclip_attn_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Launch executor: the forward of one pre-LN residual attention block (modules/module_seg_vit.py:175-196,
 * modules/module_clip_ttransformer.py:20-37; bf16 mode) as ONE call - LayerNorm, in_proj, attention, out_proj + residual,
 * LayerNorm, c_fc + activation + saved derivative, c_proj + residual: the seven launches the Python layer would issue, with the
 * same descriptors, into caller-owned buffers (bit-identical results; what it saves is ~250 us of host time per block).
 * x, x1, xo: the residual stream (M, D) in x_dtype (fp32, or bf16 with config.bf16_resid); everything else bf16 unless noted.
 * M may exceed B*T (row-padded stack): the attention touches the B*T token rows, the pad rows of o are zeroed.
 * Weights are the (out, in) bf16 copies of the reference's parameters; biases and LayerNorm affines fp32.
 * ------------------------------------------------------------------------------------------ */
typedef struct segclip_resblock_fwd_desc {
  const void* x;                 /* (M, D) x_dtype */
  const float* ln1w; const float* ln1b; const void* wqkv; const float* bqkv; const void* wo; const float* bo;
  const float* ln2w; const float* ln2b; const void* wfc; const float* bfc; const void* wpr; const float* bpr;
  void* y1; float* mean1; float* rstd1;   /* ln_1 output (M, D) and its row statistics */
  void* qkv;                     /* (M, 3D) */
  void* o;                       /* (M, D) attention output */
  void* stats;                   /* segclip_attn_stats_bytes: the attention's softmax statistics */
  void* x1;                      /* (M, D) x_dtype: the stream after the attention branch */
  void* y2; float* mean2; float* rstd2;
  void* h; int64_t ld_h;         /* (M, F) activation output, row pitch ld_h */
  void* u; int64_t ld_u;         /* nullable: what the backward keeps of the pre-activation (aux_kind as in segclip_gemm_desc) */
  void* xo;                      /* (M, D) x_dtype: the block's output */
  const void* klen;              /* nullable: int32 [B] valid keys per sample */
  int64_t M, B, T, D, F, H;
  float eps, attn_scale;
  int32_t causal, act, aux_kind, x_dtype;
} segclip_resblock_fwd_desc;
int segclip_resblock_fwd(const segclip_resblock_fwd_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise / reduction helpers
 * ------------------------------------------------------------------------------------------ */
/* dst[i] = (dst_dtype) src[i] */
int segclip_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream);
/* fp32 operand -> its two bf16 parts, three blocks along the contraction dimension (role 0: hi|lo|hi, role 1: hi|hi|lo; stack 0:
 * blocks side by side in a row of 3*cols, stack 1: three (rows x cols) blocks one after the other).  With both operands split this
 * way ONE bf16 GEMM of contraction length 3K computes A_hi B_hi + A_lo B_hi + A_hi B_lo in fp32 accumulators: the fp32 parity mode
 * of the Linear layers (reference arithmetic: main_task_align.py:102, fp32 torch.nn.Linear) on the bf16 matrix pipe. */
int segclip_split3_bf16(const float* src, void* dst, int64_t rows, int64_t cols, int64_t ld, int stack, int role, void* stream);
/* out[n] = sum_m X[m*ld + n]  (bias gradients; deterministic two-stage).  ws: colsum_ws_bytes */
size_t segclip_colsum_ws_bytes(int64_t M, int64_t N);
int segclip_colsum(const void* X, float* out, void* ws, int64_t M, int64_t N, int64_t ld, int dtype,
                   void* stream);
/* y = act(x) ; dx = dy * act'(x)  (stand-alone activation, modules/module_seg_vit.py:274,330) */
int segclip_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, void* stream);
int segclip_act_bwd(const void* dy, const void* x, void* dx, int64_t n, int act, int dtype, void* stream);
/* out = a + b (same dtype) */
int segclip_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Vision front end.  modules/module_clip_vtransformer.py:56-64.
 * im2col: image (B,3,H,W) fp32 -> cols (B*gh*gw, 3*p*p), column order (c,py,px) [layout 0, conv1]
 *         or (py,px,c) [layout 1, MAE target `patchify`, modules/module_mae.py:18-29].
 * assemble: x[b,0,:] = cls + pos[0];  x[b,1+t,:] = patches[b,t,:] + pos[1+t]   (fp32 out)
 * ------------------------------------------------------------------------------------------ */
int segclip_im2col(const float* image, void* cols, int64_t B, int64_t C, int64_t H, int64_t W, int64_t p,
                   int layout, int out_dtype, void* stream);
/* same, with rows of `ld` >= 3*p*p elements whose tail columns are zero-filled: the contraction dimension of the
 * patch-embedding GEMM padded to the kernel's alignment (ViT-L/14: 3*14*14 = 588 -> 640). */
int segclip_im2col_ld(const float* image, void* cols, int64_t B, int64_t C, int64_t H, int64_t W, int64_t p,
                      int layout, int out_dtype, int64_t ld, void* stream);
int segclip_vis_assemble(const void* patches, const float* cls, const float* pos, float* x, int64_t B,
                         int64_t T, int64_t D, int p_dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Text front end.  modules/module_clip.py:109-112 (nn.Embedding gather + positional add) and its
 * backward (scatter-add into the fp32 table gradient, which must be zero-initialised by the caller;
 * dpos (L,D) is fully written).
 * ------------------------------------------------------------------------------------------ */
int segclip_embed_fwd(const int64_t* ids, const float* table, const float* pos, float* out, int64_t B,
                      int64_t L, int64_t D, int64_t vocab, void* stream);
int segclip_embed_bwd(const int64_t* ids, const float* dout, float* dtable, float* dpos, int64_t B,
                      int64_t L, int64_t D, int64_t vocab, void* stream);

/* MAE glue of the full loss, fp32, D a multiple of 4 (reference modules/modeling.py:240-242: mean over the tokens prepended as
 * the CLS row; modules/module_mae.py:310-314: mask tokens appended, un-shuffled by ids_restore, + positional table).
 *   mean_cat:      out (B,T+1,D): out[b][0] = mean_t x[b][t], out[b][1+t] = x[b][t];  bwd: dx[b][t] = dout[b][1+t] + dout[b][0] / T
 *   mae_unshuffle: out (B,L,D)[b][j] = (ids[b][j] < K ? x[b][ids[b][j]] : mask_token) + pos[j]   (x (B,K,D), ids a permutation of 0..L-1
 *                  per sample); bwd: dx (B,K,D), dpos (L,D) = sum_b dout, mpart (L,D) = the mask-token share of that sum per
 *                  position - the mask token's gradient is the column sum of mpart (segclip_colsum). */
int segclip_mean_cat_fwd(const float* x, float* out, int64_t B, int64_t T, int64_t D, void* stream);
int segclip_mean_cat_bwd(const float* dout, float* dx, int64_t B, int64_t T, int64_t D, void* stream);
int segclip_mae_unshuffle_fwd(const float* x, const float* mask_token, const int64_t* ids, const float* pos, float* out, int64_t B,
                              int64_t K, int64_t L, int64_t D, void* stream);
int segclip_mae_unshuffle_bwd(const float* dout, const int64_t* ids, float* dx, float* dpos, float* mpart, int64_t B, int64_t K,
                              int64_t L, int64_t D, void* stream);

/* Row gather / scatter-add with int64 indices (EOT pick modules/module_clip.py:136, MAE keep /
 * un-shuffle gathers modules/module_clip_util.py:110, modules/module_mae.py:310).
 * gather: out[b, j, :] = src[b, idx[b,j], :] ; scatter_add: dsrc[b, idx[b,j], :] += dout[b, j, :]
 * (dsrc zero-initialised by the caller; indices within one b must be unique -> no atomics). */
int segclip_gather_rows(const void* src, const int64_t* idx, void* out, int64_t B, int64_t Tsrc,
                        int64_t Tout, int64_t D, int dtype, void* stream);
int segclip_scatter_rows(const void* dout, const int64_t* idx, void* dsrc, int64_t B, int64_t Tsrc,
                         int64_t Tout, int64_t D, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Learnable-center hard assignment.  modules/module_seg_vit.py:304-310 + gumbel_softmax :221-242.
 *   logits (B,G,T) fp32 = q k^T (un-scaled; computed by segclip_gemm in fp32 always)
 *   training: y = softmax((logits + gumbel)/tau, dim=G);  eval (gumbel NULL): y = softmax(logits)
 *   idx[b,t] = argmax_g y (first max wins, like torch.max);  hard = onehot(idx)
 *   soft = softmax(logits, dim=G)   (consumed by the segmentation evaluation only)
 *   counts[b,g] = max(sum_t hard, 1)
 * bwd (straight-through): dlogits = dy_soft-path gradient of y given dhard (B,G,T):
 *   dlogits[b,:,t] = (y * (dhard - sum_g dhard*y)) / tau
 * ------------------------------------------------------------------------------------------ */
int segclip_assign_fwd(const float* logits, const float* gumbel, float tau, float* y_soft, float* soft,
                       uint8_t* idx, float* hard, float* counts, int64_t B, int64_t G, int64_t T,
                       void* stream);
int segclip_assign_bwd(const float* dhard, const float* y_soft, float tau, float* dlogits, int64_t B,
                       int64_t G, int64_t T, void* stream);

/* ------------------------------------------------------------------------------------------
 * Losses.
 * l2norm: y = x / ||x||  rows (modules/modeling.py:341-345).
 * ce: mean over rows of -log softmax(logits)[label], label = row + label_offset
 *     (modules/modeling.py:205-209).  fwd writes loss (1 float) and lse per row; bwd writes
 *     dlogits = gscale * (softmax - onehot) / rows.
 * superpixel_kl: modules/modeling.py:212-224 on the hard assignment indices (label-histogram
 *     formulation).  Output is a constant w.r.t. parameters only through hard's straight-through
 *     gradient: bwd returns dhard (B,G,T).
 * masked_mse: MAE loss modules/module_mae.py:323-328: pred (B,1+T,Dp) (row 0 = CLS, skipped),
 *     target (B,T,Dp), mask (B,1+T).
 * ------------------------------------------------------------------------------------------ */
int segclip_l2norm_fwd(const float* x, float* y, float* norm, int64_t rows, int64_t cols, void* stream);
int segclip_l2norm_bwd(const float* dy, const float* y, const float* norm, float* dx, int64_t rows,
                       int64_t cols, void* stream);
int segclip_ce_fwd(const float* logits, float* lse, float* loss_rows, int64_t rows, int64_t cols,
                   int64_t label_offset, void* stream);
/* dlogits = (*gscale_ptr) * gscale * (softmax - onehot) / rows ; gscale_ptr (device scalar, upstream
 * gradient) may be NULL */
int segclip_ce_bwd(const float* logits, const float* lse, const float* gscale_ptr, float gscale,
                   float* dlogits, int64_t rows, int64_t cols, int64_t label_offset, void* stream);
/* nn.CrossEntropyLoss(ignore_index) with explicit labels (text-MAE vocabulary loss, modules/module_mae.py:353):
 * fwd: lse[r], loss_rows[r] = (labels[r] == ignore ? 0 : lse[r] - logits[r][labels[r]]), valid[r] = labels[r] != ignore
 *      (the mean over valid rows is taken by the caller: sum(loss_rows) / sum(valid));
 * bwd: dlogits[r][c] = (labels[r] == ignore ? 0 : softmax[r][c] - [c == labels[r]]) * gscale[0] * inv_count[0]. */
int segclip_ce_labels_fwd(const float* logits, const int64_t* labels, int64_t ignore_index, float* lse, float* loss_rows,
                          float* valid, int64_t rows, int64_t cols, void* stream);
int segclip_ce_labels_bwd(const float* logits, const float* lse, const int64_t* labels, int64_t ignore_index,
                          const float* gscale, const float* inv_count, float* dlogits, int64_t rows, int64_t cols,
                          void* stream);
/* loss_rows[b] = this image's share of the loss (already / (2*B*T*G)); dhard = d(sum loss_rows)/dhard */
int segclip_superpixel_kl(const float* hard, const int64_t* seg, float* loss_rows, float* dhard,
                          int64_t B, int64_t G, int64_t T, void* stream);
/* loss_rows[b*T+t] = mask[b,1+t] * mean_d (pred[b,1+t,d]-target[b,t,d])^2 ; loss = sum(loss_rows)/sum(mask[:,1:]) */
int segclip_masked_mse_fwd(const void* pred, const float* target, const float* mask, float* loss_rows,
                           int64_t B, int64_t T, int64_t Dp, int pred_dtype, void* stream);
int segclip_masked_mse_bwd(const void* pred, const float* target, const float* mask,
                           const float* gscale_ptr, const float* mask_sum, float gscale, void* dpred,
                           int64_t B, int64_t T, int64_t Dp, int pred_dtype, void* stream);
/* out[i] = x[i] * (*s)  (device scalar) ;  out[0] = scale * sum(x)  (deterministic single block) */
int segclip_scale(const float* x, const float* s, float* out, int64_t n, void* stream);
/* out = -log(-log(clamp(u, FLT_MIN, 1 - FLT_EPSILON))): Gumbel(0,1) noise of the hard assignment from uniform samples
 * (reference modules/module_seg_vit.py:223-226, torch.distributions.Gumbel(0, 1).sample); u and out may alias */
int segclip_gumbel_from_uniform(const float* u, float* out, int64_t n, void* stream);
int segclip_reduce_sum(const float* x, float* out, int64_t n, float scale, void* stream);

/* Evaluation tier: positional table at another grid, modules/module_clip_vtransformer.py:35-53 =
 * F.interpolate(table (n x n x D, channel-last), size=(h, w), mode='bicubic', align_corners=False). */
int segclip_interp_bicubic(const float* src, float* dst, int64_t n_in, int64_t h, int64_t w, int64_t D, void* stream);

/* ------------------------------------------------------------------------------------------
 * MAE random masking (integer path, bit-exact given the noise).  modules/module_clip_util.py:91-124
 * with keep_cls: noise[:,0] = -1; ids_shuffle = argsort(noise) (stable); ids_restore =
 * argsort(ids_shuffle); mask = 1 except the first len_keep of the shuffle.
 * ------------------------------------------------------------------------------------------ */
int segclip_mask_sort(const float* noise, int64_t* ids_shuffle, int64_t* ids_restore, float* mask,
                      int64_t B, int64_t L, int64_t len_keep, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training-step tail with no host synchronisation:
 *   torch.nn.utils.clip_grad_norm_(model.parameters(), clip_grad)      main_task_align.py:326
 *   AdaptAdamW.step()                                                   modules/optimization_adamw.py:111-174
 *   "skip the loss with NAN manually"                                   main_task_align.py:331-339
 *   torch.clamp_(logit_scale, max=ln 100), total_loss += float(loss)    main_task_align.py:323, 343-347
 * All tensors fp32 (the master copy); `shadow_bf16`, when non-NULL, receives the bf16 rounding of
 * the updated parameter (the compute-dtype copy the next forward's GEMMs read).
 *
 * One segclip_train_ctrl lives in device memory, zeroed once by the caller.  A NaN loss makes
 * segclip_adamw_step leave param/exp_avg/exp_avg_sq untouched and segclip_train_step_finish count
 * the skip; the bias corrections and the schedule use  step = tensor.step - ctrl->nan_skips, i.e.
 * exactly the reference's state['step'] (which is not advanced on skipped iterations).
 * ------------------------------------------------------------------------------------------ */
enum { SEGCLIP_SCHED_WARMUP_COSINE = 0, SEGCLIP_SCHED_WARMUP_CONSTANT = 1, SEGCLIP_SCHED_WARMUP_LINEAR = 2 };

typedef struct segclip_train_ctrl {
  float grad_sqnorm; /* sum of squares of all gradients (segclip_grad_sqnorm) */
  float clip_coef;   /* min(1, max_norm / (sqrt(grad_sqnorm) + 1e-6)) */
  int32_t nan_skips; /* iterations skipped so far because the loss was NaN */
  int32_t steps;     /* iterations finished (skipped ones included) */
  float loss_sum;    /* running sum of the non-NaN losses (train_epoch's total_loss) */
  float last_loss;
  int32_t reserved[2];
} segclip_train_ctrl;

typedef struct segclip_adamw_group { /* one torch param_group: optimization_adamw.py:70-73 */
  double lr, weight_decay, b1, b2, eps, warmup, lr_start, lr_end;
  int64_t t_total;  /* -1: constant lr */
  int32_t schedule; /* SEGCLIP_SCHED_* */
  int32_t reserved;
} segclip_adamw_group;

typedef struct segclip_adamw_tensor {
  float* param;
  float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  void* shadow_bf16; /* nullable */
  int64_t n;
  int32_t step;  /* host-side count of steps (this one included) in which this tensor had a grad */
  int32_t group; /* index into groups[] */
} segclip_adamw_tensor;

/* Grouped 64-channel linear layers on channel-last rows = nn.Conv1d(D, D, kernel_size 1, groups = D/64, bias=False), the
 * k_conv / v_conv of the learnable-center stage (reference modules/module_seg_vit.py:266,269 applied at :299,302):
 *   out_o(m, g*64 + n) = sum_i sum_k in_i(m, g*64 + k) * w[i*n_out + o][g*64 + n][k],   bf16 in / out, fp32 accumulation.
 * (n_in, n_out) = (1, 1), (1, 2): one or two convolutions of the same input in one pass; (2, 1): sum of two - their data
 * gradient when w holds the per-group TRANSPOSED weights.  in/out/w and the pitches are HOST arrays; w[.]: (groups*64, 64)
 * bf16 row-major; all pointers 16-byte aligned, pitches multiples of 8 elements. */
int segclip_group_linear64(const void* const* in, const int64_t* ld_in, int n_in, void* const* out, const int64_t* ld_out,
                           int n_out, const void* const* w, int64_t M, int groups, void* stream);

/* ---- pooled-feature head and contrastive loss (csrc/head.hip) ------------------------------------------------------
 * out[b][d] = max_t x[b][t][d], idx = first arg-max token (torch.max(x, dim=1), reference modules/module_seg_vit.py:441);
 * backward: dx[b][t][d] = dout[b][d] if t == idx[b][d] else 0, written as fp32 (dx, nullable) and / or bf16 (dx_bf16). */
int segclip_max_tokens_fwd(const float* x, float* out, int32_t* idx, int64_t B, int64_t T, int64_t D, void* stream);
int segclip_max_tokens_bwd(const float* dout, const int32_t* idx, float* dx, void* dx_bf16, int64_t B, int64_t T, int64_t D,
                           void* stream);
/* both[b][0] = v[b] / |v[b]|, both[b][1] = t[b] / |t[b]| (B, 2, C) - the stacked message of the embedding all-gather
 * (reference modules/modeling.py:341-345,352-354); norms (2B).  Backward: dv, dt from dboth (+ dboth2, nullable). */
int segclip_l2norm_pair_fwd(const float* v, const float* t, float* both, float* norms, int64_t B, int64_t C, void* stream);
int segclip_l2norm_pair_bwd(const float* dboth, const float* dboth2, const float* both, const float* norms, float* dv,
                            float* dt, int64_t B, int64_t C, void* stream);
/* cos (2, B, N) raw cosines ([0] = t v_all^T, [1] = v t_all^T); logits = min(exp(*logit_scale), 100) * cos; row r of either
 * matrix has label r + label_offset; *loss = mean of the 2B row losses = (CE(t2v) + CE(v2t)) / 2 (reference
 * modules/modeling.py:204-209,346-362).  Backward: dcos = (*g / 2B) * scale * (softmax - onehot); *dlogit_scale (nullable)
 * = d loss / d logit_scale through the clamp.  lse, loss_rows, ds_rows: (2B) workspaces / saved statistics. */
int segclip_clip_ce_fwd(const float* cos, const float* logit_scale, float* lse, float* loss_rows, float* loss, int64_t B,
                        int64_t N, int64_t label_offset, void* stream);
int segclip_clip_ce_bwd(const float* cos, const float* lse, const float* logit_scale, const float* g, float* dcos,
                        float* ds_rows, float* dlogit_scale, int64_t B, int64_t N, int64_t label_offset, void* stream);

/* Segment mean of the learnable-center stage (reference modules/module_seg_vit.py:308-309):
 *   out[b][g][:] = (sum_t [idx[b][t] == g] v[b][t][:]) / max(counts[b][g], 1)       (hard assignment = one-hot of idx)
 * and its backward: dv[b][t][:] = dN[b][idx[b][t]][:], dhard[b][g][t] = dN[b][g] . v[b][t] + dc[b][g] with dN = dout / max(count, 1),
 * dc = -[count >= 1] (dout[b][g] . out[b][g]) / max(count, 1).  idx (B,T) uint8, counts (B,G) fp32 as segclip_assign_fwd leaves
 * them; v / dv (B,T,D) fp32 or bf16; out, dout (B,G,D) and dhard (B,G,T) fp32.  G <= 8, D a multiple of 4 (backward: D <= 1024). */
int segclip_segmean_fwd(const uint8_t* idx, const void* v, int v_dtype, const float* counts, float* out, int64_t B, int64_t G,
                        int64_t T, int64_t D, void* stream);
int segclip_segmean_bwd(const float* dout, const float* out, const uint8_t* idx, const void* v, int v_dtype, const float* counts,
                        void* dv, float* dhard, int64_t B, int64_t G, int64_t T, int64_t D, void* stream);

/* Token rows from the centers (MAE branch, reference modules/module_seg_vit.py:338-342): out (B,M,D) = a (B,M,G) @ x (B,G,D), fp32,
 * and the backward da (B,M,G) = dout x^T, dx (B,G,D) = a^T dout.  Covers G = 8, D a multiple of 4 up to 4096; other shapes return
 * SEGCLIP_ERR_UNSUPPORTED (use segclip_gemm). */
int segclip_recon_mix_fwd(const float* a, const float* x, float* out, int64_t B, int64_t M, int64_t G, int64_t D, void* stream);
int segclip_recon_mix_bwd(const float* a, const float* x, const float* dout, float* da, float* dx, int64_t B, int64_t M, int64_t G,
                          int64_t D, void* stream);

/* Assignment logits of the center stage: attn[b][g][t] = q[b][g][:] . k[b][t][:] (fp32, un-scaled; reference
 * modules/module_seg_vit.py:304) and the backward dq = dl k, dk = dl^T q.  Covers G = 8, D = 768 | 1024; other shapes return
 * SEGCLIP_ERR_UNSUPPORTED (use segclip_gemm).  Not the summation order of the exact-fp32 GEMM. */
int segclip_center_logits_fwd(const float* q, const float* k, float* attn, int64_t B, int64_t G, int64_t T, int64_t D, void* stream);
int segclip_center_logits_bwd(const float* dl, const float* q, const float* k, float* dq, float* dk, int64_t B, int64_t G,
                              int64_t T, int64_t D, void* stream);

/* dst[i][:] = bf16(src[i][:]) for `count` fp32 tensors in ceil(count/32) launches (the compute-dtype copies of the
 * GEMM weights, refreshed once per forward instead of one cast launch per weight).  src/dst/n are HOST arrays. */
int segclip_multi_cast_bf16(const float* const* src, void* const* dst, const int64_t* n, int64_t count, void* stream);

/* dst[i][:] += src[i][:] for `count` fp32 tensor pairs in ceil(count/32) launches: a parameter used by two passes of one
 * step (the vision tower runs on the clean and on the masked image when the MAE loss is on, reference modules/modeling.py:
 * 196,237-249) receives two gradients; autograd would add them with one launch per parameter.  dst/src/n are HOST arrays. */
int segclip_multi_add_f32(float* const* dst, const float* const* src, const int64_t* n, int64_t count, void* stream);

/* number of fp32 partial sums segclip_grad_sqnorm needs in `ws` for these tensor sizes */
size_t segclip_grad_sqnorm_ws_bytes(const int64_t* n, int64_t count);
/* ctrl->grad_sqnorm = sum_i |grads[i]|^2 (deterministic two-level reduction), ctrl->clip_coef as above
 * (max_norm <= 0: coef = 1).  grads/n are HOST arrays of device pointers / element counts. */
int segclip_grad_sqnorm(const float* const* grads, const int64_t* n, int64_t count, float* ws,
                        segclip_train_ctrl* ctrl, float max_norm, void* stream);
/* fused multi-tensor AdaptAdamW step.  tensors/groups are HOST arrays (passed to the kernels by
 * value, 32 tensors per launch).  g = grad * ctrl->clip_coef; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g g;
 * denom = sqrt(v)/sqrt(1-b2^step) + eps; p = p (1 - lr_t wd) - (lr_t/(1-b1^step)) m/denom, where
 * lr_t = lr * schedule(step/t_total).  ctrl may be NULL (coef 1, no NaN logic, step = tensor.step);
 * loss may be NULL.  zero_grads != 0 also clears every grad (even on a skipped iteration). */
int segclip_adamw_step(const segclip_adamw_tensor* tensors, int64_t count, const segclip_adamw_group* groups,
                       int64_t ngroups, const segclip_train_ctrl* ctrl, const float* loss, int zero_grads,
                       void* stream);
/* bookkeeping after the step: ctrl->nan_skips += isnan(*loss); ctrl->steps++; loss_sum/last_loss;
 * *logit_scale = min(*logit_scale, clamp_max) when logit_scale != NULL. */
int segclip_train_step_finish(segclip_train_ctrl* ctrl, const float* loss, float* logit_scale, float clamp_max,
                              void* stream);

/* ------------------------------------------------------------------------------------------
 * Grouped weight gradients (bf16 operands, fp32 results):  dw_i (M_i, N_i) = dy_i^T x_i  for n problems over the SAME R token
 * rows, as ONE launch.  The four nn.Linear weight gradients of a residual block (modules/module_seg_vit.py:162-196: in_proj,
 * out_proj, c_fc, c_proj; the text blocks of modules/module_clip_ttransformer.py:20-37) have 9-36 output tiles each: launched
 * one by one, each needs 7-28 K ranges to fill the 256 CUs and leaves 64 MB of fp32 partial tiles to be combined; the gradients
 * of several consecutive blocks together fill the chip with 1-4 K ranges.
 *   segclip_wgrad_group_splits(tiles, ksteps): the K-range count the library recommends for a group with `tiles` 256x256
 *       output tiles in total over ksteps = R / 64 steps.
 *   splits == 1: results go to dw_i (row pitch ld_dw).  splits > 1: K range s of problem i is left, raw, at
 *       ws + (sum_{j<i} splits*M_j*N_j + s*M_i*N_i) floats; the caller sums the ranges in order (segclip_reduce_multi, kind 0).
 *   Not covered (SEGCLIP_ERR_UNSUPPORTED, nothing launched): M_i or N_i not multiples of 256, R not a multiple of 64, operands
 *       off 16-byte boundaries, leading dimensions not multiples of 8, n > 48.
 * ------------------------------------------------------------------------------------------ */
typedef struct segclip_wgrad_item {
  const void* dy;   /* (R, M) bf16, row pitch ld_dy */
  const void* x;    /* (R, N) bf16, row pitch ld_x  */
  float* dw;        /* (M, N) fp32, row pitch ld_dw (written when splits == 1) */
  int64_t M, N, ld_dy, ld_x, ld_dw;
} segclip_wgrad_item;
int segclip_wgrad_group_splits(int64_t tiles, int64_t ksteps);
/* the time model behind that choice (microseconds on MI355X, fitted to profiles/r04_wgrad_group.txt: rounds of 256 workgroups x (K loop + output) + the partial tiles
 * written and read back); callers use it to decide how many blocks to group */
double segclip_wgrad_group_model_us(int64_t tiles, int64_t ksteps, int splits);
size_t segclip_wgrad_group_ws_bytes(const segclip_wgrad_item* items, int n, int splits);
int segclip_wgrad_group(const segclip_wgrad_item* items, int n, int64_t R, int splits, void* ws, size_t ws_bytes,
                        void* stream);

/* ------------------------------------------------------------------------------------------
 * Half-tile tail of the 256 x 256-tile bf16 GEMM (segclip_gemm with bf16 operands, M and N multiples of 256; the nn.Linear
 * products of modules/module_seg_vit.py:162-196).  A launch over `ntiles` output tiles whose last round of 256 workgroups is
 * at most half full (ntiles > 256, 0 < ntiles % 256 <= 128: out_proj / c_proj and the data gradients of in_proj / c_fc at
 * 256 x 197 token rows are 591 tiles) runs those tail tiles as twice as many 128 x 256 workgroups.  Results are bit-identical
 * to the full tiles'.  Returns the number of tail tiles run that way (0 = none; SEGCLIP_PQ_HALF=0 disables it).
 * The same half-tile workgroups cover a last row of 128 token rows: M = 256 q + 128 (128 samples x 197 / 577 / 77 tokens)
 * runs on this kernel as q rows of full tiles + N / 256 half-tiles.
 * ------------------------------------------------------------------------------------------ */
int segclip_gemm_pq_half_tail(int64_t ntiles);

#ifdef __cplusplus
}
#endif
#endif /* SEGCLIP_HIP_H */
