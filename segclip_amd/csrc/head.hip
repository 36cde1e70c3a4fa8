// The pooled-feature head and the contrastive (InfoNCE) loss of the training step, fused into a handful of launches
// (reference modules/module_seg_vit.py:441-442 [max over the patch tokens], modules/modeling.py:338-362
// [L2-normalise, clamp(exp(logit_scale), 100), the two logits matrices] and :204-209 [two cross entropies, averaged]).
// Everything here is latency-bound: the point is the NUMBER of launches between the last forward GEMM and the first
// backward GEMM (round 3: 87 launches, 0.9 ms of kernels in a 1.2-2.4 ms serial section).
#include "common.h"

namespace {

// ---------------------------------------------------------------- max over tokens (the CLS feature of SegViT)
// x (B, T, D) fp32 -> out (B, D), idx (B, D) = first token attaining the maximum.  One thread per (b, 4 columns).
__global__ void max_tokens_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int32_t* __restrict__ idx,
                                      int64_t B, int T, int D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = D / 4;
  if (i >= B * q) return;
  const int64_t b = i / q;
  const int c = (int)(i % q) * 4;
  const float* p = x + b * T * D + c;
  f32x4 best = *reinterpret_cast<const f32x4*>(p);
  int bi[4] = {0, 0, 0, 0};
  int t = 1;
  for (; t + 4 <= T; t += 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (int64_t)(t + u) * D);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (v[u][j] > best[j]) { best[j] = v[u][j]; bi[j] = t + u; }
  }
  for (; t < T; ++t) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + (int64_t)t * D);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (v[j] > best[j]) { best[j] = v[j]; bi[j] = t; }
  }
  *reinterpret_cast<f32x4*>(out + b * D + c) = best;
#pragma unroll
  for (int j = 0; j < 4; ++j) idx[b * D + c + j] = bi[j];
}
// dx (B, T, D) = dout routed to the arg-max token, zero elsewhere; optionally also as bf16 (the operand form the residual
// stack's backward wants): one pass instead of zero-fill + scatter + cast
__global__ void max_tokens_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ idx, float* __restrict__ dx,
                                      bf16_t* __restrict__ dx16, int64_t B, int T, int D) {
  const int q = D / 4;
  const int64_t total = B * T * q;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % q) * 4;
    const int64_t bt = i / q, b = bt / T;
    const int t = (int)(bt % T);
    const f32x4 g = *reinterpret_cast<const f32x4*>(dout + b * D + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = idx[b * D + c + j] == t ? g[j] : 0.f;
    if (dx) *reinterpret_cast<f32x4*>(dx + bt * D + c) = o;
    if (dx16) {
      u32x2 w;
      w[0] = pack2bf(o[0], o[1]); w[1] = pack2bf(o[2], o[3]);
      *reinterpret_cast<u32x2*>(dx16 + bt * D + c) = w;
    }
  }
}

// ---------------------------------------------------------------- contrastive head
__device__ __forceinline__ float clip_scale(const float* ls) { return fminf(expf(*ls), 100.f); }

// both (B, 2, C): [b][0] = v[b] / |v[b]|, [b][1] = t[b] / |t[b]| (the stacked message of the embedding all-gather);
// norms (2B): [2b + which].  One wave per (b, which) row.
__global__ void l2norm_pair_fwd_kernel(const float* __restrict__ v, const float* __restrict__ t, float* __restrict__ both,
                                       float* __restrict__ norms, int64_t B, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= 2 * B) return;
  const float* x = ((r & 1) ? t : v) + (r >> 1) * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) { const float a = x[c]; s += a * a; }
  const float n = sqrtf(wave_sum(s));
  if (lane == 0) norms[r] = n;
  for (int c = lane; c < C; c += 64) both[r * C + c] = x[c] / n;
}
// dv[b] = (g - y (g . y)) / |v[b]| with g = dboth[b][0] (+ dboth2[b][0]); same for t.  One wave per row.
__global__ void l2norm_pair_bwd_kernel(const float* __restrict__ dboth, const float* __restrict__ dboth2,
                                       const float* __restrict__ both, const float* __restrict__ norms, float* __restrict__ dv,
                                       float* __restrict__ dt, int64_t B, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= 2 * B) return;
  const float* g = dboth + r * C;
  const float* g2 = dboth2 ? dboth2 + r * C : nullptr;
  const float* y = both + r * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += (g[c] + (g2 ? g2[c] : 0.f)) * y[c];
  s = wave_sum(s);
  const float inv = 1.f / norms[r];
  float* d = ((r & 1) ? dt : dv) + (r >> 1) * C;
  for (int c = lane; c < C; c += 64) d[c] = ((g[c] + (g2 ? g2[c] : 0.f)) - y[c] * s) * inv;
}
// cos (2, B, N): [0] = t . v_all^T, [1] = v . t_all^T (raw cosines); logits = clamp(exp(logit_scale), 100) * cos.
// Row r of matrix z has label r + label_offset.  lse, loss_rows: (2B).  One wave per row.
__global__ void clip_ce_fwd_kernel(const float* __restrict__ cosm, const float* __restrict__ logit_scale, float* __restrict__ lse,
                                   float* __restrict__ loss_rows, int64_t B, int64_t N, int64_t label_offset) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= 2 * B) return;
  const float sc = clip_scale(logit_scale);
  const float* p = cosm + r * N;
  float mx = -INFINITY;
  for (int64_t c = lane; c < N; c += 64) mx = fmaxf(mx, sc * p[c]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int64_t c = lane; c < N; c += 64) s += expf(sc * p[c] - mx);
  s = wave_sum(s);
  const float l = mx + logf(s);
  if (lane == 0) { lse[r] = l; loss_rows[r] = l - sc * p[(r % B) + label_offset]; }
}
// dcos = coef * sc * (softmax - onehot), coef = g / (2B); ds_rows[r] = coef * sum_j (softmax - onehot)_j cos_j
__global__ void clip_ce_bwd_kernel(const float* __restrict__ cosm, const float* __restrict__ lse, const float* __restrict__ logit_scale,
                                   const float* __restrict__ g, float* __restrict__ dcos, float* __restrict__ ds_rows, int64_t B,
                                   int64_t N, int64_t label_offset) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= 2 * B) return;
  const float sc = clip_scale(logit_scale);
  const float coef = (g ? *g : 1.f) / (float)(2 * B);
  const float* p = cosm + r * N;
  const float l = lse[r];
  const int64_t label = (r % B) + label_offset;
  float ds = 0.f;
  for (int64_t c = lane; c < N; c += 64) {
    const float d = coef * (expf(sc * p[c] - l) - (c == label ? 1.f : 0.f));
    dcos[r * N + c] = d * sc;
    ds += d * p[c];
  }
  ds = wave_sum(ds);
  if (lane == 0) ds_rows[r] = ds;
}
// out[0] = sum(loss_rows) / n  (forward; single block) ;  dls = sum(ds_rows) * d clamp(exp(ls), 100) / d ls
__global__ void clip_reduce_kernel(const float* __restrict__ x, int64_t n, const float* __restrict__ logit_scale, float* __restrict__ out,
                                   int mode) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
  s = wave_sum(s);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t += red[w];
    if (mode == 0) out[0] = t / (float)n;
    else { const float e = expf(*logit_scale); out[0] = e <= 100.f ? t * e : 0.f; }
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int segclip_max_tokens_fwd(const float* x, float* out, int32_t* idx, int64_t B, int64_t T, int64_t D, void* stream) {
  SEGCLIP_REQUIRE(D % 4 == 0 && T >= 1, "max_tokens: D=%lld must be a multiple of 4, T >= 1", (long long)D);
  if (B == 0) return 0;
  hipLaunchKernelGGL(max_tokens_fwd_kernel, dim3((unsigned)cdiv(B * (D / 4), 128)), dim3(128), 0, ST, x, out, idx, B, (int)T, (int)D);
  SEGCLIP_CHECK_LAUNCH("max_tokens_fwd");
  return 0;
}
extern "C" int segclip_max_tokens_bwd(const float* dout, const int32_t* idx, float* dx, void* dx_bf16, int64_t B, int64_t T,
                                      int64_t D, void* stream) {
  SEGCLIP_REQUIRE(D % 4 == 0 && (dx || dx_bf16), "max_tokens_bwd: D=%lld must be a multiple of 4, one output required", (long long)D);
  if (B == 0) return 0;
  const int64_t total = B * T * (D / 4);
  const int64_t blocks = cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096;
  hipLaunchKernelGGL(max_tokens_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, ST, dout, idx, dx, (bf16_t*)dx_bf16, B, (int)T, (int)D);
  SEGCLIP_CHECK_LAUNCH("max_tokens_bwd");
  return 0;
}
extern "C" int segclip_l2norm_pair_fwd(const float* v, const float* t, float* both, float* norms, int64_t B, int64_t C, void* stream) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(l2norm_pair_fwd_kernel, dim3((unsigned)cdiv(2 * B, 4)), dim3(256), 0, ST, v, t, both, norms, B, (int)C);
  SEGCLIP_CHECK_LAUNCH("l2norm_pair_fwd");
  return 0;
}
extern "C" int segclip_l2norm_pair_bwd(const float* dboth, const float* dboth2, const float* both, const float* norms, float* dv,
                                       float* dt, int64_t B, int64_t C, void* stream) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(l2norm_pair_bwd_kernel, dim3((unsigned)cdiv(2 * B, 4)), dim3(256), 0, ST, dboth, dboth2, both, norms, dv, dt, B, (int)C);
  SEGCLIP_CHECK_LAUNCH("l2norm_pair_bwd");
  return 0;
}
extern "C" int segclip_clip_ce_fwd(const float* cosm, const float* logit_scale, float* lse, float* loss_rows, float* loss,
                                   int64_t B, int64_t N, int64_t label_offset, void* stream) {
  SEGCLIP_REQUIRE(B >= 1 && label_offset >= 0 && label_offset + B <= N, "clip_ce: labels out of range");
  hipLaunchKernelGGL(clip_ce_fwd_kernel, dim3((unsigned)cdiv(2 * B, 4)), dim3(256), 0, ST, cosm, logit_scale, lse, loss_rows, B, N, label_offset);
  SEGCLIP_CHECK_LAUNCH("clip_ce_fwd");
  hipLaunchKernelGGL(clip_reduce_kernel, dim3(1), dim3(1024), 0, ST, (const float*)loss_rows, 2 * B, logit_scale, loss, 0);
  SEGCLIP_CHECK_LAUNCH("clip_ce_fwd_reduce");
  return 0;
}
extern "C" int segclip_clip_ce_bwd(const float* cosm, const float* lse, const float* logit_scale, const float* g, float* dcos,
                                   float* ds_rows, float* dlogit_scale, int64_t B, int64_t N, int64_t label_offset, void* stream) {
  SEGCLIP_REQUIRE(B >= 1 && label_offset >= 0 && label_offset + B <= N, "clip_ce: labels out of range");
  hipLaunchKernelGGL(clip_ce_bwd_kernel, dim3((unsigned)cdiv(2 * B, 4)), dim3(256), 0, ST, cosm, lse, logit_scale, g, dcos, ds_rows, B, N,
                     label_offset);
  SEGCLIP_CHECK_LAUNCH("clip_ce_bwd");
  if (dlogit_scale) {
    hipLaunchKernelGGL(clip_reduce_kernel, dim3(1), dim3(1024), 0, ST, (const float*)ds_rows, 2 * B, logit_scale, dlogit_scale, 1);
    SEGCLIP_CHECK_LAUNCH("clip_ce_bwd_reduce");
  }
  return 0;
}
