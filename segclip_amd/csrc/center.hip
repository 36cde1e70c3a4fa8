// Learnable-center stage, the segment mean (reference modules/module_seg_vit.py:308-309):
//   outputs[b][g][:] = (sum_t hard[b][g][t] v[b][t][:]) / clamp_min(sum_t hard[b][g][t], 1)
// hard is the one-hot (straight-through Gumbel) assignment of every patch token to one of G <= 8 centers, so the "GEMM" is
// a segment sum by index: as einsum + sum + clamp + division + casts it was a batched M = 8 product on 128-row tiles plus six
// elementwise launches forward and two more batched products backward (~0.5 ms of kernel time per step for 0.6 GFLOP).
// Here: one launch each way, v read once (77 MB bf16 at B = 256).
//
// Backward (dout = d outputs, c = clamp_min(count, 1), dN = dout / c):
//   dv[b][t][:]    = dN[b][idx[b][t]][:]                                  (hard is one-hot)
//   dhard[b][g][t] = dN[b][g][:] . v[b][t][:]  +  dc[b][g],   dc = -[count >= 1] (dout[b][g][:] . outputs[b][g][:]) / c
// (the second term is the gradient through the normaliser; torch's clamp_min passes the gradient where count >= 1).
#include "common.h"

namespace {

constexpr int CG = 8;      // centers (SegViT: group_num = 8)

template <int DT> __device__ __forceinline__ f32x4 ld_v4(const void* p, int64_t i);
template <> __device__ __forceinline__ f32x4 ld_v4<SEGCLIP_F32>(const void* p, int64_t i) { return *reinterpret_cast<const f32x4*>((const float*)p + i); }
template <> __device__ __forceinline__ f32x4 ld_v4<SEGCLIP_BF16>(const void* p, int64_t i) {
  const u32x2 t = *reinterpret_cast<const u32x2*>((const bf16_t*)p + i);
  return f32x4{__uint_as_float(t[0] << 16), __uint_as_float(t[0] & 0xffff0000u), __uint_as_float(t[1] << 16), __uint_as_float(t[1] & 0xffff0000u)};
}
template <int DT> __device__ __forceinline__ void st_v4(void* p, int64_t i, f32x4 v);
template <> __device__ __forceinline__ void st_v4<SEGCLIP_F32>(void* p, int64_t i, f32x4 v) { *reinterpret_cast<f32x4*>((float*)p + i) = v; }
template <> __device__ __forceinline__ void st_v4<SEGCLIP_BF16>(void* p, int64_t i, f32x4 v) {
  *reinterpret_cast<u32x2*>((bf16_t*)p + i) = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
}

// forward: one wave per (sample, 256-column chunk); a lane owns 4 columns and the G accumulators of those columns; the
// token's center index is wave-uniform.  Tokens are summed in token order (= the k order of the exact-fp32 product).
template <int DT>
__global__ __launch_bounds__(64) void segmean_fwd_kernel(const uint8_t* __restrict__ idx, const void* __restrict__ v,
                                                         const float* __restrict__ counts, float* __restrict__ out, int G, int T, int D) {
  const int b = blockIdx.y, c = (blockIdx.x * 64 + threadIdx.x) * 4;
  if (c >= D) return;
  f32x4 acc[CG];
#pragma unroll
  for (int g = 0; g < CG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint8_t* ip = idx + (int64_t)b * T;
  const int64_t base = (int64_t)b * T * D + c;
  int t = 0;
  for (; t + 4 <= T; t += 4) {
    f32x4 x[4];
    int id[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { x[u] = ld_v4<DT>(v, base + (int64_t)(t + u) * D); id[u] = __builtin_amdgcn_readfirstlane((int)ip[t + u]); }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int g = 0; g < CG; ++g)
        if (id[u] == g) acc[g] += x[u];
  }
  for (; t < T; ++t) {
    const f32x4 x = ld_v4<DT>(v, base + (int64_t)t * D);
    const int id = __builtin_amdgcn_readfirstlane((int)ip[t]);
#pragma unroll
    for (int g = 0; g < CG; ++g)
      if (id == g) acc[g] += x;
  }
#pragma unroll
  for (int g = 0; g < CG; ++g)
    if (g < G) {
      const float cn = fmaxf(counts[(int64_t)b * G + g], 1.0f);
      *reinterpret_cast<f32x4*>(out + ((int64_t)b * G + g) * D + c) = acc[g] / cn;
    }
}

// backward: one workgroup (4 waves) per sample.  A wave owns every 4th token and ALL columns of it (a lane: 4 columns in each
// 256-column group, up to 1024 columns), so a token's G dot products are complete after the wave reduction; dN (G rows) stays
// in registers.  dhard leaves through LDS, coalesced along the token axis.
constexpr int CMAXG = 4;   // 256-column groups per row: D <= 1024
template <int DT>
__global__ __launch_bounds__(256) void segmean_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                          const uint8_t* __restrict__ idx, const void* __restrict__ v,
                                                          const float* __restrict__ counts, void* __restrict__ dv,
                                                          float* __restrict__ dhard, int G, int T, int D) {
  extern __shared__ float sm[];                 // [T][CG] dots, then [CG] dc
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncg = (D + 255) / 256;
  f32x4 dn[CMAXG][CG];
  float dcp[CG];
#pragma unroll
  for (int g = 0; g < CG; ++g) dcp[g] = 0.f;
#pragma unroll
  for (int k = 0; k < CMAXG; ++k) {
    const int c = k * 256 + lane * 4;
#pragma unroll
    for (int g = 0; g < CG; ++g) {
      dn[k][g] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (k < ncg && c < D && g < G) {
        const float cn = fmaxf(counts[(int64_t)b * G + g], 1.0f);
        const f32x4 d = *reinterpret_cast<const f32x4*>(dout + ((int64_t)b * G + g) * D + c);
        const f32x4 o = *reinterpret_cast<const f32x4*>(out + ((int64_t)b * G + g) * D + c);
        dn[k][g] = d / cn;
        dcp[g] += d[0] * o[0] + d[1] * o[1] + d[2] * o[2] + d[3] * o[3];
      }
    }
  }
  float* dcs = sm + (int64_t)T * CG;
#pragma unroll
  for (int g = 0; g < CG; ++g) {
    const float s = wave_sum(dcp[g]);             // every wave holds all columns: identical in all four
    if (wave == 0 && lane == 0) {
      const float cnt = g < G ? counts[(int64_t)b * G + g] : 0.f;
      dcs[g] = cnt >= 1.0f ? -s / fmaxf(cnt, 1.0f) : 0.f;
    }
  }
  const uint8_t* ip = idx + (int64_t)b * T;
  for (int t = wave; t < T; t += 4) {
    const int id = __builtin_amdgcn_readfirstlane((int)ip[t]);
    float p[CG];
#pragma unroll
    for (int g = 0; g < CG; ++g) p[g] = 0.f;
#pragma unroll
    for (int k = 0; k < CMAXG; ++k) {
      const int c = k * 256 + lane * 4;
      if (k < ncg && c < D) {
        const int64_t o = ((int64_t)b * T + t) * D + c;
        const f32x4 x = ld_v4<DT>(v, o);
        f32x4 g4 = dn[k][0];
#pragma unroll
        for (int g = 1; g < CG; ++g)
          if (id == g) g4 = dn[k][g];
        st_v4<DT>(dv, o, g4);
#pragma unroll
        for (int g = 0; g < CG; ++g) p[g] += dn[k][g][0] * x[0] + dn[k][g][1] * x[1] + dn[k][g][2] * x[2] + dn[k][g][3] * x[3];
      }
    }
    float mine = 0.f;
#pragma unroll
    for (int g = 0; g < CG; ++g) {
      const float s = wave_sum(p[g]);
      if (lane == g) mine = s;
    }
    if (lane < CG) sm[(int64_t)t * CG + lane] = mine;
  }
  __syncthreads();
  for (int i = tid; i < G * T; i += blockDim.x) {
    const int g = i / T, t = i - g * T;
    dhard[((int64_t)b * G + g) * T + t] = sm[(int64_t)t * CG + g] + dcs[g];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Per-sample token loops for D = NCG * 256 columns (768, 1024): one workgroup of 4 waves per sample, wave w takes tokens
// w, w + 4, ...; a lane owns 4 columns in each 256-column group; the G = 8 rows of the small operand (q, dN) live in
// registers; the NEXT token's row is in flight while the current one is processed.
// ---------------------------------------------------------------------------------------------------------------------
// sums of 8 per-lane values over the wave with 10 shuffles instead of 48: a reduce-scatter over lane bits 5, 4, 3 (each step
// halves the values a lane is responsible for), then three plain steps; on return every lane holds the total of value
// g = lane >> 3
__device__ __forceinline__ float wave_sum8(const float (&v)[8], int lane) {
  const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
  float w[4], u[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = (b5 ? v[i + 4] : v[i]) + __shfl_xor(b5 ? v[i] : v[i + 4], 32, 64);
#pragma unroll
  for (int i = 0; i < 2; ++i) u[i] = (b4 ? w[i + 2] : w[i]) + __shfl_xor(b4 ? w[i] : w[i + 2], 16, 64);
  float s = (b3 ? u[1] : u[0]) + __shfl_xor(b3 ? u[0] : u[1], 8, 64);
  s += __shfl_xor(s, 4, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 1, 64);
  return s;
}
__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

// Reconstruction of the token rows from the centers (MAE branch, reference modules/module_seg_vit.py:338-342):
// out[b][m][:] = sum_g a[b][m][g] * x[b][g][:], G = 8, fp32.  As a batched GEMM with K = 8 this ran on the exact-fp32 MFMA kernel at
// 0.8 ms per call; it is 8 FMAs per output element and one pass over the output.  One wave per (sample, 256-column chunk,
// token range): the sample's 8 center rows stay in registers, a[b][m][0..7] is wave-uniform (scalar loads).
__global__ __launch_bounds__(64) void recon_mix_fwd_kernel(const float* __restrict__ a, const float* __restrict__ x, float* __restrict__ out,
                                                           int M, int D4, int chunks, int mseg) {
  const int b = blockIdx.x / chunks, c = (blockIdx.x % chunks) * 64 + threadIdx.x;
  if (c >= D4) return;
  f32x4 xg[CG];
#pragma unroll
  for (int g = 0; g < CG; ++g) xg[g] = reinterpret_cast<const f32x4*>(x)[((int64_t)b * CG + g) * D4 + c];
  const int m0 = blockIdx.y * mseg, m1 = m0 + mseg < M ? m0 + mseg : M;
  const float* ab = a + ((int64_t)b * M + m0) * CG;
  f32x4* ob = reinterpret_cast<f32x4*>(out) + ((int64_t)b * M + m0) * D4 + c;
  for (int m = m0; m < m1; ++m, ab += CG, ob += D4) {
    f32x4 acc = xg[0] * ab[0];
#pragma unroll
    for (int g = 1; g < CG; ++g) acc += xg[g] * ab[g];
    *ob = acc;
  }
}
// backward: dx[b][g][:] = sum_m a[b][m][g] * dout[b][m][:]  and  da[b][m][g] = dout[b][m][:] . x[b][g][:].  One workgroup per
// sample, one lane per 4 columns (D / 4 lanes, whole waves): the token loop carries the dx accumulators; the 8 dot products of a
// token are reduced inside each wave (wave_sum8) and across the waves through a double-buffered LDS array (one barrier per token).
__global__ __launch_bounds__(1024) void recon_mix_bwd_kernel(const float* __restrict__ a, const float* __restrict__ x,
                                                             const float* __restrict__ dout, float* __restrict__ da, float* __restrict__ dx,
                                                             int M, int D4) {
  __shared__ float red[2][16][CG];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = blockDim.x >> 6;
  const bool live = t < D4;
  f32x4 xg[CG], acc[CG];
#pragma unroll
  for (int g = 0; g < CG; ++g) {
    xg[g] = live ? reinterpret_cast<const f32x4*>(x)[((int64_t)b * CG + g) * D4 + t] : f32x4{0.f, 0.f, 0.f, 0.f};
    acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float* ab = a + (int64_t)b * M * CG;
  const f32x4* gb = reinterpret_cast<const f32x4*>(dout) + (int64_t)b * M * D4 + t;
  float* dab = da + (int64_t)b * M * CG;
  for (int m = 0; m < M; ++m) {
    const f32x4 g4 = live ? gb[(int64_t)m * D4] : f32x4{0.f, 0.f, 0.f, 0.f};
    float pd[CG];
#pragma unroll
    for (int g = 0; g < CG; ++g) {
      acc[g] += g4 * ab[m * CG + g];
      pd[g] = dot4(g4, xg[g]);
    }
    const float s = wave_sum8(pd, lane);          // lane holds the wave's total of value lane >> 3
    if ((lane & 7) == 0) red[m & 1][wave][lane >> 3] = s;
    __syncthreads();
    if (t < CG) {
      float v = red[m & 1][0][t];
      for (int w = 1; w < nw; ++w) v += red[m & 1][w][t];
      dab[m * CG + t] = v;
    }
  }
  if (live) {
#pragma unroll
    for (int g = 0; g < CG; ++g) reinterpret_cast<f32x4*>(dx)[((int64_t)b * CG + g) * D4 + t] = acc[g];
  }
}

// assignment logits attn[b][g][t] = q[b][g][:] . k[b][t][:]  (reference modules/module_seg_vit.py:304, un-scaled), fp32
template <int NCG>
__global__ __launch_bounds__(256) void center_logits_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                float* __restrict__ attn, int T, int tchunk) {
  extern __shared__ float sm[];                 // [tchunk][CG]
  constexpr int D = NCG * 256;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // blockIdx.y: a token range of the sample (small inference batches: one workgroup per sample leaves most CUs idle; every
  // logit is its own dot product, so the split does not change a bit)
  const int tb = blockIdx.y * tchunk, te = tb + tchunk < T ? tb + tchunk : T;
  f32x4 qr[NCG][CG];
#pragma unroll
  for (int kk = 0; kk < NCG; ++kk)
#pragma unroll
    for (int g = 0; g < CG; ++g) qr[kk][g] = *reinterpret_cast<const f32x4*>(q + ((int64_t)b * CG + g) * D + kk * 256 + lane * 4);
  const float* kb = k + (int64_t)b * T * D + lane * 4;
  f32x4 xn[NCG];
  if (tb + wave < te) {
#pragma unroll
    for (int kk = 0; kk < NCG; ++kk) xn[kk] = *reinterpret_cast<const f32x4*>(kb + (int64_t)(tb + wave) * D + kk * 256);
  }
  for (int t = tb + wave; t < te; t += 4) {
    f32x4 x[NCG];
#pragma unroll
    for (int kk = 0; kk < NCG; ++kk) x[kk] = xn[kk];
    if (t + 4 < te) {
#pragma unroll
      for (int kk = 0; kk < NCG; ++kk) xn[kk] = *reinterpret_cast<const f32x4*>(kb + (int64_t)(t + 4) * D + kk * 256);
    }
    float p[CG];
#pragma unroll
    for (int g = 0; g < CG; ++g) {
      p[g] = 0.f;
#pragma unroll
      for (int kk = 0; kk < NCG; ++kk) p[g] += dot4(qr[kk][g], x[kk]);
    }
    const float s = wave_sum8(p, lane);
    if ((lane & 7) == 0) sm[(int64_t)(t - tb) * CG + (lane >> 3)] = s;
  }
  __syncthreads();
  const int n = te - tb;
  for (int i = tid; i < CG * n; i += blockDim.x) {
    const int g = i / n, t = i - g * n;
    attn[((int64_t)b * CG + g) * T + tb + t] = sm[(int64_t)t * CG + g];
  }
}
// dq[b][g][:] = sum_t dl[b][g][t] k[b][t][:] ; dk[b][t][:] = sum_g dl[b][g][t] q[b][g][:]   (fp32)
template <int NCG>
__global__ __launch_bounds__(256) void center_logits_bwd_kernel(const float* __restrict__ dl, const float* __restrict__ q,
                                                                const float* __restrict__ k, float* __restrict__ dq,
                                                                float* __restrict__ dk, int T) {
  extern __shared__ float sm[];                 // [CG][T] dl of this sample, then [CG][D] dq accumulators
  constexpr int D = NCG * 256;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* dls = sm;
  float* acc_s = sm + (int64_t)CG * T;
  for (int i = tid; i < CG * T; i += blockDim.x) dls[i] = dl[(int64_t)b * CG * T + i];
  f32x4 qr[NCG][CG], acc[NCG][CG];
#pragma unroll
  for (int kk = 0; kk < NCG; ++kk)
#pragma unroll
    for (int g = 0; g < CG; ++g) {
      qr[kk][g] = *reinterpret_cast<const f32x4*>(q + ((int64_t)b * CG + g) * D + kk * 256 + lane * 4);
      acc[kk][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  __syncthreads();
  const float* kb = k + (int64_t)b * T * D + lane * 4;
  float* dkb = dk + (int64_t)b * T * D + lane * 4;
  f32x4 xn[NCG];
  if (wave < T) {
#pragma unroll
    for (int kk = 0; kk < NCG; ++kk) xn[kk] = *reinterpret_cast<const f32x4*>(kb + (int64_t)wave * D + kk * 256);
  }
  for (int t = wave; t < T; t += 4) {
    f32x4 x[NCG];
#pragma unroll
    for (int kk = 0; kk < NCG; ++kk) x[kk] = xn[kk];
    if (t + 4 < T) {
#pragma unroll
      for (int kk = 0; kk < NCG; ++kk) xn[kk] = *reinterpret_cast<const f32x4*>(kb + (int64_t)(t + 4) * D + kk * 256);
    }
    float w[CG];
#pragma unroll
    for (int g = 0; g < CG; ++g) w[g] = dls[(int64_t)g * T + t];      // the same address in every lane: LDS broadcast
#pragma unroll
    for (int kk = 0; kk < NCG; ++kk) {
      f32x4 o = w[0] * qr[kk][0];
#pragma unroll
      for (int g = 1; g < CG; ++g) o += w[g] * qr[kk][g];
      *reinterpret_cast<f32x4*>(dkb + (int64_t)t * D + kk * 256) = o;
#pragma unroll
      for (int g = 0; g < CG; ++g) acc[kk][g] += w[g] * x[kk];
    }
  }
  // dq: the four waves' partial sums meet in LDS in wave order (deterministic)
  for (int w4 = 0; w4 < 4; ++w4) {
    if (wave == w4) {
#pragma unroll
      for (int kk = 0; kk < NCG; ++kk)
#pragma unroll
        for (int g = 0; g < CG; ++g) {
          f32x4* p = reinterpret_cast<f32x4*>(acc_s + (int64_t)g * D + kk * 256 + lane * 4);
          *p = w4 == 0 ? acc[kk][g] : *p + acc[kk][g];
        }
    }
    __syncthreads();
  }
  for (int i = tid * 4; i < CG * D; i += blockDim.x * 4)
    *reinterpret_cast<f32x4*>(dq + (int64_t)b * CG * D + i) = *reinterpret_cast<const f32x4*>(acc_s + i);
}

// segment mean, the same token loop (forward: G accumulators per column; backward: dv gather + dhard dots)
template <int DT, int NCG>
__global__ __launch_bounds__(256) void segmean_fwd_fast_kernel(const uint8_t* __restrict__ idx, const void* __restrict__ v,
                                                               const float* __restrict__ counts, float* __restrict__ out, int T) {
  extern __shared__ float sm[];                 // [CG][D] accumulators
  constexpr int D = NCG * 256;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x4 acc[NCG][CG];
#pragma unroll
  for (int kk = 0; kk < NCG; ++kk)
#pragma unroll
    for (int g = 0; g < CG; ++g) acc[kk][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint8_t* ip = idx + (int64_t)b * T;
  const int64_t base = (int64_t)b * T * D + lane * 4;
  f32x4 xn[NCG];
  if (wave < T) {
#pragma unroll
    for (int kk = 0; kk < NCG; ++kk) xn[kk] = ld_v4<DT>(v, base + (int64_t)wave * D + kk * 256);
  }
  for (int t = wave; t < T; t += 4) {
    f32x4 x[NCG];
#pragma unroll
    for (int kk = 0; kk < NCG; ++kk) x[kk] = xn[kk];
    if (t + 4 < T) {
#pragma unroll
      for (int kk = 0; kk < NCG; ++kk) xn[kk] = ld_v4<DT>(v, base + (int64_t)(t + 4) * D + kk * 256);
    }
    const int id = __builtin_amdgcn_readfirstlane((int)ip[t]);
#pragma unroll
    for (int g = 0; g < CG; ++g)
      if (id == g) {
#pragma unroll
        for (int kk = 0; kk < NCG; ++kk) acc[kk][g] += x[kk];
      }
  }
  for (int w4 = 0; w4 < 4; ++w4) {               // wave order: tokens w, w+4, .. of wave 0 first (fixed, not token order)
    if (wave == w4) {
#pragma unroll
      for (int kk = 0; kk < NCG; ++kk)
#pragma unroll
        for (int g = 0; g < CG; ++g) {
          f32x4* p = reinterpret_cast<f32x4*>(sm + (int64_t)g * D + kk * 256 + lane * 4);
          *p = w4 == 0 ? acc[kk][g] : *p + acc[kk][g];
        }
    }
    __syncthreads();
  }
  for (int i = tid * 4; i < CG * D; i += blockDim.x * 4) {
    const float cn = fmaxf(counts[(int64_t)b * CG + i / D], 1.0f);
    *reinterpret_cast<f32x4*>(out + (int64_t)b * CG * D + i) = *reinterpret_cast<const f32x4*>(sm + i) / cn;
  }
}
template <int DT, int NCG>
__global__ __launch_bounds__(256) void segmean_bwd_fast_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                               const uint8_t* __restrict__ idx, const void* __restrict__ v,
                                                               const float* __restrict__ counts, void* __restrict__ dv,
                                                               float* __restrict__ dhard, int T) {
  extern __shared__ float sm[];                 // [T][CG] dots, then [CG] dc
  constexpr int D = NCG * 256;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x4 dn[NCG][CG];
  float dcp[CG];
#pragma unroll
  for (int g = 0; g < CG; ++g) {
    dcp[g] = 0.f;
    const float cn = fmaxf(counts[(int64_t)b * CG + g], 1.0f);
#pragma unroll
    for (int kk = 0; kk < NCG; ++kk) {
      const int64_t o = ((int64_t)b * CG + g) * D + kk * 256 + lane * 4;
      const f32x4 d = *reinterpret_cast<const f32x4*>(dout + o);
      dn[kk][g] = d / cn;
      dcp[g] += dot4(d, *reinterpret_cast<const f32x4*>(out + o));
    }
  }
  float* dcs = sm + (int64_t)T * CG;
  {
    const float s = wave_sum8(dcp, lane);        // every wave holds all columns: identical in all four
    if (wave == 0 && (lane & 7) == 0) {
      const int g = lane >> 3;
      const float cnt = counts[(int64_t)b * CG + g];
      dcs[g] = cnt >= 1.0f ? -s / fmaxf(cnt, 1.0f) : 0.f;
    }
  }
  const uint8_t* ip = idx + (int64_t)b * T;
  const int64_t base = (int64_t)b * T * D + lane * 4;
  f32x4 xn[NCG];
  if (wave < T) {
#pragma unroll
    for (int kk = 0; kk < NCG; ++kk) xn[kk] = ld_v4<DT>(v, base + (int64_t)wave * D + kk * 256);
  }
  for (int t = wave; t < T; t += 4) {
    f32x4 x[NCG];
#pragma unroll
    for (int kk = 0; kk < NCG; ++kk) x[kk] = xn[kk];
    if (t + 4 < T) {
#pragma unroll
      for (int kk = 0; kk < NCG; ++kk) xn[kk] = ld_v4<DT>(v, base + (int64_t)(t + 4) * D + kk * 256);
    }
    const int id = __builtin_amdgcn_readfirstlane((int)ip[t]);
    float p[CG];
#pragma unroll
    for (int g = 0; g < CG; ++g) p[g] = 0.f;
#pragma unroll
    for (int kk = 0; kk < NCG; ++kk) {
      f32x4 g4 = dn[kk][0];
#pragma unroll
      for (int g = 1; g < CG; ++g)
        if (id == g) g4 = dn[kk][g];
      st_v4<DT>(dv, base + (int64_t)t * D + kk * 256, g4);
#pragma unroll
      for (int g = 0; g < CG; ++g) p[g] += dot4(dn[kk][g], x[kk]);
    }
    const float s = wave_sum8(p, lane);
    if ((lane & 7) == 0) sm[(int64_t)t * CG + (lane >> 3)] = s;
  }
  __syncthreads();
  for (int i = tid; i < CG * T; i += blockDim.x) {
    const int g = i / T, t = i - g * T;
    dhard[((int64_t)b * CG + g) * T + t] = sm[(int64_t)t * CG + g] + dcs[g];
  }
}

}  // namespace

#define ST ((hipStream_t)stream)
// outputs (B, G, D) fp32 = segment mean of the rows of v (B, T, D; fp32 or bf16) by the center index idx (B, T); counts (B, G) =
// tokens per center (fp32, as segclip_assign_fwd leaves them).  G <= 8, D a multiple of 4.
extern "C" int segclip_segmean_fwd(const uint8_t* idx, const void* v, int v_dtype, const float* counts, float* out, int64_t B,
                                   int64_t G, int64_t T, int64_t D, void* stream) {
  SEGCLIP_REQUIRE(G >= 1 && G <= CG && D % 4 == 0 && T >= 1, "segmean: G=%lld (<= %d), D=%lld (multiple of 4)", (long long)G, CG, (long long)D);
  SEGCLIP_REQUIRE(v_dtype == SEGCLIP_F32 || v_dtype == SEGCLIP_BF16, "segmean: v must be fp32 or bf16");
  if (B == 0) return 0;
  if (G == CG && (D == 768 || D == 1024)) {     // the towers' widths: per-sample token loop (4 waves, next row in flight)
    const size_t lds = (size_t)CG * D * sizeof(float);
#define SEGF(DT_, NCG_) hipLaunchKernelGGL((segmean_fwd_fast_kernel<DT_, NCG_>), dim3((unsigned)B), dim3(256), lds, ST, idx, v, counts, out, (int)T)
    if (v_dtype == SEGCLIP_BF16) { if (D == 768) SEGF(SEGCLIP_BF16, 3); else SEGF(SEGCLIP_BF16, 4); }
    else { if (D == 768) SEGF(SEGCLIP_F32, 3); else SEGF(SEGCLIP_F32, 4); }
#undef SEGF
    SEGCLIP_CHECK_LAUNCH("segmean_fwd");
    return 0;
  }
  const dim3 grid((unsigned)cdiv(D, 256), (unsigned)B);
  if (v_dtype == SEGCLIP_BF16) hipLaunchKernelGGL(segmean_fwd_kernel<SEGCLIP_BF16>, grid, dim3(64), 0, ST, idx, v, counts, out, (int)G, (int)T, (int)D);
  else hipLaunchKernelGGL(segmean_fwd_kernel<SEGCLIP_F32>, grid, dim3(64), 0, ST, idx, v, counts, out, (int)G, (int)T, (int)D);
  SEGCLIP_CHECK_LAUNCH("segmean_fwd");
  return 0;
}
// dv (B, T, D; dtype of v) and dhard (B, G, T) fp32 from dout (B, G, D) fp32; `out` = the forward result.  D <= 1024.
extern "C" int segclip_segmean_bwd(const float* dout, const float* out, const uint8_t* idx, const void* v, int v_dtype,
                                   const float* counts, void* dv, float* dhard, int64_t B, int64_t G, int64_t T, int64_t D,
                                   void* stream) {
  SEGCLIP_REQUIRE(G >= 1 && G <= CG && D % 4 == 0 && D <= CMAXG * 256 && T >= 1, "segmean_bwd: G=%lld (<= %d), D=%lld (multiple of 4, <= %d)",
                  (long long)G, CG, (long long)D, CMAXG * 256);
  SEGCLIP_REQUIRE(v_dtype == SEGCLIP_F32 || v_dtype == SEGCLIP_BF16, "segmean: v must be fp32 or bf16");
  if (B == 0) return 0;
  const size_t lds = (size_t)(T + 1) * CG * sizeof(float);
  SEGCLIP_REQUIRE(lds <= 60000, "segmean_bwd: T=%lld too long for the LDS staging", (long long)T);
  if (G == CG && (D == 768 || D == 1024)) {
#define SEGB(DT_, NCG_) hipLaunchKernelGGL((segmean_bwd_fast_kernel<DT_, NCG_>), dim3((unsigned)B), dim3(256), lds, ST, dout, out, idx, v, counts, dv, dhard, (int)T)
    if (v_dtype == SEGCLIP_BF16) { if (D == 768) SEGB(SEGCLIP_BF16, 3); else SEGB(SEGCLIP_BF16, 4); }
    else { if (D == 768) SEGB(SEGCLIP_F32, 3); else SEGB(SEGCLIP_F32, 4); }
#undef SEGB
    SEGCLIP_CHECK_LAUNCH("segmean_bwd");
    return 0;
  }
  if (v_dtype == SEGCLIP_BF16)
    hipLaunchKernelGGL(segmean_bwd_kernel<SEGCLIP_BF16>, dim3((unsigned)B), dim3(256), lds, ST, dout, out, idx, v, counts, dv, dhard, (int)G, (int)T, (int)D);
  else
    hipLaunchKernelGGL(segmean_bwd_kernel<SEGCLIP_F32>, dim3((unsigned)B), dim3(256), lds, ST, dout, out, idx, v, counts, dv, dhard, (int)G, (int)T, (int)D);
  SEGCLIP_CHECK_LAUNCH("segmean_bwd");
  return 0;
}

// out (B,M,D) = a (B,M,8) @ x (B,8,D), fp32, and its backward (reference modules/module_seg_vit.py:342: the MAE branch's
// ReconstructLayer).  G = 8 and D a multiple of 4, at most 4096 columns; else SEGCLIP_ERR_UNSUPPORTED (the caller uses segclip_gemm).
extern "C" int segclip_recon_mix_fwd(const float* a, const float* x, float* out, int64_t B, int64_t M, int64_t G, int64_t D, void* stream) {
  if (G != CG || D % 4 != 0 || D > 4096 || D < 4) {
    segclip_set_error("recon_mix: G=%lld, D=%lld not covered", (long long)G, (long long)D);
    return SEGCLIP_ERR_UNSUPPORTED;
  }
  if (B * M == 0) return 0;
  const int D4 = (int)(D / 4), chunks = (D4 + 63) / 64;
  const int nseg = (int)(M >= 64 ? 4 : 1), mseg = (int)((M + nseg - 1) / nseg);
  hipLaunchKernelGGL(recon_mix_fwd_kernel, dim3((unsigned)(B * chunks), (unsigned)nseg), dim3(64), 0, ST, a, x, out, (int)M, D4, chunks, mseg);
  SEGCLIP_CHECK_LAUNCH("recon_mix_fwd");
  return 0;
}
extern "C" int segclip_recon_mix_bwd(const float* a, const float* x, const float* dout, float* da, float* dx, int64_t B, int64_t M,
                                     int64_t G, int64_t D, void* stream) {
  if (G != CG || D % 4 != 0 || D > 4096 || D < 4) {
    segclip_set_error("recon_mix: G=%lld, D=%lld not covered", (long long)G, (long long)D);
    return SEGCLIP_ERR_UNSUPPORTED;
  }
  if (B == 0) return 0;
  const int D4 = (int)(D / 4), threads = ((D4 + 63) / 64) * 64;
  hipLaunchKernelGGL(recon_mix_bwd_kernel, dim3((unsigned)B), dim3((unsigned)threads), 0, ST, a, x, dout, da, dx, (int)M, D4);
  SEGCLIP_CHECK_LAUNCH("recon_mix_bwd");
  return 0;
}
// Assignment logits of the center stage, attn[b][g][t] = q[b][g][:] . k[b][t][:] (fp32; reference modules/module_seg_vit.py:304),
// and their backward dq = dl k, dk = dl^T q, as per-sample token loops.  G = 8, D = 768 or 1024 (else SEGCLIP_ERR_UNSUPPORTED:
// the caller uses segclip_gemm).  The summation order over D differs from the exact-fp32 GEMM's: used in bf16 mode only.
extern "C" int segclip_center_logits_fwd(const float* q, const float* k, float* attn, int64_t B, int64_t G, int64_t T, int64_t D,
                                         void* stream) {
  if (!(G == CG && (D == 768 || D == 1024)) || (size_t)T * CG * sizeof(float) > 60000) {
    segclip_set_error("center_logits: G=%lld, D=%lld not covered", (long long)G, (long long)D);
    return SEGCLIP_ERR_UNSUPPORTED;
  }
  if (B == 0) return 0;
  // fewer samples than CUs (inference batches): token ranges of at least 64 tokens over blockIdx.y
  int nchunk = B >= 128 ? 1 : (int)((255 + B) / B);
  if (nchunk > (int)((T + 63) / 64)) nchunk = (int)((T + 63) / 64);
  if (nchunk < 1) nchunk = 1;
  const int tchunk = (int)((T + nchunk - 1) / nchunk);
  nchunk = (int)((T + tchunk - 1) / tchunk);
  const size_t lds = (size_t)tchunk * CG * sizeof(float);
  if (D == 768) hipLaunchKernelGGL(center_logits_fwd_kernel<3>, dim3((unsigned)B, (unsigned)nchunk), dim3(256), lds, ST, q, k, attn, (int)T, tchunk);
  else hipLaunchKernelGGL(center_logits_fwd_kernel<4>, dim3((unsigned)B, (unsigned)nchunk), dim3(256), lds, ST, q, k, attn, (int)T, tchunk);
  SEGCLIP_CHECK_LAUNCH("center_logits_fwd");
  return 0;
}
extern "C" int segclip_center_logits_bwd(const float* dl, const float* q, const float* k, float* dq, float* dk, int64_t B,
                                         int64_t G, int64_t T, int64_t D, void* stream) {
  const size_t lds = ((size_t)T * CG + (size_t)CG * D) * sizeof(float);
  if (!(G == CG && (D == 768 || D == 1024)) || lds > 64000) {
    segclip_set_error("center_logits: G=%lld, D=%lld not covered", (long long)G, (long long)D);
    return SEGCLIP_ERR_UNSUPPORTED;
  }
  if (B == 0) return 0;
  if (D == 768) hipLaunchKernelGGL(center_logits_bwd_kernel<3>, dim3((unsigned)B), dim3(256), lds, ST, dl, q, k, dq, dk, (int)T);
  else hipLaunchKernelGGL(center_logits_bwd_kernel<4>, dim3((unsigned)B), dim3(256), lds, ST, dl, q, k, dq, dk, (int)T);
  SEGCLIP_CHECK_LAUNCH("center_logits_bwd");
  return 0;
}
