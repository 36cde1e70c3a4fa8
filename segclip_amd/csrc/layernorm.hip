// LayerNorm forward / backward (HBM-bound).  One wave per row, the row lives in registers
// (cols <= 2048), fp32 statistics, 16-byte (fp32) / 8-byte (bf16) vector loads, wave shuffles for the
// two reductions.  Backward fuses the residual-gradient add (dx = LN'(dy) + dres) and produces
// dgamma/dbeta through a deterministic two-stage reduction (per-block partials -> column sums).
#include <stdlib.h>

#include "common.h"
#include "reduce_rows.h"

namespace {

// Row mapping of ONE operand (forward: the output y; backward: the incoming gradient dy): row r of the kernel's row space
// lives at row (r / seg_in) * seg_out + off + r % seg_in of the operand; seg_in = 0: identity.  It lets LayerNorm write its
// output straight into a slice of every sample of a larger (B, S, D) buffer - LN(cat([q, inputs], dim=1)) without the cat
// (learnable-center cross-attention, reference modules/module_seg_vit.py:294-296,211).
struct RowMap {
  int64_t seg_in, seg_out, off;
  __device__ __forceinline__ int64_t at(int64_t r) const { return seg_in ? (r / seg_in) * seg_out + off + r % seg_in : r; }
};

constexpr int MAXV = 8;          // float4 slots per lane -> cols <= 8*4*64 = 2048
constexpr int WAVES = 4;         // rows per block iteration

__device__ __forceinline__ f32x4 load4(const void* base, int dtype, int64_t idx) {
  f32x4 r;
  if (dtype == SEGCLIP_F32) {
    r = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
  } else {
    const u32x2 t = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(base) + idx);
    r[0] = __uint_as_float(t[0] << 16); r[1] = __uint_as_float(t[0] & 0xffff0000u);
    r[2] = __uint_as_float(t[1] << 16); r[3] = __uint_as_float(t[1] & 0xffff0000u);
  }
  return r;
}
__device__ __forceinline__ void store4(void* base, int dtype, int64_t idx, f32x4 v) {
  if (dtype == SEGCLIP_F32) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx) = v;
  } else {
    u32x2 t;
    t[0] = pack2bf(v[0], v[1]); t[1] = pack2bf(v[2], v[3]);
    *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(base) + idx) = t;
  }
}

template <int NV>
__global__ __launch_bounds__(WAVES * 64) void ln_fwd_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, void* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int64_t rows, int cols, float eps, int xd, int yd, RowMap om) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int nv = NV;
  for (int64_t row = (int64_t)blockIdx.x * WAVES + wave; row < rows; row += (int64_t)gridDim.x * WAVES) {
    const int64_t orow = om.at(row);
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (i < nv && c < cols) {
        v[i] = load4(x, xd, row * cols + c);
        s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
      } else {
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const float mu = wave_sum(s) / cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (i < nv && c < cols) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mu; q += d * d; }
      }
    }
    const float rs = rsqrtf(wave_sum(q) / cols + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (i < nv && c < cols) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mu) * rs * g[j] + b[j];
        store4(y, yd, orow * cols + c, o);
      }
    }
  }
}

// raw (as loaded) 4-element group of a row: fp32 -> 16 bytes, bf16 -> 8 bytes; converted where it is consumed, so the
// software pipeline below keeps the NEXT row in half the registers
template <int DT> struct Raw4;
template <> struct Raw4<SEGCLIP_F32> {
  f32x4 v;
  __device__ __forceinline__ void ld(const void* base, int64_t idx) { v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx); }
  __device__ __forceinline__ f32x4 get() const { return v; }
};
template <> struct Raw4<SEGCLIP_BF16> {
  u32x2 v;
  __device__ __forceinline__ void ld(const void* base, int64_t idx) { v = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(base) + idx); }
  __device__ __forceinline__ f32x4 get() const {
    return f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16),
                 __uint_as_float(v[1] & 0xffff0000u)};
  }
};

// Backward.  One wave per row; a wave walks rows  blockIdx*WAVES + wave + k*gridDim*WAVES  with a two-deep software
// pipeline: ALL loads of row k+1 (x, dy, dres, mean, rstd) are issued before row k is reduced and stored, so every
// wave always has a full row (10-16 bytes per element) in flight and the two dependent wave reductions of a row are
// covered by the next row's memory time (the first version issued the dres load after the reductions: two exposed
// round trips per row, 3.3-3.5 TB/s).  gamma stays in registers.  Element types are template parameters (raw bf16
// data is kept packed until used).  HAS_RES: dx += dres, and the column sums of dres are accumulated.
template <int NV, int DYD, int XD, int DXD, bool HAS_RES, bool HAS_DX2>
__global__ __launch_bounds__(WAVES * 64) void ln_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const void* __restrict__ dres, void* __restrict__ dx,
                                                            void* __restrict__ dx2, float* __restrict__ part,
                                                            int64_t rows, int cols, RowMap dmap) {
  __shared__ float red[WAVES][64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 ag[NV], ab[NV], ar[NV], gm[NV];
  bool on[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ag[i] = f32x4{0.f, 0.f, 0.f, 0.f}; ab[i] = f32x4{0.f, 0.f, 0.f, 0.f}; ar[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int c = (lane + 64 * i) * 4;
    on[i] = c < cols;
    gm[i] = on[i] ? *reinterpret_cast<const f32x4*>(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  int64_t row = (int64_t)blockIdx.x * WAVES + wave;
  Raw4<XD> nx[NV]; Raw4<DYD> nd[NV]; Raw4<DXD> nr[NV];
  float nmu = 0.f, nrs = 0.f;
  auto issue = [&](int64_t r) {
    nmu = mean[r]; nrs = rstd[r];
    const int64_t gr = dmap.at(r);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (on[i]) {
        const int64_t idx = r * cols + (lane + 64 * i) * 4;
        nx[i].ld(x, idx);
        nd[i].ld(dy, gr * cols + (lane + 64 * i) * 4);
        if (HAS_RES) nr[i].ld(dres, idx);
      }
    }
  };
  if (row < rows) issue(row);
  for (; row < rows; row += stride) {
    // take over the landed row, then put the next one in flight
    Raw4<XD> cx[NV]; Raw4<DYD> cd[NV]; Raw4<DXD> cr[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { cx[i] = nx[i]; cd[i] = nd[i]; if (HAS_RES) cr[i] = nr[i]; }
    const float mu = nmu, rs = nrs;
    if (row + stride < rows) issue(row + stride);
    f32x4 xh[NV], gg[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (on[i]) {
        const f32x4 xv = cx[i].get(), d = cd[i].get();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[i][j] = (xv[j] - mu) * rs;
          gg[i][j] = d[j] * gm[i][j];
          s1 += gg[i][j];
          s2 += gg[i][j] * xh[i][j];
          ag[i][j] += d[j] * xh[i][j];
          ab[i][j] += d[j];
        }
      }
    }
    const float c1 = wave_sum(s1) / cols, c2 = wave_sum(s2) / cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (on[i]) {
        const int64_t idx = row * cols + (lane + 64 * i) * 4;
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rs * (gg[i][j] - c1 - xh[i][j] * c2);
        if (HAS_RES) {
          const f32x4 r = cr[i].get();
#pragma unroll
          for (int j = 0; j < 4; ++j) { o[j] += r[j]; ar[i][j] += r[j]; }
        }
        store4(dx, DXD, idx, o);
        if (HAS_DX2) store4(dx2, SEGCLIP_BF16, idx, o);
      }
    }
  }
  // block partials: part[blockIdx][0] = dgamma, [1] = dbeta, [2] = column sums of dres
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    for (int pass = 0; pass < (HAS_RES ? 3 : 2); ++pass) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wave][lane * 4 + j] = pass == 0 ? ag[i][j] : (pass == 1 ? ab[i][j] : ar[i][j]);
      __syncthreads();
      if (wave == 0) {
        const int c = (lane + 64 * i) * 4;
        if (c < cols) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) t += red[w][lane * 4 + j];
            part[((int64_t)blockIdx.x * 3 + pass) * cols + c + j] = t;
          }
        }
      }
    }
  }
}

// ---- several affine outputs of ONE normalisation -------------------------------------------------------------------------
// The learnable-center stage normalises the same (B*T, D) token rows three times with different (gamma, beta): `self.norm`
// feeding the k/v group convolutions and `ln_1(k)` of each of the two cross-attention layers (reference
// modules/module_seg_vit.py:289,294-296,211).  mean/rstd are the same for the three, so the forward reads the row once and
// writes three outputs, and the backward - linear in dy_k * gamma_k - reads x once, sums the three incoming gradients inside
// the row and writes ONE dx (it was 3 x (x read + dx written) + 2 fp32 adds of B*T*D).  Each output has its own row mapping.
constexpr int LN_MULTI = 3;
struct LnMultiFwd { const float* gamma[LN_MULTI]; const float* beta[LN_MULTI]; void* y[LN_MULTI]; RowMap map[LN_MULTI]; };
struct LnMultiBwd { const float* gamma[LN_MULTI]; const void* dy[LN_MULTI]; RowMap map[LN_MULTI]; };

template <int NV, int YD>
__global__ __launch_bounds__(WAVES * 64) void ln_fwd_multi_kernel(const float* __restrict__ x, LnMultiFwd a, float* __restrict__ mean,
                                                                  float* __restrict__ rstd, int64_t rows, int cols, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * WAVES + wave; row < rows; row += (int64_t)gridDim.x * WAVES) {
    f32x4 v[NV];
    bool on[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      on[i] = c < cols;
      v[i] = on[i] ? *reinterpret_cast<const f32x4*>(x + row * cols + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mu = wave_sum(s) / cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (on[i]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[i][j] -= mu; q += v[i][j] * v[i][j]; }
      }
    }
    const float rs = rsqrtf(wave_sum(q) / cols + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int k = 0; k < LN_MULTI; ++k) {
      const int64_t orow = a.map[k].at(row);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (on[i]) {
          const f32x4 g = *reinterpret_cast<const f32x4*>(a.gamma[k] + c);
          const f32x4 b = *reinterpret_cast<const f32x4*>(a.beta[k] + c);
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = v[i][j] * rs * g[j] + b[j];
          store4(a.y[k], YD, orow * cols + c, o);
        }
      }
    }
  }
}

// backward of the above: the pipelined row walk of ln_bwd_kernel with LN_MULTI incoming gradients.
// part[blockIdx][2k] = dgamma_k, [2k+1] = dbeta_k
template <int NV, int DYD>
__global__ __launch_bounds__(WAVES * 64) void ln_bwd_multi_kernel(const float* __restrict__ x, LnMultiBwd a,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  float* __restrict__ dx, float* __restrict__ part, int64_t rows,
                                                                  int cols) {
  __shared__ float red[WAVES][64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 ag[LN_MULTI][NV], ab[LN_MULTI][NV];
  bool on[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    on[i] = (lane + 64 * i) * 4 < cols;
#pragma unroll
    for (int k = 0; k < LN_MULTI; ++k) { ag[k][i] = f32x4{0.f, 0.f, 0.f, 0.f}; ab[k][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  int64_t row = (int64_t)blockIdx.x * WAVES + wave;
  f32x4 nx[NV];
  Raw4<DYD> nd[LN_MULTI][NV];
  float nmu = 0.f, nrs = 0.f;
  auto issue = [&](int64_t r) {
    nmu = mean[r]; nrs = rstd[r];
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (on[i]) nx[i] = *reinterpret_cast<const f32x4*>(x + r * cols + (lane + 64 * i) * 4);
#pragma unroll
    for (int k = 0; k < LN_MULTI; ++k) {
      const int64_t gr = a.map[k].at(r);
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (on[i]) nd[k][i].ld(a.dy[k], gr * cols + (lane + 64 * i) * 4);
    }
  };
  if (row < rows) issue(row);
  for (; row < rows; row += stride) {
    f32x4 xh[NV], gg[NV];
    Raw4<DYD> cd[LN_MULTI][NV];
    const float mu = nmu, rs = nrs;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) xh[i][j] = (nx[i][j] - mu) * rs;
      gg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < LN_MULTI; ++k) cd[k][i] = nd[k][i];
    }
    if (row + stride < rows) issue(row + stride);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MULTI; ++k) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (on[i]) {
          const f32x4 d = cd[k][i].get();
          const f32x4 g = *reinterpret_cast<const f32x4*>(a.gamma[k] + (lane + 64 * i) * 4);   // L1-resident
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            gg[i][j] += d[j] * g[j];
            ag[k][i][j] += d[j] * xh[i][j];
            ab[k][i][j] += d[j];
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (on[i]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1 += gg[i][j]; s2 += gg[i][j] * xh[i][j]; }
      }
    }
    const float c1 = wave_sum(s1) / cols, c2 = wave_sum(s2) / cols;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (on[i]) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rs * (gg[i][j] - c1 - xh[i][j] * c2);
        *reinterpret_cast<f32x4*>(dx + row * cols + (lane + 64 * i) * 4) = o;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int pass = 0; pass < 2 * LN_MULTI; ++pass) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wave][lane * 4 + j] = (pass & 1) ? ab[pass >> 1][i][j] : ag[pass >> 1][i][j];
      __syncthreads();
      if (wave == 0) {
        const int c = (lane + 64 * i) * 4;
        if (c < cols) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) t += red[w][lane * 4 + j];
            part[((int64_t)blockIdx.x * 2 * LN_MULTI + pass) * cols + c + j] = t;
          }
        }
      }
    }
  }
}

// 3 workgroups of 4 waves per CU (the pipelined kernel holds two rows per wave: ~130 VGPRs at 768 columns)
int ln_blocks(int64_t rows) {
  static const int cap = [] { const char* e = segclip_tuning_env("SEGCLIP_LN_BLOCKS"); const int v = e ? atoi(e) : 768; return v < 1 ? 1 : v; }();
  int64_t b = cdiv(rows, WAVES);
  return (int)(b < cap ? (b < 1 ? 1 : b) : cap);
}

}  // namespace

static int ln_fwd_impl(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t rows,
                       int64_t cols, float eps, int x_dtype, int y_dtype, RowMap om, void* stream) {
  SEGCLIP_REQUIRE(cols % 4 == 0 && cols <= MAXV * 256, "layernorm: cols=%lld must be a multiple of 4 and <= %d",
                  (long long)cols, MAXV * 256);
  if (rows == 0) return 0;
  const int64_t fb = cdiv(rows, WAVES) < 4096 ? cdiv(rows, WAVES) : 4096;
#define LNF(NV) hipLaunchKernelGGL(ln_fwd_kernel<NV>, dim3((unsigned)fb), dim3(WAVES * 64), 0, (hipStream_t)stream, x, \
                                   gamma, beta, y, mean, rstd, rows, (int)cols, eps, x_dtype, y_dtype, om)
  switch ((int)cdiv(cols / 4, 64)) {
    case 1: LNF(1); break; case 2: LNF(2); break; case 3: LNF(3); break; case 4: LNF(4); break;
    case 5: case 6: LNF(6); break; default: LNF(8); break;
  }
#undef LNF
  SEGCLIP_CHECK_LAUNCH("layernorm_fwd");
  return 0;
}
extern "C" int segclip_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                     float* rstd, int64_t rows, int64_t cols, float eps, int x_dtype, int y_dtype,
                                     void* stream) {
  return ln_fwd_impl(x, gamma, beta, y, mean, rstd, rows, cols, eps, x_dtype, y_dtype, RowMap{0, 0, 0}, stream);
}
// the same, with the OUTPUT rows mapped: row r -> (r / seg_in) * seg_out + seg_off + r % seg_in of y (x, mean, rstd: plain rows)
extern "C" int segclip_layernorm_fwd_seg(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                         float* rstd, int64_t rows, int64_t cols, float eps, int x_dtype, int y_dtype,
                                         int64_t seg_in, int64_t seg_out, int64_t seg_off, void* stream) {
  SEGCLIP_REQUIRE(seg_in >= 1 && seg_out >= seg_in && seg_off >= 0 && seg_off + seg_in <= seg_out, "layernorm_fwd_seg: bad row mapping");
  return ln_fwd_impl(x, gamma, beta, y, mean, rstd, rows, cols, eps, x_dtype, y_dtype, RowMap{seg_in, seg_out, seg_off}, stream);
}

extern "C" size_t segclip_layernorm_bwd_ws_bytes(int64_t rows, int64_t cols) {
  return (size_t)ln_blocks(rows) * 3 * cols * sizeof(float);
}

static int ln_bwd_impl(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres,
                       void* dx, void* dx_bf16, float* dgamma, float* dbeta, float* dres_colsum, void* ws, int64_t rows,
                       int64_t cols, int dy_dtype, int x_dtype, int dx_dtype, RowMap gm, void* stream) {
  SEGCLIP_REQUIRE(cols % 4 == 0 && cols <= MAXV * 256, "layernorm: cols=%lld must be a multiple of 4 and <= %d",
                  (long long)cols, MAXV * 256);
  SEGCLIP_REQUIRE(ws != nullptr, "layernorm_bwd: workspace required");
  if (rows == 0) return 0;
  const int nb = ln_blocks(rows);
  const bool has_res = dres != nullptr, has_dx2 = dx_bf16 != nullptr;
  SEGCLIP_REQUIRE(x_dtype == SEGCLIP_F32 || x_dtype == SEGCLIP_BF16, "layernorm_bwd: bad x dtype");
#define LNB4(NV, DYD, XD, DXD)                                                                                         \
  do {                                                                                                                 \
    if (has_res && has_dx2) LNBK(NV, DYD, XD, DXD, true, true);                                                        \
    else if (has_res) LNBK(NV, DYD, XD, DXD, true, false);                                                             \
    else if (has_dx2) LNBK(NV, DYD, XD, DXD, false, true);                                                             \
    else LNBK(NV, DYD, XD, DXD, false, false);                                                                         \
  } while (0)
#define LNBK(NV, DYD, XD, DXD, R, D2)                                                                                   \
  hipLaunchKernelGGL((ln_bwd_kernel<NV, DYD, XD, DXD, R, D2>), dim3(nb), dim3(WAVES * 64), 0, (hipStream_t)stream, dy, x, \
                     gamma, mean, rstd, dres, dx, dx_bf16, (float*)ws, rows, (int)cols, gm)
#define LNB(NV)                                                                                                        \
  do {                                                                                                                 \
    const int key = (dy_dtype << 2) | (x_dtype << 1) | dx_dtype;                                                       \
    switch (key) {                                                                                                     \
      case 0: LNB4(NV, SEGCLIP_F32, SEGCLIP_F32, SEGCLIP_F32); break;                                                  \
      case 1: LNB4(NV, SEGCLIP_F32, SEGCLIP_F32, SEGCLIP_BF16); break;                                                 \
      case 2: LNB4(NV, SEGCLIP_F32, SEGCLIP_BF16, SEGCLIP_F32); break;                                                 \
      case 3: LNB4(NV, SEGCLIP_F32, SEGCLIP_BF16, SEGCLIP_BF16); break;                                                \
      case 4: LNB4(NV, SEGCLIP_BF16, SEGCLIP_F32, SEGCLIP_F32); break;                                                 \
      case 5: LNB4(NV, SEGCLIP_BF16, SEGCLIP_F32, SEGCLIP_BF16); break;                                                \
      case 6: LNB4(NV, SEGCLIP_BF16, SEGCLIP_BF16, SEGCLIP_F32); break;                                                \
      default: LNB4(NV, SEGCLIP_BF16, SEGCLIP_BF16, SEGCLIP_BF16); break;                                              \
    }                                                                                                                  \
  } while (0)
  switch ((int)cdiv(cols / 4, 64)) {
    case 1: LNB(1); break; case 2: LNB(2); break; case 3: LNB(3); break; case 4: LNB(4); break;
    case 5: case 6: LNB(6); break; default: LNB(8); break;
  }
#undef LNB
#undef LNB4
#undef LNBK
  SEGCLIP_CHECK_LAUNCH("layernorm_bwd");
  if (dgamma != nullptr) {   // dgamma == NULL: the caller combines the nb partial rows of ws itself (segclip_reduce_multi)
    launch_reduce_rows((const float*)ws, nb, (dres && dres_colsum ? 3 : 2) * cols, 3 * cols, dgamma, dbeta,
                       dres ? dres_colsum : nullptr, cols, (hipStream_t)stream);
    SEGCLIP_CHECK_LAUNCH("layernorm_bwd_reduce");
  }
  return 0;
}
extern "C" int segclip_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                     const float* rstd, const void* dres, void* dx, void* dx_bf16, float* dgamma,
                                     float* dbeta, float* dres_colsum, void* ws, int64_t rows, int64_t cols,
                                     int dy_dtype, int x_dtype, int dx_dtype, void* stream) {
  return ln_bwd_impl(dy, x, gamma, mean, rstd, dres, dx, dx_bf16, dgamma, dbeta, dres_colsum, ws, rows, cols, dy_dtype, x_dtype,
                     dx_dtype, RowMap{0, 0, 0}, stream);
}
// the same, with the rows of dy mapped like the output of segclip_layernorm_fwd_seg
extern "C" int segclip_layernorm_bwd_seg(const void* dy, const void* x, const float* gamma, const float* mean,
                                         const float* rstd, const void* dres, void* dx, void* dx_bf16, float* dgamma,
                                         float* dbeta, float* dres_colsum, void* ws, int64_t rows, int64_t cols,
                                         int dy_dtype, int x_dtype, int dx_dtype, int64_t seg_in, int64_t seg_out,
                                         int64_t seg_off, void* stream) {
  SEGCLIP_REQUIRE(seg_in >= 1 && seg_out >= seg_in && seg_off >= 0 && seg_off + seg_in <= seg_out, "layernorm_bwd_seg: bad row mapping");
  return ln_bwd_impl(dy, x, gamma, mean, rstd, dres, dx, dx_bf16, dgamma, dbeta, dres_colsum, ws, rows, cols, dy_dtype, x_dtype,
                     dx_dtype, RowMap{seg_in, seg_out, seg_off}, stream);
}

// ---- LN_MULTI (= 3) affine outputs of one normalisation; fp32 x, cols 768 / 1024, y / dy fp32 or bf16 (all the same);
// anything else: SEGCLIP_ERR_UNSUPPORTED, nothing launched (the caller runs the single-output functions).
// maps: n x {seg_in, seg_out, seg_off} (seg_in = 0: identity), see segclip_layernorm_fwd_seg.
static bool ln_multi_covers(int n, int64_t cols, int x_dtype, int y_dtype) {
  return n == LN_MULTI && (cols == 768 || cols == 1024) && x_dtype == SEGCLIP_F32 && (y_dtype == SEGCLIP_F32 || y_dtype == SEGCLIP_BF16);
}
static bool ln_multi_maps(const int64_t* maps, RowMap* out) {
  for (int k = 0; k < LN_MULTI; ++k) {
    out[k] = RowMap{maps[3 * k], maps[3 * k + 1], maps[3 * k + 2]};
    if (out[k].seg_in && !(out[k].seg_in >= 1 && out[k].off >= 0 && out[k].off + out[k].seg_in <= out[k].seg_out)) return false;
  }
  return true;
}
extern "C" int segclip_layernorm_fwd_multi(const void* x, int n, const float* const* gamma, const float* const* beta,
                                           void* const* y, const int64_t* maps, float* mean, float* rstd, int64_t rows,
                                           int64_t cols, float eps, int x_dtype, int y_dtype, void* stream) {
  if (!ln_multi_covers(n, cols, x_dtype, y_dtype)) return SEGCLIP_ERR_UNSUPPORTED;
  LnMultiFwd a;
  SEGCLIP_REQUIRE(ln_multi_maps(maps, a.map), "layernorm_fwd_multi: bad row mapping");
  for (int k = 0; k < LN_MULTI; ++k) { a.gamma[k] = gamma[k]; a.beta[k] = beta[k]; a.y[k] = y[k]; }
  if (rows == 0) return 0;
  const int64_t fb = cdiv(rows, WAVES) < 4096 ? cdiv(rows, WAVES) : 4096;
#define LNM(NV, YD) hipLaunchKernelGGL((ln_fwd_multi_kernel<NV, YD>), dim3((unsigned)fb), dim3(WAVES * 64), 0, (hipStream_t)stream, \
                                       (const float*)x, a, mean, rstd, rows, (int)cols, eps)
  if (cols == 768) { if (y_dtype == SEGCLIP_BF16) LNM(3, SEGCLIP_BF16); else LNM(3, SEGCLIP_F32); }
  else { if (y_dtype == SEGCLIP_BF16) LNM(4, SEGCLIP_BF16); else LNM(4, SEGCLIP_F32); }
#undef LNM
  SEGCLIP_CHECK_LAUNCH("layernorm_fwd_multi");
  return 0;
}
// the multi backward holds ~220 VGPRs (two rows of x + 3 dy in flight, 6 column accumulators): 2 workgroups per CU
static int ln_multi_blocks(int64_t rows) {
  const int64_t b = cdiv(rows, WAVES);
  return (int)(b < 512 ? (b < 1 ? 1 : b) : 512);
}
extern "C" size_t segclip_layernorm_bwd_multi_ws_bytes(int64_t rows, int64_t cols, int n) {
  return (size_t)ln_multi_blocks(rows) * 2 * n * cols * sizeof(float);
}
// dgb: (2n, cols) fp32 = [dgamma_0, dbeta_0, dgamma_1, dbeta_1, ...]
extern "C" int segclip_layernorm_bwd_multi(const void* const* dy, const void* x, int n, const float* const* gamma,
                                           const int64_t* maps, const float* mean, const float* rstd, void* dx, float* dgb,
                                           void* ws, int64_t rows, int64_t cols, int dy_dtype, int x_dtype, void* stream) {
  if (!ln_multi_covers(n, cols, x_dtype, dy_dtype)) return SEGCLIP_ERR_UNSUPPORTED;
  SEGCLIP_REQUIRE(ws != nullptr && dgb != nullptr, "layernorm_bwd_multi: workspace and dgb required");
  LnMultiBwd a;
  SEGCLIP_REQUIRE(ln_multi_maps(maps, a.map), "layernorm_bwd_multi: bad row mapping");
  for (int k = 0; k < LN_MULTI; ++k) { a.gamma[k] = gamma[k]; a.dy[k] = dy[k]; }
  if (rows == 0) {
    const hipError_t e = hipMemsetAsync(dgb, 0, (size_t)2 * n * cols * sizeof(float), (hipStream_t)stream);
    SEGCLIP_REQUIRE(e == hipSuccess, "layernorm_bwd_multi: memset failed: %s", hipGetErrorString(e));
    return 0;
  }
  const int nb = ln_multi_blocks(rows);
#define LNM(NV, DYD) hipLaunchKernelGGL((ln_bwd_multi_kernel<NV, DYD>), dim3(nb), dim3(WAVES * 64), 0, (hipStream_t)stream, \
                                        (const float*)x, a, mean, rstd, (float*)dx, (float*)ws, rows, (int)cols)
  if (cols == 768) { if (dy_dtype == SEGCLIP_BF16) LNM(3, SEGCLIP_BF16); else LNM(3, SEGCLIP_F32); }
  else { if (dy_dtype == SEGCLIP_BF16) LNM(4, SEGCLIP_BF16); else LNM(4, SEGCLIP_F32); }
#undef LNM
  SEGCLIP_CHECK_LAUNCH("layernorm_bwd_multi");
  const int64_t w = (int64_t)2 * n * cols;
  launch_reduce_rows((const float*)ws, nb, w, w, dgb, nullptr, nullptr, w, (hipStream_t)stream);
  SEGCLIP_CHECK_LAUNCH("layernorm_bwd_multi_reduce");
  return 0;
}
