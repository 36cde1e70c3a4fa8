// HBM-bound / latency-bound helpers of the SegCLIP hot path: casts, bias-gradient column sums,
// activations, vision/text front ends, row gathers, learnable-center hard assignment, losses and the
// MAE masking sort.  All integer outputs (argmax, ranks) are computed in fp32/integer arithmetic only.
#include "common.h"
#include "reduce_rows.h"

namespace {

constexpr int TPB = 256;
inline int grid1d(int64_t n, int per_thread = 1) {
  int64_t b = cdiv(n, (int64_t)TPB * per_thread);
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
__device__ __forceinline__ float ldx(const void* p, int dtype, int64_t i) {
  return dtype == SEGCLIP_F32 ? reinterpret_cast<const float*>(p)[i] : bf2f(reinterpret_cast<const bf16_t*>(p)[i]);
}
__device__ __forceinline__ void stx(void* p, int dtype, int64_t i, float v) {
  if (dtype == SEGCLIP_F32) reinterpret_cast<float*>(p)[i] = v; else reinterpret_cast<bf16_t*>(p)[i] = f2bf(v);
}
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < nw; ++w) t += red[w];
  return t;
}

// ---------------------------------------------------------------- cast / add / act
__global__ void cast_kernel(const void* __restrict__ src, void* __restrict__ dst, int64_t n, int sd, int dd) {
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 v;
    if (sd == SEGCLIP_F32) v = reinterpret_cast<const f32x4*>(src)[i];
    else {
      const u32x2 t = reinterpret_cast<const u32x2*>(src)[i];
      v[0] = __uint_as_float(t[0] << 16); v[1] = __uint_as_float(t[0] & 0xffff0000u);
      v[2] = __uint_as_float(t[1] << 16); v[3] = __uint_as_float(t[1] & 0xffff0000u);
    }
    if (dd == SEGCLIP_F32) reinterpret_cast<f32x4*>(dst)[i] = v;
    else { u32x2 t; t[0] = pack2bf(v[0], v[1]); t[1] = pack2bf(v[2], v[3]); reinterpret_cast<u32x2*>(dst)[i] = t; }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    stx(dst, dd, i, ldx(src, sd, i));
  }
}
__global__ void add_kernel(const void* a, const void* b, void* out, int64_t n, int dt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    stx(out, dt, i, ldx(a, dt, i) + ldx(b, dt, i));
}
__global__ void act_fwd_kernel(const void* x, void* y, int64_t n, int act, int dt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    stx(y, dt, i, apply_act(act, ldx(x, dt, i)));
}
__device__ __forceinline__ f32x4 ld4x_any(const void* X, int dt, int64_t idx) {
  if (dt == SEGCLIP_F32) return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(X) + idx);
  const u32x2 t = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(X) + idx);
  return f32x4{__uint_as_float(t[0] << 16), __uint_as_float(t[0] & 0xffff0000u), __uint_as_float(t[1] << 16),
               __uint_as_float(t[1] & 0xffff0000u)};
}
__global__ void act_bwd_kernel(const void* dy, const void* x, void* dx, int64_t n, int act, int dt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    stx(dx, dt, i, ldx(dy, dt, i) * apply_act_grad(act, ldx(x, dt, i)));
}
// 4 elements per thread, 16-byte (fp32) / 8-byte (bf16) accesses; needs n % 4 == 0 and 16-byte aligned pointers
__global__ void act_bwd_vec4_kernel(const void* __restrict__ dy, const void* __restrict__ x, void* __restrict__ dx, int64_t n4,
                                    int act, int dt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 g = ld4x_any(dy, dt, i * 4), v = ld4x_any(x, dt, i * 4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = g[j] * apply_act_grad(act, v[j]);
    if (dt == SEGCLIP_F32) {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(dx) + i * 4) = o;
    } else {
      u32x2 t;
      t[0] = pack2bf(o[0], o[1]); t[1] = pack2bf(o[2], o[3]);
      *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(dx) + i * 4) = t;
    }
  }
}
__global__ void scale_kernel(const float* x, const float* s, float* out, int64_t n) {
  const float sc = *s;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = x[i] * sc;
}
// Gumbel(0,1) from uniform samples: g = -log(-log(clamp(u, tiny, 1 - eps)))  (torch.distributions.Gumbel.sample as
// config.gumbel wrote it op by op: clamp, log, neg, log, neg); logf is the same library routine ATen's log kernel calls
__global__ void gumbel_from_uniform_kernel(const float* __restrict__ u, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = u[i];
    v = v < 1.17549435e-38f ? 1.17549435e-38f : (v > 1.0f - 1.1920929e-07f ? 1.0f - 1.1920929e-07f : v);
    out[i] = -logf(-logf(v));
  }
}
// out[0] = scale * sum(x[0..n))   (single block, deterministic)
__global__ void reduce_sum_kernel(const float* x, float* out, int64_t n, float scale) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s * scale;
}

// ---------------------------------------------------------------- column sums (bias gradients)
// stage 1: block = 64 column-quads x 4 row lanes over one row chunk -> part[chunk][N]; stage 2 sums <= CS_MAXCHUNK chunks.
constexpr int CS_MAXCHUNK = 256;
__device__ __forceinline__ f32x4 ld4x(const void* X, int dt, int64_t idx, bool vec, int nvalid) {
  f32x4 r = {0.f, 0.f, 0.f, 0.f};
  if (vec) {
    if (dt == SEGCLIP_F32) r = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(X) + idx);
    else {
      const u32x2 t = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(X) + idx);
      r[0] = __uint_as_float(t[0] << 16); r[1] = __uint_as_float(t[0] & 0xffff0000u);
      r[2] = __uint_as_float(t[1] << 16); r[3] = __uint_as_float(t[1] & 0xffff0000u);
    }
  } else {
    for (int j = 0; j < nvalid; ++j) r[j] = ldx(X, dt, idx + j);
  }
  return r;
}
__global__ __launch_bounds__(256) void colsum_partial_kernel(const void* __restrict__ X, float* __restrict__ part,
                                                             int64_t M, int64_t N, int64_t ld, int dt, int64_t rows_per) {
  __shared__ f32x4 red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int64_t c = ((int64_t)blockIdx.x * 64 + cl) * 4;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = r0 + rows_per < M ? r0 + rows_per : M;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c < N) {
    const int nvalid = (int)(N - c < 4 ? N - c : 4);
    const int esz = dt == SEGCLIP_F32 ? 4 : 2;
    const bool vec = nvalid == 4 && (ld * esz) % (4 * esz) == 0 &&
                     ((reinterpret_cast<uintptr_t>(X) + c * esz) % (4 * esz)) == 0;
    for (int64_t r = r0 + rl; r < r1; r += 4) s += ld4x(X, dt, r * ld + c, vec, nvalid);
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < N) {
    const f32x4 t = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    for (int j = 0; j < 4 && c + j < N; ++j) part[(int64_t)blockIdx.y * N + c + j] = t[j];
  }
}
__global__ void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t nchunk, int64_t N) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  float s = 0.f;
  for (int64_t k = 0; k < nchunk; ++k) s += part[k * N + c];
  out[c] = s;
}
// 256-row chunks for tall matrices; a short one (M = B*8 center rows) would leave each thread a 64-deep dependent load
// chain on a handful of workgroups (22 us for 2048 x 3072 bf16), so chunks shrink to 32 rows until ~1024 workgroups exist
static inline int64_t colsum_chunks(int64_t M, int64_t N) {
  int64_t c = cdiv(M, 256);
  const int64_t colblocks = cdiv(N > 0 ? N : 1, 256);
  const int64_t want = cdiv(1024, colblocks), most = cdiv(M, 32);
  if (c < want) c = want < most ? want : most;
  return c < 1 ? 1 : (c > CS_MAXCHUNK ? CS_MAXCHUNK : c);
}

// ---------------------------------------------------------------- positional-table interpolation (eval tier)
// F.interpolate(mode='bicubic', align_corners=False) on a channel-last (n x n x D) table -> (h x w x D):
// source coordinate (dst + 0.5) * in/out - 0.5 (not clamped), cubic-convolution weights with A = -0.75, tap indices
// clamped to the border -- torch's upsample_bicubic2d.
__device__ __forceinline__ void cubic_weights(float t, float (&w)[4]) {
  const float A = -0.75f;
  const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = 2.0f - t;
  w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
  w[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
  w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
  w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}
__global__ void interp_bicubic_kernel(const float* __restrict__ src, float* __restrict__ dst, int n_in, int h, int w, int D) {
  const int64_t total = (int64_t)h * w * D;
  const float sy = (float)n_in / (float)h, sx = (float)n_in / (float)w;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const int ox = (int)((i / D) % w), oy = (int)(i / ((int64_t)D * w));
    const float fy = sy * (oy + 0.5f) - 0.5f, fx = sx * (ox + 0.5f) - 0.5f;
    const float iyf = floorf(fy), ixf = floorf(fx);
    float wy[4], wx[4];
    cubic_weights(fy - iyf, wy);
    cubic_weights(fx - ixf, wx);
    const int iy = (int)iyf, ix = (int)ixf;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int yy = min(max(iy - 1 + a, 0), n_in - 1);
      float row = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int xx = min(max(ix - 1 + b, 0), n_in - 1);
        row += wx[b] * src[((int64_t)yy * n_in + xx) * D + d];
      }
      acc += wy[a] * row;
    }
    dst[i] = acc;
  }
}

// ---------------------------------------------------------------- vision front end
// layout 0: col = c*p*p + py*p + px (conv1 weight order) ; layout 1: col = (py*p + px)*C + c (MAE patchify)
// rows of `ld` >= C*p*p elements; columns beyond C*p*p are zero (K padded to the GEMM's alignment, e.g. 3*14*14 = 588 -> 640)
__global__ void im2col_kernel(const float* __restrict__ img, void* __restrict__ cols, int64_t B, int C, int H, int W,
                              int p, int layout, int od, int64_t ld) {
  const int gw = W / p, gh = H / p;
  const int64_t kdim = (int64_t)C * p * p, total = B * gh * gw * ld;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / ld;
    const int col = (int)(i % ld);
    if (col >= kdim) { stx(cols, od, i, 0.f); continue; }
    int c, py, px;
    if (layout == 0) { c = col / (p * p); py = (col / p) % p; px = col % p; }
    else { c = col % C; py = (col / C) / p; px = (col / C) % p; }
    const int64_t b = row / (gh * gw);
    const int gy = (int)((row / gw) % gh), gx = (int)(row % gw);
    stx(cols, od, i, img[((b * C + c) * H + gy * p + py) * W + gx * p + px]);
  }
}
// layout 0 with p % 8 == 0 and ld % 8 == 0: one thread per 8 consecutive columns = 8 consecutive pixels of one patch row
// (two 16-byte loads, one 16- or 32-byte store; one 64-bit division per 8 elements instead of five per element:
// 256 x 3 x 224 x 224 -> bf16 209 -> ~50 us)
__global__ void im2col_vec8_kernel(const float* __restrict__ img, void* __restrict__ cols, int64_t B, int C, int H, int W,
                                   int p, int od, int64_t ld) {
  const int gw = W / p, gh = H / p, ld8 = (int)(ld / 8), pp = p * p;
  const int kdim = C * pp;
  const int64_t total = B * gh * gw * ld8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / ld8;
    const int col = (int)(i - row * ld8) * 8;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b4 = a;
    if (col < kdim) {
      const int c = col / pp, rem = col - c * pp, py = rem / p, px = rem - py * p;
      const int r32 = (int)(row % (gh * gw));
      const int64_t b = row / (gh * gw);
      const int gy = r32 / gw, gx = r32 - gy * gw;
      const float* src = img + ((b * C + c) * H + gy * p + py) * W + gx * p + px;
      a = *reinterpret_cast<const f32x4*>(src);
      b4 = *reinterpret_cast<const f32x4*>(src + 4);
    }
    if (od == SEGCLIP_BF16) {
      u32x4 o = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b4[0], b4[1]), pack2bf(b4[2], b4[3])};
      reinterpret_cast<u32x4*>(cols)[i] = o;
    } else {
      reinterpret_cast<f32x4*>(cols)[2 * i] = a;
      reinterpret_cast<f32x4*>(cols)[2 * i + 1] = b4;
    }
  }
}
__global__ void vis_assemble_kernel(const void* __restrict__ patches, const float* __restrict__ cls,
                                    const float* __restrict__ pos, float* __restrict__ x, int64_t B, int T, int D, int pd) {
  const int64_t total = B * (T + 1) * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const int t = (int)((i / D) % (T + 1));
    const int64_t b = i / ((int64_t)D * (T + 1));
    const float v = t == 0 ? cls[d] : ldx(patches, pd, (b * T + (t - 1)) * D + d);
    x[i] = v + pos[(int64_t)t * D + d];
  }
}

// ---------------------------------------------------------------- text front end
__global__ void embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                 const float* __restrict__ pos, float* __restrict__ out, int64_t BL, int L, int D,
                                 int64_t vocab) {
  const int d4 = D / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < BL * d4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / d4;
    const int c = (int)(i % d4);
    int64_t id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const f32x4 e = reinterpret_cast<const f32x4*>(table + id * D)[c];
    const f32x4 p = reinterpret_cast<const f32x4*>(pos + (row % L) * D)[c];
    reinterpret_cast<f32x4*>(out + row * D)[c] = e + p;
  }
}
__global__ void embed_bwd_table_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dout,
                                       float* __restrict__ dtable, int64_t BL, int D, int64_t vocab) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < BL * D; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / D;
    int64_t id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    // positions behind the EOT token receive an exactly zero gradient (causal tower, EOT pooling) and all share the pad id:
    // adding +0 changes nothing, skipping it removes 3/4 of the atomics and the contention on the pad row (377 -> ~100 us)
    const float v = dout[i];
    if (v != 0.f) atomicAdd(dtable + id * D + (i % D), v);
  }
}
// dpos[l][d] = sum_b dout[b][l][d], sequential over b (deterministic); one thread per (l, 4 columns), 4 samples in flight
__global__ void embed_bwd_pos_kernel(const float* __restrict__ dout, float* __restrict__ dpos, int64_t B, int L, int D) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;  // over L*D, D % 4 == 0
  const int64_t LD = (int64_t)L * D;
  if (i >= LD) return;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int64_t b = 0;
  for (; b + 4 <= B; b += 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(dout + (b + u) * LD + i);
#pragma unroll
    for (int u = 0; u < 4; ++u) s += v[u];
  }
  for (; b < B; ++b) s += *reinterpret_cast<const f32x4*>(dout + b * LD + i);
  *reinterpret_cast<f32x4*>(dpos + i) = s;
}

// ---------------------------------------------------------------- row gather / scatter
__global__ void gather_rows_kernel(const void* __restrict__ src, const int64_t* __restrict__ idx, void* __restrict__ out,
                                   int64_t B, int Ts, int To, int D, int dt, int scatter) {
  const int64_t total = B * To * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const int64_t bj = i / D, b = bj / To;
    int64_t t = idx[bj];
    t = t < 0 ? 0 : (t >= Ts ? Ts - 1 : t);
    const int64_t s = (b * Ts + t) * D + d;
    if (!scatter) stx(out, dt, i, ldx(src, dt, s));
    else stx(out, dt, s, ldx(out, dt, s) + ldx(src, dt, i));  // unique indices per b: no race
  }
}

// ---------------------------------------------------------------- MAE glue (reference modules/modeling.py:240-242, modules/module_mae.py:310-314)
// mean-CLS concat: out[b][0][:] = mean_t x[b][t][:], out[b][1+t][:] = x[b][t][:].  One wave per (sample, 256-column chunk):
// the token loop is the reduction, so the sum has one fixed order.
__global__ __launch_bounds__(64) void mean_cat_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int D4, int chunks) {
  const int b = blockIdx.x / chunks, c = (blockIdx.x % chunks) * 64 + threadIdx.x;
  if (c >= D4) return;
  const f32x4* xs = reinterpret_cast<const f32x4*>(x) + (int64_t)b * T * D4 + c;
  f32x4* os = reinterpret_cast<f32x4*>(out) + (int64_t)b * (T + 1) * D4 + c;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < T; ++t) {
    const f32x4 v = xs[(int64_t)t * D4];
    acc += v;
    os[(int64_t)(t + 1) * D4] = v;
  }
  const float inv = 1.0f / (float)T;
  os[0] = acc * f32x4{inv, inv, inv, inv};
}
// dx[b][t][:] = dout[b][1+t][:] + dout[b][0][:] / T
__global__ void mean_cat_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, int64_t B, int T, int D4) {
  const int64_t total = B * T * D4;
  const float inv = 1.0f / (float)T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % D4);
    const int64_t bt = i / D4, b = bt / T, t = bt % T;
    const f32x4* dr = reinterpret_cast<const f32x4*>(dout) + (b * (T + 1)) * D4 + c;
    reinterpret_cast<f32x4*>(dx)[i] = dr[(t + 1) * (int64_t)D4] + dr[0] * f32x4{inv, inv, inv, inv};
  }
}
// decoder input: out[b][j][:] = (ids[b][j] < K ? x[b][ids[b][j]][:] : mask_token[:]) + pos[j][:]
//   = gather(cat([x, mask_token.expand(B, L - K, D)], 1), ids) + pos  without the concatenated tensor
__global__ void mae_unshuffle_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mask_token, const int64_t* __restrict__ ids,
                                         const float* __restrict__ pos, float* __restrict__ out, int64_t B, int K, int L, int D4) {
  const int64_t total = B * L * D4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % D4);
    const int64_t bj = i / D4, b = bj / L, j = bj % L;
    const int64_t id = ids[bj];
    const f32x4 v = id < K ? reinterpret_cast<const f32x4*>(x)[(b * K + (id < 0 ? 0 : id)) * D4 + c] : reinterpret_cast<const f32x4*>(mask_token)[c];
    reinterpret_cast<f32x4*>(out)[i] = v + reinterpret_cast<const f32x4*>(pos)[j * D4 + c];
  }
}
// backward: one wave per (position j, 256-column chunk) walks the samples: dx[b][ids[b][j]][:] = dout[b][j][:] (kept tokens: ids is
// a permutation per sample, every kept row is written exactly once), dpos[j][:] = sum_b dout[b][j][:], mpart[j][:] = the part of
// that sum that belongs to mask tokens (the caller sums mpart over j for the mask token's gradient).  Fixed summation order.
__global__ __launch_bounds__(64) void mae_unshuffle_bwd_kernel(const float* __restrict__ dout, const int64_t* __restrict__ ids, float* __restrict__ dx,
                                                               float* __restrict__ dpos, float* __restrict__ mpart, int64_t B, int K, int L, int D4,
                                                               int chunks) {
  const int j = blockIdx.x / chunks, c = (blockIdx.x % chunks) * 64 + threadIdx.x;
  if (c >= D4) return;
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, m = {0.f, 0.f, 0.f, 0.f};
  int64_t b = 0;
  for (; b + 8 <= B; b += 8) {          // eight samples' loads in flight (the walk is latency-bound); the sums keep sample order
    f32x4 g[8];
    int64_t id[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      g[u] = reinterpret_cast<const f32x4*>(dout)[((b + u) * L + j) * D4 + c];
      id[u] = ids[(b + u) * L + j];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s += g[u];
      if (id[u] < K) reinterpret_cast<f32x4*>(dx)[((b + u) * K + (id[u] < 0 ? 0 : id[u])) * D4 + c] = g[u];
      else m += g[u];
    }
  }
  for (; b < B; ++b) {
    const f32x4 g = reinterpret_cast<const f32x4*>(dout)[(b * L + j) * D4 + c];
    const int64_t id = ids[b * L + j];
    s += g;
    if (id < K) reinterpret_cast<f32x4*>(dx)[(b * K + (id < 0 ? 0 : id)) * D4 + c] = g;
    else m += g;
  }
  reinterpret_cast<f32x4*>(dpos)[(int64_t)j * D4 + c] = s;
  reinterpret_cast<f32x4*>(mpart)[(int64_t)j * D4 + c] = m;
}

// ---------------------------------------------------------------- learnable-center assignment
// logits (B,G,T).  One thread per (b,t).  counts accumulated with exact integer-valued float atomics.
__global__ void assign_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ gumbel, float tau,
                                  float* __restrict__ y_soft, float* __restrict__ soft, uint8_t* __restrict__ idx,
                                  float* __restrict__ hard, float* __restrict__ counts, int64_t B, int G, int T) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T) return;
  const int64_t b = i / T;
  const int t = (int)(i % T);
  const float* lp = logits + b * G * T + t;
  const float* gp = gumbel ? gumbel + b * G * T + t : nullptr;
  float m1 = -INFINITY, m2 = -INFINITY;
  for (int g = 0; g < G; ++g) {
    const float l = lp[(int64_t)g * T];
    const float z = gp ? (l + gp[(int64_t)g * T]) / tau : l;
    m1 = fmaxf(m1, z); m2 = fmaxf(m2, l);
  }
  float s1 = 0.f, s2 = 0.f;
  for (int g = 0; g < G; ++g) {
    const float l = lp[(int64_t)g * T];
    const float z = gp ? (l + gp[(int64_t)g * T]) / tau : l;
    s1 += expf(z - m1); s2 += expf(l - m2);
  }
  int best = 0; float bestv = -INFINITY;
  for (int g = 0; g < G; ++g) {
    const float l = lp[(int64_t)g * T];
    const float z = gp ? (l + gp[(int64_t)g * T]) / tau : l;
    const float y = expf(z - m1) / s1;
    y_soft[b * G * T + (int64_t)g * T + t] = y;
    if (soft) soft[b * G * T + (int64_t)g * T + t] = expf(l - m2) / s2;
    if (y > bestv) { bestv = y; best = g; }  // first maximum wins (torch.max)
  }
  idx[i] = (uint8_t)best;
  for (int g = 0; g < G; ++g) hard[b * G * T + (int64_t)g * T + t] = g == best ? 1.f : 0.f;
  atomicAdd(counts + b * G + best, 1.0f);
}
// dlogits = softmax-bwd through y_soft/tau of the straight-through gradient dhard
__global__ void assign_bwd_kernel(const float* __restrict__ dhard, const float* __restrict__ y_soft, float tau,
                                  float* __restrict__ dlogits, int64_t B, int G, int T) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T) return;
  const int64_t b = i / T;
  const int t = (int)(i % T);
  const int64_t base = b * G * T + t;
  float dot = 0.f;
  for (int g = 0; g < G; ++g) dot += dhard[base + (int64_t)g * T] * y_soft[base + (int64_t)g * T];
  for (int g = 0; g < G; ++g) {
    const float y = y_soft[base + (int64_t)g * T];
    dlogits[base + (int64_t)g * T] = y * (dhard[base + (int64_t)g * T] - dot) / tau;
  }
}

// ---------------------------------------------------------------- losses
__global__ void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ norm,
                                  int64_t rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) { const float v = x[row * cols + c]; s += v * v; }
  const float n = sqrtf(wave_sum(s));
  if (lane == 0) norm[row] = n;
  for (int c = lane; c < cols; c += 64) y[row * cols + c] = x[row * cols + c] / n;
}
__global__ void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ norm,
                                  float* __restrict__ dx, int64_t rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += dy[row * cols + c] * y[row * cols + c];
  s = wave_sum(s);
  const float inv = 1.f / norm[row];
  for (int c = lane; c < cols; c += 64) dx[row * cols + c] = (dy[row * cols + c] - y[row * cols + c] * s) * inv;
}
// one wave per row: lse and -log softmax[label]
__global__ void ce_fwd_kernel(const float* __restrict__ logits, float* __restrict__ lse, float* __restrict__ loss_rows,
                              int64_t rows, int64_t cols, int64_t label_offset) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = logits + row * cols;
  float mx = -INFINITY;
  for (int64_t c = lane; c < cols; c += 64) mx = fmaxf(mx, p[c]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int64_t c = lane; c < cols; c += 64) s += expf(p[c] - mx);
  s = wave_sum(s);
  const float l = mx + logf(s);
  if (lane == 0) { lse[row] = l; loss_rows[row] = l - p[row + label_offset]; }
}
__global__ void ce_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ lse, const float* gscale_ptr,
                              float gscale, float* __restrict__ dlogits, int64_t rows, int64_t cols, int64_t label_offset) {
  const float gs = (gscale_ptr ? *gscale_ptr : 1.f) * gscale / (float)rows;
  const int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / cols, c = i % cols;
    const float p = expf(logits[i] - lse[row]);
    dlogits[i] = gs * (p - (c == row + label_offset ? 1.f : 0.f));
  }
}

// cross entropy with explicit labels and an ignore index; one wave per row (the vocabulary: 49408 columns)
__global__ void ce_labels_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, int64_t ignore,
                                     float* __restrict__ lse, float* __restrict__ loss_rows, float* __restrict__ valid,
                                     int64_t rows, int64_t cols) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = logits + row * cols;
  float mx = -INFINITY;
  for (int64_t c = lane; c < cols; c += 64) mx = fmaxf(mx, p[c]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int64_t c = lane; c < cols; c += 64) s += expf(p[c] - mx);
  s = wave_sum(s);
  const float l = mx + logf(s);
  if (lane == 0) {
    const int64_t lab = labels[row];
    const bool ok = lab != ignore && lab >= 0 && lab < cols;
    lse[row] = l;
    loss_rows[row] = ok ? l - p[lab] : 0.f;
    valid[row] = ok ? 1.f : 0.f;
  }
}
__global__ void ce_labels_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ lse,
                                     const int64_t* __restrict__ labels, int64_t ignore, const float* __restrict__ gscale,
                                     const float* __restrict__ inv_count, float* __restrict__ dlogits, int64_t rows,
                                     int64_t cols) {
  const float gs = gscale[0] * inv_count[0];
  const int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / cols, c = i % cols;
    const int64_t lab = labels[row];
    const bool ok = lab != ignore && lab >= 0 && lab < cols;
    dlogits[i] = ok ? gs * (expf(logits[i] - lse[row]) - (c == lab ? 1.f : 0.f)) : 0.f;
  }
}

// superpixel-KL: one block per image.  hard (B,G,T) one-hot, seg (B,T) int64.  Emits loss_rows[b] (already
// divided by coef = B*T*G and by 2) and the un-scaled gradient dhard (B,G,T) of that per-image loss.
constexpr int KL_MAXG = 16;
__global__ void superpixel_kl_kernel(const float* __restrict__ hard, const int64_t* __restrict__ seg,
                                     float* __restrict__ loss_rows, float* __restrict__ dhard, int64_t B, int G, int T) {
  extern __shared__ __attribute__((aligned(16))) char kl_smem[];
  float* hm = reinterpret_cast<float*>(kl_smem);            // [T][G] dJ/dm per patch
  int* lab = reinterpret_cast<int*>(hm + (size_t)T * G);    // [T] representative index of the label
  __shared__ float red[16];
  const int64_t b = blockIdx.x;
  const float* h = hard + b * G * T;
  const int64_t* sg = seg + b * T;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const int64_t me = sg[t];
    int rep = t;
    for (int l = 0; l < t; ++l) if (sg[l] == me) { rep = l; break; }
    lab[t] = rep;
  }
  __syncthreads();
  const float coef = (float)(B * T * G);
  float lsum = 0.f;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    float a[KL_MAXG], m[KL_MAXG];
    int n = 0;
    for (int c = 0; c < G; ++c) m[c] = 0.f;
    for (int l = 0; l < T; ++l)
      if (lab[l] == lab[t]) { ++n; for (int c = 0; c < G; ++c) m[c] += h[(int64_t)c * T + l]; }
    const float invn = 1.f / fmaxf((float)n, 1.f);
    float ma = -INFINITY, mm = -INFINITY;
    for (int c = 0; c < G; ++c) { a[c] = h[(int64_t)c * T + t]; m[c] *= invn; ma = fmaxf(ma, a[c]); mm = fmaxf(mm, m[c]); }
    float sa = 0.f, sm = 0.f;
    for (int c = 0; c < G; ++c) { sa += expf(a[c] - ma); sm += expf(m[c] - mm); }
    const float la = ma + logf(sa), lm = mm + logf(sm);
    float J = 0.f, sG = 0.f, sH = 0.f, Gc[KL_MAXG], Hc[KL_MAXG];
    for (int c = 0; c < G; ++c) {
      const float u = a[c] - la, v = m[c] - lm, q = expf(u), p = expf(v);
      J += (p - q) * (v - u);
      Gc[c] = -p + q - q * (v - u);
      Hc[c] = p - q + p * (v - u);
      sG += Gc[c]; sH += Hc[c];
    }
    lsum += J;
    for (int c = 0; c < G; ++c) {
      const float u = a[c] - la, v = m[c] - lm;
      // direct path through log_softmax(h)/softmax(h)
      dhard[b * G * T + (int64_t)c * T + t] = (Gc[c] - expf(u) * sG) * (0.5f / coef);
      hm[(size_t)t * G + c] = (Hc[c] - expf(v) * sH) * invn * (0.5f / coef);
    }
  }
  __syncthreads();
  // path through the superpixel mean: dh[l,:] += sum_{g in S(l)} dJ/dm[g,:] / n_g
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    float acc[KL_MAXG];
    for (int c = 0; c < G; ++c) acc[c] = 0.f;
    for (int g = 0; g < T; ++g)
      if (lab[g] == lab[t]) for (int c = 0; c < G; ++c) acc[c] += hm[(size_t)g * G + c];
    for (int c = 0; c < G; ++c) dhard[b * G * T + (int64_t)c * T + t] += acc[c];
  }
  lsum = block_sum(lsum, red);
  if (threadIdx.x == 0) loss_rows[b] = lsum * 0.5f / coef;
}

// MAE masked MSE: per (b,t) row: mask * mean_d (pred - target)^2 ; one wave per row
__global__ void masked_mse_fwd_kernel(const void* __restrict__ pred, const float* __restrict__ target,
                                      const float* __restrict__ mask, float* __restrict__ loss_rows, int64_t B, int T,
                                      int Dp, int pd) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= B * T) return;
  const int64_t b = row / T, t = row % T;
  const float mk = mask[b * (T + 1) + 1 + t];
  float s = 0.f;
  if (mk != 0.f)
    for (int d = lane; d < Dp; d += 64) {
      const float e = ldx(pred, pd, (b * (T + 1) + 1 + t) * Dp + d) - target[row * Dp + d];
      s += e * e;
    }
  s = wave_sum(s);
  if (lane == 0) loss_rows[row] = mk * s / Dp;
}
// dpred (B,1+T,Dp): row 0 zero; gs = gscale * (*gptr) / sum(mask)
__global__ void masked_mse_bwd_kernel(const void* __restrict__ pred, const float* __restrict__ target,
                                      const float* __restrict__ mask, const float* gptr, const float* msum, float gscale,
                                      void* __restrict__ dpred, int64_t B, int T, int Dp, int pd) {
  const float gs = (gptr ? *gptr : 1.f) * gscale / (*msum) * 2.f / Dp;
  const int64_t total = B * (T + 1) * Dp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % Dp);
    const int64_t bt = i / Dp, b = bt / (T + 1), t1 = bt % (T + 1);
    float v = 0.f;
    if (t1 > 0) {
      const float mk = mask[bt];
      if (mk != 0.f) v = gs * mk * (ldx(pred, pd, i) - target[(b * T + t1 - 1) * Dp + d]);
    }
    stx(dpred, pd, i, v);
  }
}

// MAE masking: stable rank sort per row (L <= 4096), one block per row
__global__ void mask_sort_kernel(const float* __restrict__ noise, int64_t* __restrict__ ids_shuffle,
                                 int64_t* __restrict__ ids_restore, float* __restrict__ mask, int L, int len_keep) {
  extern __shared__ __attribute__((aligned(16))) char ms_smem[];
  float* nz = reinterpret_cast<float*>(ms_smem);
  const int64_t b = blockIdx.x;
  for (int i = threadIdx.x; i < L; i += blockDim.x) nz[i] = i == 0 ? -1.0f : noise[b * L + i];
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const float v = nz[i];
    int rank = 0;
    for (int j = 0; j < L; ++j) { const float w = nz[j]; rank += (w < v) || (w == v && j < i); }
    ids_restore[b * L + i] = rank;
    ids_shuffle[b * L + rank] = i;
    mask[b * L + i] = rank >= len_keep ? 1.f : 0.f;
  }
}

// fp32 -> two bf16 parts (hi = rne(x), lo = rne(x - hi): 16 mantissa bits together), written three times along the CONTRACTION
// dimension of a GEMM operand: role 0 (A) = hi | lo | hi, role 1 (B) = hi | hi | lo, so that ONE bf16 GEMM of contraction length
// 3 K sums A_hi B_hi + A_lo B_hi + A_hi B_lo in its fp32 accumulators (config.f32_split: the parity mode on the bf16 matrix pipe).
// stack = 0: the contraction runs along the row (dst row pitch 3 * cols, blocks side by side); stack = 1: along the rows (dst =
// three (rows x cols) blocks one after the other).  4 elements per thread.
__global__ void split3_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t rows, int64_t cols, int64_t ld,
                              int stack, int role) {
  const int64_t c4 = cols >> 2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * c4) return;
  const int64_t r = i / c4, c = (i - r * c4) << 2;
  const f32x4 x = *reinterpret_cast<const f32x4*>(src + r * ld + c);
  u32x2 hi, lo;
  float h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { h[j] = bf2f(f2bf(x[j])); l[j] = x[j] - h[j]; }
  hi[0] = pack2bf(h[0], h[1]); hi[1] = pack2bf(h[2], h[3]);
  lo[0] = pack2bf(l[0], l[1]); lo[1] = pack2bf(l[2], l[3]);
  const int64_t pitch = stack ? cols : 3 * cols, blk = stack ? rows * cols : cols;
  bf16_t* d = dst + r * pitch + c;
  *reinterpret_cast<u32x2*>(d) = hi;
  *reinterpret_cast<u32x2*>(d + blk) = role == 0 ? lo : hi;
  *reinterpret_cast<u32x2*>(d + 2 * blk) = role == 0 ? hi : lo;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int segclip_split3_bf16(const float* src, void* dst, int64_t rows, int64_t cols, int64_t ld, int stack, int role,
                                   void* stream) {
  SEGCLIP_REQUIRE(cols % 4 == 0 && ld % 4 == 0, "split3: cols and ld must be multiples of 4");
  SEGCLIP_REQUIRE((role == 0 || role == 1) && (stack == 0 || stack == 1), "split3: role / stack must be 0 or 1");
  if (rows == 0 || cols == 0) return 0;
  const int64_t n = rows * (cols >> 2);
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, ST, src, (bf16_t*)dst, rows, cols, ld, stack, role);
  SEGCLIP_CHECK_LAUNCH("split3");
  return 0;
}

extern "C" int segclip_cast(const void* src, void* dst, int64_t n, int sd, int dd, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(cast_kernel, dim3(grid1d(n, 4)), dim3(TPB), 0, ST, src, dst, n, sd, dd);
  SEGCLIP_CHECK_LAUNCH("cast");
  return 0;
}
extern "C" int segclip_add(const void* a, const void* b, void* out, int64_t n, int dt, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(add_kernel, dim3(grid1d(n)), dim3(TPB), 0, ST, a, b, out, n, dt);
  SEGCLIP_CHECK_LAUNCH("add");
  return 0;
}
extern "C" int segclip_act_fwd(const void* x, void* y, int64_t n, int act, int dt, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(grid1d(n)), dim3(TPB), 0, ST, x, y, n, act, dt);
  SEGCLIP_CHECK_LAUNCH("act_fwd");
  return 0;
}
extern "C" int segclip_act_bwd(const void* dy, const void* x, void* dx, int64_t n, int act, int dt, void* stream) {
  if (n == 0) return 0;
  if (n % 4 == 0 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0)
    hipLaunchKernelGGL(act_bwd_vec4_kernel, dim3(grid1d(n / 4)), dim3(TPB), 0, ST, dy, x, dx, n / 4, act, dt);
  else
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid1d(n)), dim3(TPB), 0, ST, dy, x, dx, n, act, dt);
  SEGCLIP_CHECK_LAUNCH("act_bwd");
  return 0;
}
extern "C" int segclip_scale(const float* x, const float* s, float* out, int64_t n, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(scale_kernel, dim3(grid1d(n)), dim3(TPB), 0, ST, x, s, out, n);
  SEGCLIP_CHECK_LAUNCH("scale");
  return 0;
}
extern "C" int segclip_reduce_sum(const float* x, float* out, int64_t n, float scale, void* stream) {
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, ST, x, out, n, scale);
  SEGCLIP_CHECK_LAUNCH("reduce_sum");
  return 0;
}
extern "C" size_t segclip_colsum_ws_bytes(int64_t M, int64_t N) { return (size_t)colsum_chunks(M, N) * N * sizeof(float); }
extern "C" int segclip_colsum(const void* X, float* out, void* ws, int64_t M, int64_t N, int64_t ld, int dt, void* stream) {
  SEGCLIP_REQUIRE(ws != nullptr, "colsum: workspace required");
  if (N == 0) return 0;
  // a short fp32 matrix (per-sample partial sums): one pass of the row-reduce kernel, no partial stage
  if (dt == SEGCLIP_F32 && M <= 4096 && N % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
    launch_reduce_rows((const float*)X, M, N, ld, out, nullptr, nullptr, N, ST);
    SEGCLIP_CHECK_LAUNCH("colsum_rows");
    return 0;
  }
  const int64_t nchunk = colsum_chunks(M, N);
  const int64_t rows_per = cdiv(M > 0 ? M : 1, nchunk);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)cdiv(N, 256), (unsigned)nchunk), dim3(256), 0, ST, X, (float*)ws,
                     M, N, ld, dt, rows_per);
  SEGCLIP_CHECK_LAUNCH("colsum_partial");
  if (N % 4 == 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0)
    launch_reduce_rows((const float*)ws, nchunk, N, N, out, nullptr, nullptr, N, ST);
  else
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, ST, (const float*)ws, out, nchunk, N);
  SEGCLIP_CHECK_LAUNCH("colsum_final");
  return 0;
}
extern "C" int segclip_im2col_ld(const float* image, void* cols, int64_t B, int64_t C, int64_t H, int64_t W, int64_t p,
                                 int layout, int od, int64_t ld, void* stream) {
  SEGCLIP_REQUIRE(H % p == 0 && W % p == 0, "im2col: %lldx%lld not divisible by patch %lld", (long long)H, (long long)W,
                  (long long)p);
  SEGCLIP_REQUIRE(ld >= C * p * p, "im2col: ld=%lld < C*p*p=%lld", (long long)ld, (long long)(C * p * p));
  const int64_t total = B * (H / p) * (W / p) * ld;
  if (total == 0) return 0;
  if (layout == 0 && p % 8 == 0 && ld % 8 == 0 && W % 4 == 0 && (reinterpret_cast<uintptr_t>(image) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(cols) & 15) == 0 && (od == SEGCLIP_BF16 || od == SEGCLIP_F32))
    hipLaunchKernelGGL(im2col_vec8_kernel, dim3(grid1d(total / 8)), dim3(TPB), 0, ST, image, cols, B, (int)C, (int)H, (int)W,
                       (int)p, od, ld);
  else
    hipLaunchKernelGGL(im2col_kernel, dim3(grid1d(total)), dim3(TPB), 0, ST, image, cols, B, (int)C, (int)H, (int)W, (int)p,
                       layout, od, ld);
  SEGCLIP_CHECK_LAUNCH("im2col");
  return 0;
}
extern "C" int segclip_im2col(const float* image, void* cols, int64_t B, int64_t C, int64_t H, int64_t W, int64_t p,
                              int layout, int od, void* stream) {
  return segclip_im2col_ld(image, cols, B, C, H, W, p, layout, od, C * p * p, stream);
}
extern "C" int segclip_vis_assemble(const void* patches, const float* cls, const float* pos, float* x, int64_t B, int64_t T,
                                    int64_t D, int pd, void* stream) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(vis_assemble_kernel, dim3(grid1d(B * (T + 1) * D)), dim3(TPB), 0, ST, patches, cls, pos, x, B, (int)T,
                     (int)D, pd);
  SEGCLIP_CHECK_LAUNCH("vis_assemble");
  return 0;
}
extern "C" int segclip_embed_fwd(const int64_t* ids, const float* table, const float* pos, float* out, int64_t B, int64_t L,
                                 int64_t D, int64_t vocab, void* stream) {
  SEGCLIP_REQUIRE(D % 4 == 0, "embed: D=%lld must be a multiple of 4", (long long)D);
  if (B == 0) return 0;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid1d(B * L * D / 4)), dim3(TPB), 0, ST, ids, table, pos, out, B * L, (int)L,
                     (int)D, vocab);
  SEGCLIP_CHECK_LAUNCH("embed_fwd");
  return 0;
}
extern "C" int segclip_embed_bwd(const int64_t* ids, const float* dout, float* dtable, float* dpos, int64_t B, int64_t L,
                                 int64_t D, int64_t vocab, void* stream) {
  if (B == 0) return 0;
  if (dtable) {
    hipLaunchKernelGGL(embed_bwd_table_kernel, dim3(grid1d(B * L * D)), dim3(TPB), 0, ST, ids, dout, dtable, B * L, (int)D,
                       vocab);
    SEGCLIP_CHECK_LAUNCH("embed_bwd_table");
  }
  if (dpos) {
    SEGCLIP_REQUIRE(D % 4 == 0, "embed_bwd: D=%lld must be a multiple of 4", (long long)D);
    hipLaunchKernelGGL(embed_bwd_pos_kernel, dim3((unsigned)cdiv(L * D / 4, 64)), dim3(64), 0, ST, dout, dpos, B, (int)L, (int)D);
    SEGCLIP_CHECK_LAUNCH("embed_bwd_pos");
  }
  return 0;
}
extern "C" int segclip_gather_rows(const void* src, const int64_t* idx, void* out, int64_t B, int64_t Ts, int64_t To,
                                   int64_t D, int dt, void* stream) {
  if (B * To * D == 0) return 0;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid1d(B * To * D)), dim3(TPB), 0, ST, src, idx, out, B, (int)Ts, (int)To,
                     (int)D, dt, 0);
  SEGCLIP_CHECK_LAUNCH("gather_rows");
  return 0;
}
extern "C" int segclip_scatter_rows(const void* dout, const int64_t* idx, void* dsrc, int64_t B, int64_t Ts, int64_t To,
                                    int64_t D, int dt, void* stream) {
  if (B * To * D == 0) return 0;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid1d(B * To * D)), dim3(TPB), 0, ST, dout, idx, dsrc, B, (int)Ts, (int)To,
                     (int)D, dt, 1);
  SEGCLIP_CHECK_LAUNCH("scatter_rows");
  return 0;
}
extern "C" int segclip_gumbel_from_uniform(const float* u, float* out, int64_t n, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(gumbel_from_uniform_kernel, dim3(grid1d(n)), dim3(TPB), 0, ST, u, out, n);
  SEGCLIP_CHECK_LAUNCH("gumbel_from_uniform");
  return 0;
}
extern "C" int segclip_mean_cat_fwd(const float* x, float* out, int64_t B, int64_t T, int64_t D, void* stream) {
  SEGCLIP_REQUIRE(D % 4 == 0 && T > 0, "mean_cat: D=%lld must be a multiple of 4, T=%lld positive", (long long)D, (long long)T);
  if (B == 0) return 0;
  const int D4 = (int)(D / 4), chunks = (D4 + 63) / 64;
  hipLaunchKernelGGL(mean_cat_fwd_kernel, dim3((unsigned)(B * chunks)), dim3(64), 0, ST, x, out, (int)T, D4, chunks);
  SEGCLIP_CHECK_LAUNCH("mean_cat_fwd");
  return 0;
}
extern "C" int segclip_mean_cat_bwd(const float* dout, float* dx, int64_t B, int64_t T, int64_t D, void* stream) {
  SEGCLIP_REQUIRE(D % 4 == 0 && T > 0, "mean_cat: D=%lld must be a multiple of 4, T=%lld positive", (long long)D, (long long)T);
  if (B == 0) return 0;
  hipLaunchKernelGGL(mean_cat_bwd_kernel, dim3(grid1d(B * T * (D / 4))), dim3(TPB), 0, ST, dout, dx, B, (int)T, (int)(D / 4));
  SEGCLIP_CHECK_LAUNCH("mean_cat_bwd");
  return 0;
}
extern "C" int segclip_mae_unshuffle_fwd(const float* x, const float* mask_token, const int64_t* ids, const float* pos, float* out,
                                         int64_t B, int64_t K, int64_t L, int64_t D, void* stream) {
  SEGCLIP_REQUIRE(D % 4 == 0 && K <= L, "mae_unshuffle: D=%lld must be a multiple of 4 and K=%lld <= L=%lld", (long long)D, (long long)K, (long long)L);
  if (B * L == 0) return 0;
  hipLaunchKernelGGL(mae_unshuffle_fwd_kernel, dim3(grid1d(B * L * (D / 4))), dim3(TPB), 0, ST, x, mask_token, ids, pos, out, B, (int)K,
                     (int)L, (int)(D / 4));
  SEGCLIP_CHECK_LAUNCH("mae_unshuffle_fwd");
  return 0;
}
extern "C" int segclip_mae_unshuffle_bwd(const float* dout, const int64_t* ids, float* dx, float* dpos, float* mpart, int64_t B, int64_t K,
                                         int64_t L, int64_t D, void* stream) {
  SEGCLIP_REQUIRE(D % 4 == 0 && K <= L, "mae_unshuffle: D=%lld must be a multiple of 4 and K=%lld <= L=%lld", (long long)D, (long long)K, (long long)L);
  if (L == 0) return 0;
  const int D4 = (int)(D / 4), chunks = (D4 + 63) / 64;
  hipLaunchKernelGGL(mae_unshuffle_bwd_kernel, dim3((unsigned)(L * chunks)), dim3(64), 0, ST, dout, ids, dx, dpos, mpart, B, (int)K, (int)L, D4,
                     chunks);
  SEGCLIP_CHECK_LAUNCH("mae_unshuffle_bwd");
  return 0;
}
extern "C" int segclip_assign_fwd(const float* logits, const float* gumbel, float tau, float* y_soft, float* soft,
                                  uint8_t* idx, float* hard, float* counts, int64_t B, int64_t G, int64_t T, void* stream) {
  SEGCLIP_REQUIRE(G <= 255, "assign: G=%lld too large", (long long)G);
  if (B == 0) return 0;
  hipError_t e = hipMemsetAsync(counts, 0, (size_t)B * G * sizeof(float), ST);
  SEGCLIP_REQUIRE(e == hipSuccess, "assign: memset failed: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(assign_fwd_kernel, dim3((unsigned)cdiv(B * T, TPB)), dim3(TPB), 0, ST, logits, gumbel, tau, y_soft,
                     soft, idx, hard, counts, B, (int)G, (int)T);
  SEGCLIP_CHECK_LAUNCH("assign_fwd");
  return 0;
}
extern "C" int segclip_assign_bwd(const float* dhard, const float* y_soft, float tau, float* dlogits, int64_t B, int64_t G,
                                  int64_t T, void* stream) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(assign_bwd_kernel, dim3((unsigned)cdiv(B * T, TPB)), dim3(TPB), 0, ST, dhard, y_soft, tau, dlogits, B,
                     (int)G, (int)T);
  SEGCLIP_CHECK_LAUNCH("assign_bwd");
  return 0;
}
extern "C" int segclip_l2norm_fwd(const float* x, float* y, float* norm, int64_t rows, int64_t cols, void* stream) {
  if (rows == 0) return 0;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, ST, x, y, norm, rows, (int)cols);
  SEGCLIP_CHECK_LAUNCH("l2norm_fwd");
  return 0;
}
extern "C" int segclip_l2norm_bwd(const float* dy, const float* y, const float* norm, float* dx, int64_t rows, int64_t cols,
                                  void* stream) {
  if (rows == 0) return 0;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, ST, dy, y, norm, dx, rows, (int)cols);
  SEGCLIP_CHECK_LAUNCH("l2norm_bwd");
  return 0;
}
extern "C" int segclip_ce_fwd(const float* logits, float* lse, float* loss_rows, int64_t rows, int64_t cols,
                              int64_t label_offset, void* stream) {
  SEGCLIP_REQUIRE(rows + label_offset <= cols && label_offset >= 0, "ce: labels [%lld,%lld) outside %lld columns",
                  (long long)label_offset, (long long)(rows + label_offset), (long long)cols);
  if (rows == 0) return 0;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, ST, logits, lse, loss_rows, rows, cols,
                     label_offset);
  SEGCLIP_CHECK_LAUNCH("ce_fwd");
  return 0;
}
extern "C" int segclip_ce_bwd(const float* logits, const float* lse, const float* gscale_ptr, float gscale, float* dlogits,
                              int64_t rows, int64_t cols, int64_t label_offset, void* stream) {
  if (rows == 0) return 0;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(grid1d(rows * cols)), dim3(TPB), 0, ST, logits, lse, gscale_ptr, gscale, dlogits,
                     rows, cols, label_offset);
  SEGCLIP_CHECK_LAUNCH("ce_bwd");
  return 0;
}
extern "C" int segclip_ce_labels_fwd(const float* logits, const int64_t* labels, int64_t ignore_index, float* lse,
                                     float* loss_rows, float* valid, int64_t rows, int64_t cols, void* stream) {
  if (rows == 0) return 0;
  hipLaunchKernelGGL(ce_labels_fwd_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, ST, logits, labels, ignore_index, lse,
                     loss_rows, valid, rows, cols);
  SEGCLIP_CHECK_LAUNCH("ce_labels_fwd");
  return 0;
}
extern "C" int segclip_ce_labels_bwd(const float* logits, const float* lse, const int64_t* labels, int64_t ignore_index,
                                     const float* gscale, const float* inv_count, float* dlogits, int64_t rows,
                                     int64_t cols, void* stream) {
  if (rows == 0) return 0;
  hipLaunchKernelGGL(ce_labels_bwd_kernel, dim3(grid1d(rows * cols)), dim3(TPB), 0, ST, logits, lse, labels, ignore_index,
                     gscale, inv_count, dlogits, rows, cols);
  SEGCLIP_CHECK_LAUNCH("ce_labels_bwd");
  return 0;
}
extern "C" int segclip_superpixel_kl(const float* hard, const int64_t* seg, float* loss_rows, float* dhard, int64_t B,
                                     int64_t G, int64_t T, void* stream) {
  SEGCLIP_REQUIRE(G <= KL_MAXG, "superpixel_kl: G=%lld > %d", (long long)G, KL_MAXG);
  const size_t sm = (size_t)T * G * sizeof(float) + (size_t)T * sizeof(int);
  SEGCLIP_REQUIRE(sm <= 60000, "superpixel_kl: T=%lld too large", (long long)T);
  if (B == 0) return 0;
  hipLaunchKernelGGL(superpixel_kl_kernel, dim3((unsigned)B), dim3(256), sm, ST, hard, seg, loss_rows, dhard, B, (int)G, (int)T);
  SEGCLIP_CHECK_LAUNCH("superpixel_kl");
  return 0;
}
extern "C" int segclip_masked_mse_fwd(const void* pred, const float* target, const float* mask, float* loss_rows, int64_t B,
                                      int64_t T, int64_t Dp, int pd, void* stream) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(masked_mse_fwd_kernel, dim3((unsigned)cdiv(B * T, 4)), dim3(256), 0, ST, pred, target, mask, loss_rows, B,
                     (int)T, (int)Dp, pd);
  SEGCLIP_CHECK_LAUNCH("masked_mse_fwd");
  return 0;
}
extern "C" int segclip_masked_mse_bwd(const void* pred, const float* target, const float* mask, const float* gscale_ptr,
                                      const float* mask_sum, float gscale, void* dpred, int64_t B, int64_t T, int64_t Dp,
                                      int pd, void* stream) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(masked_mse_bwd_kernel, dim3(grid1d(B * (T + 1) * Dp)), dim3(TPB), 0, ST, pred, target, mask, gscale_ptr,
                     mask_sum, gscale, dpred, B, (int)T, (int)Dp, pd);
  SEGCLIP_CHECK_LAUNCH("masked_mse_bwd");
  return 0;
}
extern "C" int segclip_mask_sort(const float* noise, int64_t* ids_shuffle, int64_t* ids_restore, float* mask, int64_t B,
                                 int64_t L, int64_t len_keep, void* stream) {
  SEGCLIP_REQUIRE(L <= 4096, "mask_sort: L=%lld > 4096", (long long)L);
  if (B == 0) return 0;
  hipLaunchKernelGGL(mask_sort_kernel, dim3((unsigned)B), dim3(256), (size_t)L * sizeof(float), ST, noise, ids_shuffle,
                     ids_restore, mask, (int)L, (int)len_keep);
  SEGCLIP_CHECK_LAUNCH("mask_sort");
  return 0;
}
extern "C" int segclip_interp_bicubic(const float* src, float* dst, int64_t n_in, int64_t h, int64_t w, int64_t D,
                                      void* stream) {
  SEGCLIP_REQUIRE(n_in > 0 && h > 0 && w > 0 && D > 0, "interp_bicubic: empty grid");
  hipLaunchKernelGGL(interp_bicubic_kernel, dim3(grid1d(h * w * D)), dim3(TPB), 0, ST, src, dst, (int)n_in, (int)h, (int)w, (int)D);
  SEGCLIP_CHECK_LAUNCH("interp_bicubic");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// segclip_reduce_multi: up to 16 deferred reductions of one kind in ONE launch (the entry table travels in the kernel
// arguments).  Workgroup b serves entry e with start[e] <= b < start[e+1].
namespace {
struct ReduceLaunch {
  segclip_reduce_entry e[SEGCLIP_REDUCE_MAX];
  int start[SEGCLIP_REDUCE_MAX + 1];
  int n;
};

// SLABS: out[i] = scale * sum_s src[s*width + i]; 256 threads x float4 per workgroup
__global__ __launch_bounds__(256) void reduce_multi_slabs_kernel(ReduceLaunch L) {
  int ei = 0;
  while (ei + 1 < L.n && (int)blockIdx.x >= L.start[ei + 1]) ++ei;
  const segclip_reduce_entry& E = L.e[ei];
  const int64_t total4 = E.width >> 2;
  const int64_t i = (int64_t)((int)blockIdx.x - L.start[ei]) * 256 + threadIdx.x;
  if (i >= total4) return;
  const f32x4* p = reinterpret_cast<const f32x4*>(E.src) + i;
  const int splits = (int)E.rows;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  int s = 0;
#pragma unroll 1
  for (; s + 8 <= splits; s += 8) {
    f32x4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = p[(int64_t)(s + u) * total4];
    v += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  }
  if (s + 4 <= splits) {
    const f32x4 a = p[(int64_t)s * total4], b = p[(int64_t)(s + 1) * total4], c = p[(int64_t)(s + 2) * total4],
                d = p[(int64_t)(s + 3) * total4];
    v += (a + b) + (c + d);
    s += 4;
  }
  for (; s < splits; ++s) v += p[(int64_t)s * total4];
  v *= E.scale;
  if (E.out_dtype == SEGCLIP_BF16) reinterpret_cast<u32x2*>(E.out0)[i] = u32x2{pack2bf(v.x, v.y), pack2bf(v.z, v.w)};
  else reinterpret_cast<f32x4*>(E.out0)[i] = v;
}

// ROWS: column sums of a (rows x width) partial matrix; workgroup = 16 columns (4 lanes x float4) x 64 row lanes
// (same arithmetic and summation order as reduce_rows_kernel)
__global__ __launch_bounds__(256) void reduce_multi_rows_kernel(ReduceLaunch L) {
  __shared__ f32x4 red[4][4];
  int ei = 0;
  while (ei + 1 < L.n && (int)blockIdx.x >= L.start[ei + 1]) ++ei;
  const segclip_reduce_entry& E = L.e[ei];
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int64_t c = ((int64_t)((int)blockIdx.x - L.start[ei]) * 4 + cl) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c < E.width) {
    const float* p = E.src + c;
    int64_t r = rl;
#pragma unroll 1
    for (; r + 7 * 64 < E.rows; r += 8 * 64) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (r + u * 64) * E.ld);
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; r < E.rows; r += 64) s += *reinterpret_cast<const f32x4*>(p + r * E.ld);
  }
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) {
    s.x += __shfl_xor(s.x, o, 64);
    s.y += __shfl_xor(s.y, o, 64);
    s.z += __shfl_xor(s.z, o, 64);
    s.w += __shfl_xor(s.w, o, 64);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < 4) red[w][cl] = s;
  __syncthreads();
  if (threadIdx.x < 4 && c < E.width) {
    const f32x4 t = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    const int64_t k = c / E.seg;                   // seg % 4 == 0: the four columns share a segment
    float* o = k == 0 ? (float*)E.out0 : (k == 1 ? E.out1 : E.out2);
    if (o) *reinterpret_cast<f32x4*>(o + (c - k * E.seg)) = t;
  }
}
}  // namespace

extern "C" int segclip_reduce_multi(const segclip_reduce_entry* entries, int n, int kind, void* stream) {
  SEGCLIP_REQUIRE(n >= 0 && n <= SEGCLIP_REDUCE_MAX, "reduce_multi: n=%d entries (max %d)", n, SEGCLIP_REDUCE_MAX);
  SEGCLIP_REQUIRE(kind == SEGCLIP_REDUCE_SLABS || kind == SEGCLIP_REDUCE_ROWS, "reduce_multi: unknown kind %d", kind);
  if (n == 0) return 0;
  ReduceLaunch L;
  int64_t blocks = 0;
  for (int i = 0; i < n; ++i) {
    const segclip_reduce_entry& e = entries[i];
    SEGCLIP_REQUIRE(e.src && e.out0 && e.rows >= 1 && e.width >= 4 && e.width % 4 == 0,
                    "reduce_multi: entry %d: src/out0 null, rows < 1 or width %% 4 != 0", i);
    SEGCLIP_REQUIRE((reinterpret_cast<uintptr_t>(e.src) & 15) == 0 && (reinterpret_cast<uintptr_t>(e.out0) & 15) == 0,
                    "reduce_multi: entry %d: src / out0 must be 16-byte aligned", i);
    if (kind == SEGCLIP_REDUCE_ROWS) {
      SEGCLIP_REQUIRE(e.seg >= 4 && e.seg % 4 == 0 && e.ld % 4 == 0 && e.width <= 3 * e.seg && e.width % e.seg == 0,
                      "reduce_multi: entry %d: seg / ld must be multiples of 4, width = 1..3 segments", i);
      SEGCLIP_REQUIRE((!e.out1 || (reinterpret_cast<uintptr_t>(e.out1) & 15) == 0) &&
                      (!e.out2 || (reinterpret_cast<uintptr_t>(e.out2) & 15) == 0), "reduce_multi: entry %d: unaligned out", i);
    } else {
      SEGCLIP_REQUIRE(e.out_dtype == SEGCLIP_F32 || e.out_dtype == SEGCLIP_BF16, "reduce_multi: entry %d: bad out dtype", i);
    }
    L.e[i] = e;
    L.start[i] = (int)blocks;
    blocks += kind == SEGCLIP_REDUCE_SLABS ? cdiv(e.width / 4, 256) : cdiv(e.width, 16);
    SEGCLIP_REQUIRE(blocks < ((int64_t)1 << 31), "reduce_multi: too many workgroups");
  }
  L.start[n] = (int)blocks;
  L.n = n;
  if (kind == SEGCLIP_REDUCE_SLABS)
    hipLaunchKernelGGL(reduce_multi_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, L);
  else
    hipLaunchKernelGGL(reduce_multi_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, L);
  SEGCLIP_CHECK_LAUNCH("reduce_multi");
  return 0;
}
