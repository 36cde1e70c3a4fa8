// out[c] = sum_r part[r][c] for a (rows x width) fp32 partial matrix: the deterministic second stage of every
// column reduction here (LayerNorm dgamma|dbeta|colsum(dres), bias-gradient column sums, GEMM-epilogue column sums).
// Workgroup = 16 columns (4 lanes x float4) x 64 row lanes, so a 1024 x 2304 partial matrix is read by 144
// workgroups with 16 independent 16-byte loads in flight per lane (the previous 64-column x 16-row-lane
// version ran on 36 CUs with a 64-deep dependent chain: 25-30 us for 9 MB).
// The output may be split into up to three arrays of `seg` columns each (out_k[c - k*seg]).
#pragma once
#include "common.h"

namespace {

// RL = row lanes per workgroup (64 or 256): with 256 a 1024-row partial matrix needs four loads per lane, all in
// flight at once (one memory latency instead of a chain of batches).
template <int RL>
__global__ __launch_bounds__(RL * 4) void reduce_rows_kernel(const float* __restrict__ part, int64_t rows, int64_t width,
                                                             int64_t ld, float* __restrict__ out0,
                                                             float* __restrict__ out1, float* __restrict__ out2,
                                                             int64_t seg) {
  constexpr int NW = RL * 4 / 64;
  __shared__ f32x4 red[NW][4];
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;  // a wave holds 16 row lanes x 4 column lanes
  const int64_t c = ((int64_t)blockIdx.x * 4 + cl) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c < width) {
    const float* p = part + c;
    int64_t r = rl;
#pragma unroll 1
    for (; r + 3 * RL < rows; r += 4 * RL) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (r + u * RL) * ld);
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    for (; r < rows; r += RL) s += *reinterpret_cast<const f32x4*>(p + r * ld);
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
  if (threadIdx.x < 4 && c < width) {
    f32x4 t = red[0][cl];
#pragma unroll
    for (int k = 1; k < NW; ++k) t += red[k][cl];
    const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t cc = c + j;
      if (cc >= width) break;
      const int64_t k = cc / seg;
      float* o = k == 0 ? out0 : (k == 1 ? out1 : out2);
      o[cc - k * seg] = v[j];
    }
  }
}

// requires width % 4 == 0, ld % 4 == 0 and a 16-byte aligned `part`
inline void launch_reduce_rows(const float* part, int64_t rows, int64_t width, int64_t ld, float* out0, float* out1,
                               float* out2, int64_t seg, hipStream_t st) {
  if (rows >= 512)
    hipLaunchKernelGGL(reduce_rows_kernel<256>, dim3((unsigned)cdiv(width, 16)), dim3(1024), 0, st, part, rows, width, ld,
                       out0, out1, out2, seg);
  else
    hipLaunchKernelGGL(reduce_rows_kernel<64>, dim3((unsigned)cdiv(width, 16)), dim3(256), 0, st, part, rows, width, ld,
                       out0, out1, out2, seg);
}

}  // namespace
