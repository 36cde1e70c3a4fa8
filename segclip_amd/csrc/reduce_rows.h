// out[c] = sum_r part[r][c] for a (rows x width) fp32 partial matrix: the deterministic second stage of every
// column reduction here (LayerNorm dgamma|dbeta|colsum(dres), bias-gradient column sums, GEMM-epilogue column sums).
// Workgroup = 16 columns (4 lanes x float4) x 64 row lanes, so a 1024 x 2304 partial matrix is read by 144
// workgroups with 16 independent 16-byte loads in flight per lane (the previous 64-column x 16-row-lane
// version ran on 36 CUs with a 64-deep dependent chain: 25-30 us for 9 MB).
// The output may be split into up to three arrays of `seg` columns each (out_k[c - k*seg]).
#pragma once
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ part, int64_t rows, int64_t width,
                                                          int64_t ld, float* __restrict__ out0, float* __restrict__ out1,
                                                          float* __restrict__ out2, int64_t seg) {
  __shared__ f32x4 red[4][4];
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;  // rl 0..63; a wave holds 16 row lanes x 4 column lanes
  const int64_t c = ((int64_t)blockIdx.x * 4 + cl) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c < width) {
    const float* p = part + c;
    int64_t r = rl;
    // eight independent 16-byte loads in flight per lane (the chain is latency-, not bandwidth-bound)
#pragma unroll 1
    for (; r + 7 * 64 < rows; r += 8 * 64) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (r + u * 64) * ld);
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; r < rows; r += 64) s += *reinterpret_cast<const f32x4*>(p + r * ld);
  }
  // 16 row lanes of a wave: lanes differ in bits 2..5
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
    const f32x4 t = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
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
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)cdiv(width, 16)), dim3(256), 0, st, part, rows, width, ld, out0,
                     out1, out2, seg);
}

}  // namespace
