// Exact-fp32 GEMM on the f32-input matrix cores (v_mfma_f32_32x32x2_f32): the 1e-3 parity path.
// C[z](m,n) = epi(alpha * sum_k A(m,k) * B(n,k)),  arbitrary element strides for A and B
// (forward NT, dgrad NN and wgrad TN are the same kernel with different strides).
//
// Tile 128x128x32, 256 threads = 4 waves in a 2x2 arrangement, each wave 64x64 = 2x2 MFMA tiles.
// LDS image is k-major ([k][m], pitch 129 floats) so that the one-float-per-lane MFMA operand
// (lane l: row l&31, k = l>>5) is a conflict-free ds_read_b32 of 32 consecutive floats.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, PITCH = 129, NT = 256;

struct GemmArgs {
  const float* A; const float* B; float* C;
  const float* bias; const float* residual; float* aux;
  int64_t M, N, K, sam, sak, sbn, sbk, ldc, ldr, ldaux;
  int64_t nb2, bsA1, bsA2, bsB1, bsB2, bsC1, bsC2, bsR1, bsR2;
  int act, mul_dact, aux_kind; float alpha;
};

// stage a [rows x BK] tile of X(row,k) = X[row*sr + k*sk] into lds[k][row]
__device__ __forceinline__ void stage_tile(const float* __restrict__ X, int64_t row0, int64_t nrows, int64_t k0,
                                           int64_t K, int64_t sr, int64_t sk, float* lds, int tid) {
  if (sk == 1) {
    // k contiguous: thread -> (row = tid/8 + 32*i, 4 consecutive k starting at (tid%8)*4)
    const int kq = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (tid >> 3) + 32 * i;
      const int64_t gr = row0 + r;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (gr < nrows) {
        const float* p = X + gr * sr + (k0 + kq);
        if (k0 + kq + 3 < K && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
          const float4 t = *reinterpret_cast<const float4*>(p);
          v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (k0 + kq + j < K) v[j] = p[j];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) lds[(kq + j) * PITCH + r] = v[j];
    }
  } else {
    // row index contiguous (sr == 1) or fully general: thread -> (k = tid/32 + 8*i, rows (tid%32)*4..+3)
    const int rq = (tid & 31) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = (tid >> 5) + 8 * i;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (k0 + k < K) {
        const float* p = X + (row0 + rq) * sr + (k0 + k) * sk;
        if (sr == 1 && row0 + rq + 3 < nrows && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
          const float4 t = *reinterpret_cast<const float4*>(p);
          v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (row0 + rq + j < nrows) v[j] = p[(int64_t)j * sr];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) lds[k * PITCH + rq + j] = v[j];
    }
  }
}

__global__ __launch_bounds__(NT) void gemm_f32_kernel(GemmArgs g) {
  __shared__ float As[BK * PITCH];
  __shared__ float Bs[BK * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t z = blockIdx.z, z1 = z / g.nb2, z2 = z % g.nb2;
  const float* A = g.A + z1 * g.bsA1 + z2 * g.bsA2;
  const float* B = g.B + z1 * g.bsB1 + z2 * g.bsB2;
  const int64_t coff = z1 * g.bsC1 + z2 * g.bsC2;
  const int64_t roff = z1 * g.bsR1 + z2 * g.bsR2;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int li = lane & 31, lk = lane >> 5;
  for (int64_t k0 = 0; k0 < g.K; k0 += BK) {
    stage_tile(A, m0, g.M, k0, g.K, g.sam, g.sak, As, tid);
    stage_tile(B, n0, g.N, k0, g.K, g.sbn, g.sbk, Bs, tid);
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < BK; kk += 2) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[(kk + lk) * PITCH + wm * 64 + i * 32 + li];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[(kk + lk) * PITCH + wn * 64 + j * 32 + li];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue: C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t n = n0 + wn * 64 + j * 32 + li;
      if (n >= g.N) continue;
      const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m >= g.M) continue;
        float v = g.alpha * acc[i][j][r];
        if (g.mul_dact) {
          const float sd = g.aux[coff + m * g.ldaux + n];
          v *= g.aux_kind ? sd : apply_act_grad(g.act, sd);
        } else {
          v += bv;
          if (g.act != SEGCLIP_ACT_NONE) {
            if (g.aux) g.aux[coff + m * g.ldaux + n] = g.aux_kind ? apply_act_grad(g.act, v) : v;
            v = apply_act(g.act, v);
          }
          if (g.residual) v += g.residual[roff + m * g.ldr + n];
        }
        g.C[coff + m * g.ldc + n] = v;
      }
    }
}

}  // namespace

int segclip_gemm_f32_launch(const segclip_gemm_desc* d, hipStream_t stream) {
  GemmArgs g;
  g.A = (const float*)d->A; g.B = (const float*)d->B; g.C = (float*)d->C;
  g.bias = d->bias; g.residual = (const float*)d->residual; g.aux = (float*)d->aux;
  g.M = d->M; g.N = d->N; g.K = d->K; g.sam = d->sam; g.sak = d->sak; g.sbn = d->sbn; g.sbk = d->sbk;
  g.ldc = d->ldc; g.ldr = d->ldr; g.ldaux = d->ldaux;
  g.nb2 = d->nb2 > 0 ? d->nb2 : 1;
  g.bsA1 = d->bsA1; g.bsA2 = d->bsA2; g.bsB1 = d->bsB1; g.bsB2 = d->bsB2; g.bsC1 = d->bsC1; g.bsC2 = d->bsC2; g.bsR1 = d->bsR1; g.bsR2 = d->bsR2;
  g.act = d->act; g.mul_dact = d->mul_dact; g.aux_kind = d->aux_kind; g.alpha = d->alpha;
  const int64_t nb = (d->nb1 > 0 ? d->nb1 : 1) * g.nb2;
  dim3 grid((unsigned)cdiv(d->N, BN), (unsigned)cdiv(d->M, BM), (unsigned)nb);
  SEGCLIP_REQUIRE(grid.y <= 65535 && nb <= 65535, "gemm_f32: grid too large (M=%lld batch=%lld)",
                  (long long)d->M, (long long)nb);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(NT), 0, stream, g);
  SEGCLIP_CHECK_LAUNCH("gemm_f32");
  return 0;
}
