// Exact-fp32 GEMM on the f32-input matrix cores (v_mfma_f32_32x32x2_f32): the 1e-3 parity path.
// C[z](m,n) = epi(alpha * sum_k A(m,k) * B(n,k)),  arbitrary element strides for A and B
// (forward NT, dgrad NN and wgrad TN are the same kernel with different strides).
//
// Tile 128x128x32 (or 64x64x64 for small problems), 256 threads = 4 waves in a 2x2 arrangement, each wave 2x2 (1x1) MFMA
// tiles; the next K-step's operands are loaded into registers while the current one is multiplied.
// LDS image is k-major ([k][m], pitch rows+1 floats) so that the one-float-per-lane MFMA operand
// (lane l: row l&31, k = l>>5) is a conflict-free ds_read_b32 of 32 consecutive floats.
#include "common.h"

namespace {

constexpr int NT = 256;

struct GemmArgs {
  const float* A; const float* B; float* C;
  const float* bias; const float* residual; float* aux;
  int64_t M, N, K, sam, sak, sbn, sbk, ldc, ldr, ldaux;
  int64_t nb2, bsA1, bsA2, bsB1, bsB2, bsC1, bsC2, bsR1, bsR2;
  int act, mul_dact, aux_kind; float alpha;
};

// One [ROWS x BKT] tile of X(row,k) = X[row*sr + k*sk] goes global -> registers (tile_load: 4 x 16 bytes per thread,
// issued one K-step ahead of its use) -> LDS image lds[k][row] (tile_store).  Chunk c = it*256 + tid:
//   k contiguous (sk == 1): row = c / (BKT/4), 4 consecutive k;   otherwise: k = c / (ROWS/4), 4 consecutive rows.
template <int ROWS, int BKT>
__device__ __forceinline__ void tile_load(const float* __restrict__ X, int64_t row0, int64_t nrows, int64_t k0, int64_t K,
                                          int64_t sr, int64_t sk, int tid, float (&v)[ROWS * BKT / 1024][4]) {
  constexpr int NCH = ROWS * BKT / 1024;
#pragma unroll
  for (int it = 0; it < NCH; ++it) {
    const int c = it * NT + tid;
    v[it][0] = v[it][1] = v[it][2] = v[it][3] = 0.f;
    if (sk == 1) {
      const int r = c / (BKT / 4), kq = (c % (BKT / 4)) * 4;
      const int64_t gr = row0 + r;
      if (gr < nrows) {
        const float* p = X + gr * sr + (k0 + kq);
        if (k0 + kq + 3 < K && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
          const float4 t = *reinterpret_cast<const float4*>(p);
          v[it][0] = t.x; v[it][1] = t.y; v[it][2] = t.z; v[it][3] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (k0 + kq + j < K) v[it][j] = p[j];
        }
      }
    } else {
      const int k = c / (ROWS / 4), rq = (c % (ROWS / 4)) * 4;
      if (k0 + k < K) {
        const float* p = X + (row0 + rq) * sr + (k0 + k) * sk;
        if (sr == 1 && row0 + rq + 3 < nrows && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
          const float4 t = *reinterpret_cast<const float4*>(p);
          v[it][0] = t.x; v[it][1] = t.y; v[it][2] = t.z; v[it][3] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (row0 + rq + j < nrows) v[it][j] = p[(int64_t)j * sr];
        }
      }
    }
  }
}

template <int ROWS, int BKT>
__device__ __forceinline__ void tile_store(float* lds, int64_t sk, int tid, const float (&v)[ROWS * BKT / 1024][4]) {
  constexpr int NCH = ROWS * BKT / 1024, PITCH = ROWS + 1;
#pragma unroll
  for (int it = 0; it < NCH; ++it) {
    const int c = it * NT + tid;
    if (sk == 1) {
      const int r = c / (BKT / 4), kq = (c % (BKT / 4)) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) lds[(kq + j) * PITCH + r] = v[it][j];
    } else {
      const int k = c / (ROWS / 4), rq = (c % (ROWS / 4)) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) lds[k * PITCH + rq + j] = v[it][j];
    }
  }
}

// TI x TI MFMA tiles per wave, WM x (4 / WM) waves: workgroup tile 32 TI WM x 32 TI (4 / WM), K-step BKT.
//   <2, 32, 2>: 128 x 128 tiles;  <1, 64, 2>: 64 x 64 tiles for problems with few tiles (the 256 x 256 x 512 logits products
//   of the contrastive head ran on FOUR workgroups and took 93 us each, un-overlapped, between the towers and the backward);
//   <1, 32, 1>: 32 x 128 tiles for M <= 32 (the assignment logits of the semantic-group block: 8 centers x 196 patches x
//   256 samples - on 128-row tiles 15/16 of the MFMA work multiplied zero rows: 194 -> 111 us; K-step 64 instead of 32: 187 us,
//   the 20 KB of LDS per workgroup, i.e. 7 resident workgroups per CU, are what hides the staging latency).
template <int TI, int BKT, int WM>
__global__ __launch_bounds__(NT) void gemm_f32_kernel(GemmArgs g) {
  constexpr int WN = 4 / WM, BM = 32 * TI * WM, BN = 32 * TI * WN, PA = BM + 1, PB = BN + 1;
  constexpr int NCA = BM * BKT / 1024, NCB = BN * BKT / 1024;
  extern __shared__ __attribute__((aligned(16))) float smem_f32[];
  float* As = smem_f32;
  float* Bs = smem_f32 + BKT * PA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int64_t z = blockIdx.z, z1 = z / g.nb2, z2 = z % g.nb2;
  const float* A = g.A + z1 * g.bsA1 + z2 * g.bsA2;
  const float* B = g.B + z1 * g.bsB1 + z2 * g.bsB2;
  const int64_t coff = z1 * g.bsC1 + z2 * g.bsC2;
  const int64_t roff = z1 * g.bsR1 + z2 * g.bsR2;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;

  f32x16 acc[TI][TI];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int li = lane & 31, lk = lane >> 5;
  float ra[NCA][4], rb[NCB][4];
  tile_load<BM, BKT>(A, m0, g.M, 0, g.K, g.sam, g.sak, tid, ra);
  tile_load<BN, BKT>(B, n0, g.N, 0, g.K, g.sbn, g.sbk, tid, rb);
  for (int64_t k0 = 0; k0 < g.K; k0 += BKT) {
    tile_store<BM, BKT>(As, g.sak, tid, ra);
    tile_store<BN, BKT>(Bs, g.sbk, tid, rb);
    __syncthreads();
    if (k0 + BKT < g.K) {   // next K-step's operands travel while this one is multiplied
      tile_load<BM, BKT>(A, m0, g.M, k0 + BKT, g.K, g.sam, g.sak, tid, ra);
      tile_load<BN, BKT>(B, n0, g.N, k0 + BKT, g.K, g.sbn, g.sbk, tid, rb);
    }
    // a short last K-step (K = 8 in the weight-gradient-like product of the assignment logits) multiplies no padding
    const int kmax = g.K - k0 < BKT ? (int)((g.K - k0 + 1) & ~1) : BKT;
#pragma unroll 4
    for (int kk = 0; kk < kmax; kk += 2) {
      float a[TI], b[TI];
#pragma unroll
      for (int i = 0; i < TI; ++i) a[i] = As[(kk + lk) * PA + wm * 32 * TI + i * 32 + li];
#pragma unroll
      for (int j = 0; j < TI; ++j) b[j] = Bs[(kk + lk) * PB + wn * 32 * TI + j * 32 + li];
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue: C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TI; ++j) {
      const int64_t n = n0 + wn * 32 * TI + j * 32 + li;
      if (n >= g.N) continue;
      const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 32 * TI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
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
  // M <= 32: 32 x 128 tiles.  Otherwise, with fewer than half a chip of 128 x 128 tiles: 64 x 64 tiles (4x the workgroups),
  // K-step 64.  256 x 256 x 512: 94 -> 39 us (K-step 128: 47 us - the time is the LDS staging and the one-accumulator MFMA
  // chain of a wave, not the round trips)
  static const int force_small = [] { const char* e = segclip_tuning_env("SEGCLIP_GEMM_F32_SMALL"); return e ? atoi(e) : -1; }();
  const bool skinny = force_small != 0 && d->M <= 32;
  const bool small = !skinny && (force_small >= 0 ? force_small != 0 : cdiv(d->N, 128) * cdiv(d->M, 128) * nb < 128);
  const int64_t bm = skinny ? 32 : (small ? 64 : 128), bn = skinny ? 128 : (small ? 64 : 128);
  dim3 grid((unsigned)cdiv(d->N, bn), (unsigned)cdiv(d->M, bm), (unsigned)nb);
  SEGCLIP_REQUIRE(grid.y <= 65535 && nb <= 65535, "gemm_f32: grid too large (M=%lld batch=%lld)",
                  (long long)d->M, (long long)nb);
  constexpr size_t lds_small = 2 * 64 * 65 * sizeof(float), lds_big = 2 * 32 * 129 * sizeof(float),
                   lds_skinny = 32 * (33 + 129) * sizeof(float);
  if (skinny) hipLaunchKernelGGL((gemm_f32_kernel<1, 32, 1>), grid, dim3(NT), lds_skinny, stream, g);
  else if (small) hipLaunchKernelGGL((gemm_f32_kernel<1, 64, 2>), grid, dim3(NT), lds_small, stream, g);
  else hipLaunchKernelGGL((gemm_f32_kernel<2, 32, 2>), grid, dim3(NT), lds_big, stream, g);
  SEGCLIP_CHECK_LAUNCH("gemm_f32");
  return 0;
}
