// bf16 MFMA GEMM (v_mfma_f32_32x32x16_bf16, fp32 accumulate) with fused epilogue.
// C[z](m,n) = epi(alpha * sum_k A(m,k) * B(n,k)).
//
// One kernel template covers the three layouts of a training step:
//   forward  Y = X W^T      : A k-contiguous, B k-contiguous          <A_KS=0, B_KS=0>
//   dgrad    dX = dY W      : A k-contiguous, B k-strided (W[n][k])   <A_KS=0, B_KS=1>
//   wgrad    dW = dY^T X    : A k-strided,   B k-strided             <A_KS=1, B_KS=1>
// k-contiguous operands are staged as a [128 rows][64 k] LDS image (16-B chunk index XOR
// ((row>>1)&7): conflict-free ds_read_b128 for the 32x32x16 fragment).  k-strided operands are
// staged as [64 k][128 rows] (pitch 160) and the fragment is built with the gfx950 LDS
// transpose read ds_read_b64_tr_b16, so neither W nor the activations are ever transposed in HBM.
// The A operand may be fp32 in HBM (residual-stream gradients): it is rounded to bf16 while staging.
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles (64 acc VGPRs).
// Register-staged double buffer: the 8 global loads of tile t+1 are issued before the MFMAs of tile t
// and stay in flight under them; conversion / zero-masking / ds_write happen after the MFMAs (one
// barrier per K-step).  All loads are unconditional from clamped addresses (no divergent branch around a
// load).  Workgroup ids are remapped so each XCD walks a contiguous run of tiles (A panel reuse in its L2).
// Split-K (grid.y) for the wgrad shapes, combined by a deterministic slab reduction.
#include <stdlib.h>

#include "gemm_bf16_common.h"
#include "reduce_rows.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NT = 256;
constexpr int KS_PITCH = 160;                 // elements per k-row of a k-strided image (320 B)
constexpr int SZ_DIRECT = BM * BK * 2;        // 16384 B
constexpr int SZ_KS = BK * KS_PITCH * 2;      // 20480 B

// raw staged data of one operand tile: 4 chunks of 8 elements per thread (+ validity bits)
template <typename T> struct Stage;
template <> struct Stage<bf16_t> { u32x4 v[4]; unsigned ok; };
template <> struct Stage<float> { f32x4 lo[4], hi[4]; unsigned ok; };

template <typename T> __device__ __forceinline__ void ld8(Stage<T>& s, int i, const T* p);
template <> __device__ __forceinline__ void ld8<bf16_t>(Stage<bf16_t>& s, int i, const bf16_t* p) {
  s.v[i] = *reinterpret_cast<const u32x4*>(p);
}
template <> __device__ __forceinline__ void ld8<float>(Stage<float>& s, int i, const float* p) {
  s.lo[i] = *reinterpret_cast<const f32x4*>(p);
  s.hi[i] = *reinterpret_cast<const f32x4*>(p + 4);
}
__device__ __forceinline__ u32x4 chunk(const Stage<bf16_t>& s, int i) {
  return ((s.ok >> i) & 1u) ? s.v[i] : u32x4{0u, 0u, 0u, 0u};
}
__device__ __forceinline__ u32x4 chunk(const Stage<float>& s, int i) {
  u32x4 v;
  v[0] = pack2bf(s.lo[i][0], s.lo[i][1]); v[1] = pack2bf(s.lo[i][2], s.lo[i][3]);
  v[2] = pack2bf(s.hi[i][0], s.hi[i][1]); v[3] = pack2bf(s.hi[i][2], s.hi[i][3]);
  return ((s.ok >> i) & 1u) ? v : u32x4{0u, 0u, 0u, 0u};
}

// ---- global -> registers (FAST: 16-byte aligned rows, branch-free) ----------------------------
// k-contiguous operand: X(row,k) = X[row*ld + k]; thread -> chunk c = tid&7 (8 k), rows tid>>3 + 32*i
template <typename T>
__device__ __forceinline__ void gload_direct_fast(Stage<T>& s, const T* __restrict__ X, int64_t ld, int64_t row0,
                                                  int64_t nrows, int64_t k0, int64_t K, int tid) {
  const int64_t k = k0 + (tid & 7) * 8;
  const bool kok = k < K;
  const int64_t kc = kok ? k : K - 8;
  unsigned ok = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = row0 + (tid >> 3) + 32 * i;
    ok |= (unsigned)(kok && r < nrows) << i;
    ld8<T>(s, i, X + (r < nrows ? r : nrows - 1) * ld + kc);
  }
  s.ok = ok;
}
// k-strided operand: X(row,k) = X[k*ld + row]; thread -> chunk c = tid&15 (8 rows), k = tid>>4 + 16*i
template <typename T>
__device__ __forceinline__ void gload_ks_fast(Stage<T>& s, const T* __restrict__ X, int64_t ld, int64_t row0,
                                              int64_t nrows, int64_t k0, int64_t K, int tid) {
  const int64_t r = row0 + (tid & 15) * 8;
  const bool rok = r < nrows;
  const int64_t rc = rok ? r : nrows - 8;
  unsigned ok = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t k = k0 + (tid >> 4) + 16 * i;
    ok |= (unsigned)(rok && k < K) << i;
    ld8<T>(s, i, X + (k < K ? k : K - 1) * ld + rc);
  }
  s.ok = ok;
}

// ---- global -> registers (generic: any alignment / ragged K; element-wise guarded) -------------
template <typename T> __device__ __forceinline__ void put8(Stage<T>& s, int i, const float* e);
template <> __device__ __forceinline__ void put8<bf16_t>(Stage<bf16_t>& s, int i, const float* e) {
#pragma unroll
  for (int j = 0; j < 4; ++j) s.v[i][j] = pack2bf(e[2 * j], e[2 * j + 1]);
}
template <> __device__ __forceinline__ void put8<float>(Stage<float>& s, int i, const float* e) {
#pragma unroll
  for (int j = 0; j < 4; ++j) { s.lo[i][j] = e[j]; s.hi[i][j] = e[4 + j]; }
}
template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }

template <typename T>
__device__ __forceinline__ void gload_direct_slow(Stage<T>& s, const T* __restrict__ X, int64_t ld, int64_t row0,
                                                  int64_t nrows, int64_t k0, int64_t K, int tid) {
  const int64_t k = k0 + (tid & 7) * 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = row0 + (tid >> 3) + 32 * i;
    float e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = (r < nrows && k + j < K) ? ldf<T>(X + r * ld + k + j) : 0.f;
    put8<T>(s, i, e);
  }
  s.ok = 0xF;
}
template <typename T>
__device__ __forceinline__ void gload_ks_slow(Stage<T>& s, const T* __restrict__ X, int64_t ld, int64_t row0,
                                              int64_t nrows, int64_t k0, int64_t K, int tid) {
  const int64_t r = row0 + (tid & 15) * 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t k = k0 + (tid >> 4) + 16 * i;
    float e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = (k < K && r + j < nrows) ? ldf<T>(X + k * ld + r + j) : 0.f;
    put8<T>(s, i, e);
  }
  s.ok = 0xF;
}

// ---- registers -> LDS -----------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void sstore_direct(const Stage<T>& s, char* lds, int tid) {
  const int c = tid & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (tid >> 3) + 32 * i;
    *reinterpret_cast<u32x4*>(lds + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = chunk(s, i);
  }
}
template <typename T>
__device__ __forceinline__ void sstore_ks(const Stage<T>& s, char* lds, int tid) {
  const int c = tid & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = (tid >> 4) + 16 * i;
    *reinterpret_cast<u32x4*>(lds + k * (KS_PITCH * 2) + c * 16) = chunk(s, i);
  }
}

// ---- LDS -> MFMA fragments ------------------------------------------------------------------
// fragment of the 32-row sub-tile starting at `rbase`, k16-chunk kc: lane l -> row l&31, k = 8*(l>>5)+0..7
__device__ __forceinline__ bf16x8_t frag_direct(const char* lds, int rbase, int kc, int lane) {
  const int r = rbase + (lane & 31);
  const int c = kc * 2 + (lane >> 5);
  return *reinterpret_cast<const bf16x8_t*>(lds + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
}
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__device__ __forceinline__ bf16x8_t frag_ks(const char* lds, int rbase, int kc, int lane) {
  // ds_read_b64_tr_b16: within each 16-lane group, lane q supplies the address of 4 contiguous
  // b16 of k-row (q>>2); lane q receives column q of the 4x16 block -> 4 consecutive k for its row.
  const int g4 = lane >> 4, q = lane & 15;
  const int krow = kc * 16 + 8 * (g4 >> 1) + (q >> 2);
  const int col = rbase + 16 * (g4 & 1) + 4 * (q & 3);
  const char* p = lds + krow * (KS_PITCH * 2) + col * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * (KS_PITCH * 2)));
  s16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, r);
}

template <bool A_KS, bool B_KS, bool A_F32, bool FAST>
__global__ __launch_bounds__(NT) void gemm_bf16_kernel(Args g) {
  constexpr int SZA = A_KS ? SZ_KS : SZ_DIRECT;
  constexpr int SZB = B_KS ? SZ_KS : SZ_DIRECT;
  __shared__ __attribute__((aligned(16))) char smem[2 * (SZA + SZB)];
  using TA = typename std::conditional<A_F32, float, bf16_t>::type;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware bijective remap of the workgroup id (block b runs on XCD b%8)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int64_t n0 = (int64_t)(wg % g.nbx) * BN, m0 = (int64_t)(wg / g.nbx) * BM;
  const int64_t z = blockIdx.z, z1 = z / g.nb2, z2 = z % g.nb2;
  const TA* A = reinterpret_cast<const TA*>(g.A) + z1 * g.bsA1 + z2 * g.bsA2;
  const bf16_t* B = g.B + z1 * g.bsB1 + z2 * g.bsB2;
  const int64_t coff = z1 * g.bsC1 + z2 * g.bsC2;
  const int64_t roff = z1 * g.bsR1 + z2 * g.bsR2;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int64_t kbeg = (int64_t)blockIdx.y * g.kper;
  const int64_t kend = kbeg + g.kper < g.K ? kbeg + g.kper : g.K;
  const int64_t nk = (kend - kbeg + BK - 1) / BK;

  Stage<TA> sa;
  Stage<bf16_t> sb;
  auto gload = [&](int64_t k0) {
    if constexpr (FAST) {
      if constexpr (A_KS) gload_ks_fast<TA>(sa, A, g.lda, m0, g.M, k0, kend, tid);
      else gload_direct_fast<TA>(sa, A, g.lda, m0, g.M, k0, kend, tid);
      if constexpr (B_KS) gload_ks_fast<bf16_t>(sb, B, g.ldb, n0, g.N, k0, kend, tid);
      else gload_direct_fast<bf16_t>(sb, B, g.ldb, n0, g.N, k0, kend, tid);
    } else {
      if constexpr (A_KS) gload_ks_slow<TA>(sa, A, g.lda, m0, g.M, k0, kend, tid);
      else gload_direct_slow<TA>(sa, A, g.lda, m0, g.M, k0, kend, tid);
      if constexpr (B_KS) gload_ks_slow<bf16_t>(sb, B, g.ldb, n0, g.N, k0, kend, tid);
      else gload_direct_slow<bf16_t>(sb, B, g.ldb, n0, g.N, k0, kend, tid);
    }
  };
  auto sstore = [&](int buf) {
    char* la = smem + buf * (SZA + SZB);
    char* lb = la + SZA;
    if constexpr (A_KS) sstore_ks(sa, la, tid); else sstore_direct(sa, la, tid);
    if constexpr (B_KS) sstore_ks(sb, lb, tid); else sstore_direct(sb, lb, tid);
  };

  gload(kbeg);
  sstore(0);
  __syncthreads();
  for (int64_t t = 0; t < nk; ++t) {
    const int buf = (int)(t & 1);
    if (t + 1 < nk) gload(kbeg + (t + 1) * BK);
    const char* la = smem + buf * (SZA + SZB);
    const char* lb = la + SZA;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      bf16x8_t a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = A_KS ? frag_ks(la, wm * 64 + i * 32, kc, lane) : frag_direct(la, wm * 64 + i * 32, kc, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[j] = B_KS ? frag_ks(lb, wn * 64 + j * 32, kc, lane) : frag_direct(lb, wn * 64 + j * 32, kc, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nk) sstore(buf ^ 1);
    __syncthreads();
  }

  // C/D map of a 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (g.splits > 1) {
    const int li = lane & 31, lk = lane >> 5;
    float* slab = g.slab + ((int64_t)blockIdx.y * gridDim.z + z) * g.M * g.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t n = n0 + wn * 64 + j * 32 + li;
        if (n >= g.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          if (m < g.M) slab[m * g.N + n] = acc[i][j][r];
        }
      }
    return;
  }
  const bool full = m0 + BM <= g.M && n0 + BN <= g.N && ((g.ldc | g.ldaux | coff) & 1) == 0;
  if (g.c_dtype == SEGCLIP_BF16) {
    if (full) epilogue_mode<bf16_t, true>(g, acc, m0, n0, wm, wn, lane, coff, roff);
    else epilogue_mode<bf16_t, false>(g, acc, m0, n0, wm, wn, lane, coff, roff);
  } else {
    if (full) epilogue_mode<float, true>(g, acc, m0, n0, wm, wn, lane, coff, roff);
    else epilogue_mode<float, false>(g, acc, m0, n0, wm, wn, lane, coff, roff);
  }
}

#if GB_PART == 4
// C = alpha * sum_s slab[s]  (split-K combine; deterministic order)
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, void* C, int64_t M, int64_t N, int64_t ldc,
                                     int64_t nb2, int64_t bsC1, int64_t bsC2, int splits, int64_t nz, float alpha,
                                     int c_dtype) {
  const int64_t total = nz * M * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += slab[(int64_t)s * total + i];
    v *= alpha;
    const int64_t z = i / (M * N), rem = i % (M * N), m = rem / N, n = rem % N;
    const int64_t o = (z / nb2) * bsC1 + (z % nb2) * bsC2 + m * ldc + n;
    if (c_dtype == SEGCLIP_BF16) ((bf16_t*)C)[o] = f2bf(v); else ((float*)C)[o] = v;
  }
}

// contiguous fp32/bf16 C (ldc == N, one batch): 16-byte loads, four slabs in flight, no index arithmetic
__global__ __launch_bounds__(256) void splitk_reduce_vec_kernel(const float* __restrict__ slab, void* __restrict__ C,
                                                                int64_t total4, int splits, float alpha, int c_dtype) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const f32x4* p = reinterpret_cast<const f32x4*>(slab) + i;
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
  v *= alpha;
  if (c_dtype == SEGCLIP_BF16)
    reinterpret_cast<u32x2*>(C)[i] = u32x2{pack2bf(v.x, v.y), pack2bf(v.z, v.w)};
  else
    reinterpret_cast<f32x4*>(C)[i] = v;
}

#endif  // GB_PART == 4 (reduction kernels)
}  // namespace

// Compiled once per GB_PART (build.sh), like gemm_bf16_p8.hip: parts 0..3 hold the four kernel instances (fp32 / bf16
// left operand x aligned / ragged) of one operand layout <A_KS = part>>1, B_KS = part&1> behind a launcher; part 4 is
// the split-K reduction kernels and the host-side dispatcher.
#ifndef GB_PART
#error "compile with -DGB_PART=0..4 (see build.sh)"
#endif
#define GB_LAUNCHER(NAME, AK, BKS)                                                                              \
  void NAME(bool a_f32, bool fast, dim3 grid, hipStream_t stream, const void* args) {                           \
    const Args g = *reinterpret_cast<const Args*>(args);                                                        \
    if (a_f32) {                                                                                                \
      if (fast) hipLaunchKernelGGL((gemm_bf16_kernel<AK, BKS, true, true>), grid, dim3(NT), 0, stream, g);      \
      else hipLaunchKernelGGL((gemm_bf16_kernel<AK, BKS, true, false>), grid, dim3(NT), 0, stream, g);          \
    } else {                                                                                                    \
      if (fast) hipLaunchKernelGGL((gemm_bf16_kernel<AK, BKS, false, true>), grid, dim3(NT), 0, stream, g);     \
      else hipLaunchKernelGGL((gemm_bf16_kernel<AK, BKS, false, false>), grid, dim3(NT), 0, stream, g);         \
    }                                                                                                           \
  }
#if GB_PART == 0
GB_LAUNCHER(segclip_gb_launch_ff, false, false)
#elif GB_PART == 1
GB_LAUNCHER(segclip_gb_launch_fk, false, true)
#elif GB_PART == 2
GB_LAUNCHER(segclip_gb_launch_kf, true, false)
#elif GB_PART == 3
GB_LAUNCHER(segclip_gb_launch_kk, true, true)
#endif

#if GB_PART == 4
void segclip_gb_launch_ff(bool, bool, dim3, hipStream_t, const void*);
void segclip_gb_launch_fk(bool, bool, dim3, hipStream_t, const void*);
void segclip_gb_launch_kf(bool, bool, dim3, hipStream_t, const void*);
void segclip_gb_launch_kk(bool, bool, dim3, hipStream_t, const void*);

bool segclip_gemm_bf16_dma_try(const segclip_gemm_desc* d, const void* args, int splits, int64_t kper, int64_t nb,
                               hipStream_t stream);
bool segclip_gemm_bf16_p8_try(const segclip_gemm_desc* d, const void* args, int splits, int64_t kper, int64_t nb,
                              hipStream_t stream);
bool segclip_gemm_bf16_pq_try(const segclip_gemm_desc* d, const void* args, int splits, int64_t nb, hipStream_t stream);
int segclip_gemm_bf16_dma_pick_bn(const segclip_gemm_desc* d, int64_t nbatch_splits);

static bool aligned16(const segclip_gemm_desc* d) {
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool a_ks = d->sak != 1, b_ks = d->sbk != 1;
  const int64_t lda = a_ks ? d->sak : d->sam, ldb = b_ks ? d->sbk : d->sbn;
  const int64_t aes = d->a_dtype == SEGCLIP_F32 ? 4 : 2;
  bool fast = al(d->A) && al(d->B) && lda % 8 == 0 && ldb % 8 == 0 && (d->bsA1 * aes) % 16 == 0 &&
              (d->bsA2 * aes) % 16 == 0 && (d->bsB1 * 2) % 16 == 0 && (d->bsB2 * 2) % 16 == 0;
  fast = fast && (a_ks ? (d->M % 8 == 0 && d->M >= 8) : (d->K % 8 == 0 && d->K >= 8));
  fast = fast && (b_ks ? (d->N % 8 == 0 && d->N >= 8) : (d->K % 8 == 0 && d->K >= 8));
  return fast;
}
// the LDS-DMA kernel (256x128 tiles) takes the large aligned bf16 x bf16 problems
static bool want_dma(const segclip_gemm_desc* d) {
  return d->a_dtype == SEGCLIP_BF16 && d->b_dtype == SEGCLIP_BF16 && aligned16(d) && d->K % 64 == 0 && d->K >= 64 &&
         d->M >= 64 && d->N >= 16;
}
static int choose_splits(const segclip_gemm_desc* d) {
  if (d->bias || d->residual || d->aux || d->act != SEGCLIP_ACT_NONE || d->mul_dact) return 1;
  const int64_t nb = (d->nb1 > 0 ? d->nb1 : 1) * (d->nb2 > 0 ? d->nb2 : 1);
  const int64_t ksteps = cdiv(d->K, BK);
  static const int force_tile = [] { const char* e = segclip_tuning_env("SEGCLIP_GEMM_TILE"); return e ? atoi(e) : 0; }();
  if (want_dma(d) && force_tile == 128) {
    // 128x128 tiles, two workgroups per CU: one round = 512 workgroups
    const int64_t tiles = cdiv(d->M, 128) * cdiv(d->N, 128) * nb;
    if (tiles >= 320 || ksteps < 32) return 1;
    int64_t s = 512 / tiles;
    if (s > ksteps / 8) s = ksteps / 8;
    if (s > 32) s = 32;
    return s < 1 ? 1 : (int)s;
  }
  if (want_dma(d)) {
    // 256-row tiles, long K loops: aim at one full round of the 256 CUs
    const int64_t bn = d->N > 128 ? 256 : 128;
    const int64_t tiles = cdiv(d->M, 256) * cdiv(d->N, bn) * nb;
    if (tiles >= 160 || ksteps < 32) return 1;
    int64_t s = 256 / tiles;
    if (s > ksteps / 8) s = ksteps / 8;
    if (s > 32) s = 32;
    return s < 1 ? 1 : (int)s;
  }
  const int64_t tiles = cdiv(d->M, BM) * cdiv(d->N, BN) * nb;
  if (tiles >= 384 || ksteps < 16) return 1;
  int64_t s = cdiv(768, tiles);
  if (s > ksteps / 4) s = ksteps / 4;
  if (s > 32) s = 32;
  return s < 1 ? 1 : (int)s;
}

size_t segclip_gemm_bf16_pq_tail_ws_bytes(const segclip_gemm_desc* d);
size_t segclip_gemm_bf16_ws_bytes(const segclip_gemm_desc* d) {
  const int s = choose_splits(d);
  if (s <= 1) {
    // no split-K: the tail split of gemm_bf16_pq.hip may want a workspace (bf16 / fp32-residual outputs on full 256-tiles)
    if (want_dma(d) && d->sak == 1 && (d->c_dtype == SEGCLIP_BF16 || (d->c_dtype == SEGCLIP_F32 && d->residual)))
      return segclip_gemm_bf16_pq_tail_ws_bytes(d);
    return 0;
  }
  const int64_t nb = (d->nb1 > 0 ? d->nb1 : 1) * (d->nb2 > 0 ? d->nb2 : 1);
  return (size_t)s * nb * d->M * d->N * sizeof(float);
}

// number of K splits segclip_gemm_bf16_launch() will use for this descriptor (1 = no split-K, nothing in ws)
int segclip_gemm_bf16_splits(const segclip_gemm_desc* d) {
  int splits = choose_splits(d);
  if (splits > 1 && (d->ws == nullptr || (size_t)d->ws_bytes < segclip_gemm_bf16_ws_bytes(d))) splits = 1;
  if (splits <= 1) return 1;
  const int64_t kper = cdiv(cdiv(d->K, BK), splits) * BK;
  return (int)cdiv(d->K, kper);
}

int segclip_gemm_bf16_launch(const segclip_gemm_desc* d, hipStream_t stream) {
  const bool a_ks = d->sak != 1, b_ks = d->sbk != 1;
  SEGCLIP_REQUIRE(!a_ks || d->sam == 1, "gemm_bf16: A needs unit stride along k or m (sam=%lld sak=%lld)",
                  (long long)d->sam, (long long)d->sak);
  SEGCLIP_REQUIRE(!b_ks || d->sbn == 1, "gemm_bf16: B needs unit stride along k or n (sbn=%lld sbk=%lld)",
                  (long long)d->sbn, (long long)d->sbk);
  SEGCLIP_REQUIRE(d->b_dtype == SEGCLIP_BF16, "gemm_bf16: B must be bf16");
  const bool a_f32 = d->a_dtype == SEGCLIP_F32;
  Args g;
  g.A = d->A; g.B = (const bf16_t*)d->B; g.C = d->C; g.bias = d->bias; g.residual = d->residual; g.aux = d->aux;
  g.M = d->M; g.N = d->N; g.K = d->K;
  g.lda = a_ks ? d->sak : d->sam; g.ldb = b_ks ? d->sbk : d->sbn;
  g.ldc = d->ldc; g.ldr = d->ldr; g.ldaux = d->ldaux;
  g.nb2 = d->nb2 > 0 ? d->nb2 : 1;
  g.bsA1 = d->bsA1; g.bsA2 = d->bsA2; g.bsB1 = d->bsB1; g.bsB2 = d->bsB2; g.bsC1 = d->bsC1; g.bsC2 = d->bsC2;
  g.bsR1 = d->bsR1; g.bsR2 = d->bsR2;
  g.c_dtype = d->c_dtype; g.r_dtype = d->r_dtype; g.act = d->act; g.mul_dact = d->mul_dact; g.aux_kind = d->aux_kind; g.alpha = d->alpha;
  g.nbx = (int)cdiv(d->N, BN); g.nby = (int)cdiv(d->M, BM);
  const int64_t nb = (d->nb1 > 0 ? d->nb1 : 1) * g.nb2;
  SEGCLIP_REQUIRE(nb <= 65535, "gemm_bf16: batch too large (%lld)", (long long)nb);
  g.splits = choose_splits(d);
  if (g.splits > 1 && (d->ws == nullptr || (size_t)d->ws_bytes < segclip_gemm_bf16_ws_bytes(d))) g.splits = 1;
  g.kper = g.splits > 1 ? cdiv(cdiv(d->K, BK), g.splits) * BK : d->K;
  if (g.splits > 1) g.splits = (int)cdiv(d->K, g.kper);
  g.slab = (float*)d->ws;
  g.vec_epi = 0;
  g.touch = 0;
  g.abl = 0;
  g.colgroups = 1;
  static const int xw_epi = [] { const char* e = segclip_tuning_env("SEGCLIP_EPI_XW"); return e ? atoi(e) : 2; }();
  g.xw_epi = xw_epi;
  static const int slab_staged = [] { const char* e = segclip_tuning_env("SEGCLIP_P8_SLAB_STAGED"); return e ? atoi(e) : 1; }();
  g.slab_staged = slab_staged;
  g.colsum_part = nullptr;
  dim3 grid((unsigned)(g.nbx * g.nby), (unsigned)g.splits, (unsigned)nb);
  const bool fast = aligned16(d);
  bool launched = false;
  if (d->colsum) {
    SEGCLIP_REQUIRE(d->colsum_ws != nullptr, "gemm: colsum needs colsum_ws");
    if (!(want_dma(d) && d->M % 128 == 0 && d->N % 256 == 0 && g.splits == 1 && nb == 1)) {
      segclip_set_error("gemm: fused colsum unsupported for this shape");
      return SEGCLIP_ERR_UNSUPPORTED;
    }
    g.colsum_part = d->colsum_ws;
  }
  if (d->res_row_mod > 0) {   // only gemm_bf16_pq.hip's fp32-residual epilogue addresses the residual modulo a row count
    launched = want_dma(d) && nb == 1 && g.splits == 1 && segclip_gemm_bf16_pq_try(d, &g, 1, nb, stream);
    if (!launched) {
      segclip_set_error("gemm: res_row_mod needs the fp32-residual epilogue on full 256 x 256 bf16 tiles");
      return SEGCLIP_ERR_UNSUPPORTED;
    }
  } else if (want_dma(d)) {
    static const int force_tile = [] { const char* e = segclip_tuning_env("SEGCLIP_GEMM_TILE"); return e ? atoi(e) : 0; }();
    // 256x256 tiles: the phase-pipelined kernel (gemm_bf16_p8.hip); 256x128 / 128x128 tiles: the one-barrier-per-K-tile
    // kernel (gemm_bf16_dma.hip)
    // 256x256 tiles, bf16 output, full tiles, no split-K: the persistent kernel with the overlapped output path
    // (gemm_bf16_pq.hip); else the phase-pipelined one-tile-per-workgroup kernel (gemm_bf16_p8.hip)
    if (force_tile == 0 && segclip_gemm_bf16_dma_pick_bn(d, nb * g.splits) == 256) {
      launched = segclip_gemm_bf16_pq_try(d, &g, g.splits, nb, stream);
      if (!launched) launched = segclip_gemm_bf16_p8_try(d, &g, g.splits, g.kper, nb, stream);
    }
    if (!launched) launched = segclip_gemm_bf16_dma_try(d, &g, g.splits, g.kper, nb, stream);
  }
  if (d->colsum && (!launched || !g.colsum_part)) {
    segclip_set_error("gemm: fused colsum unsupported for this shape");
    return SEGCLIP_ERR_UNSUPPORTED;
  }
  if (!launched && d->aux_kind == 2 && d->aux) {
    segclip_set_error("gemm: aux_kind 2 needs full 256 x 256 tiles, 16-byte aligned operands and no split-K");
    return SEGCLIP_ERR_UNSUPPORTED;
  }
  if (!launched) {
    if (!a_ks && !b_ks) segclip_gb_launch_ff(a_f32, fast, grid, stream, &g);
    else if (!a_ks && b_ks) segclip_gb_launch_fk(a_f32, fast, grid, stream, &g);
    else if (a_ks && b_ks) segclip_gb_launch_kk(a_f32, fast, grid, stream, &g);
    else segclip_gb_launch_kf(a_f32, fast, grid, stream, &g);
  }
  SEGCLIP_CHECK_LAUNCH("gemm_bf16");
  if (d->colsum && !(d->flags & SEGCLIP_GEMM_DEFER_COLSUM)) {
    launch_reduce_rows((const float*)d->colsum_ws, d->M / 64, d->N, d->N, d->colsum, nullptr, nullptr, d->N, stream);
    SEGCLIP_CHECK_LAUNCH("gemm_colsum_reduce");
  }
  if (g.splits > 1 && !(d->flags & SEGCLIP_GEMM_DEFER_SPLITK)) {
    const int64_t total = nb * d->M * d->N;
    const int blocks = (int)(cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048);
    const int64_t cal = d->c_dtype == SEGCLIP_BF16 ? 7 : 15;
    if (nb == 1 && d->ldc == d->N && total % 4 == 0 && (reinterpret_cast<uintptr_t>(d->C) & cal) == 0 &&
        (reinterpret_cast<uintptr_t>(d->ws) & 15) == 0)
      hipLaunchKernelGGL(splitk_reduce_vec_kernel, dim3((unsigned)cdiv(total / 4, 256)), dim3(256), 0, stream,
                         (const float*)d->ws, d->C, total / 4, g.splits, d->alpha, d->c_dtype);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)d->ws, d->C, d->M,
                       d->N, d->ldc, g.nb2, d->bsC1, d->bsC2, g.splits, nb, d->alpha, d->c_dtype);
    SEGCLIP_CHECK_LAUNCH("splitk_reduce");
  }
  return 0;
}
#endif  // GB_PART == 4
