// bf16 MFMA GEMM, LDS-DMA pipeline (the fast path for the large, aligned shapes of the training step).
// C[z](m,n) = epi(alpha * sum_k A(m,k) * B(n,k)),  A and B bf16, each k-contiguous or k-strided.
//
// Tile 256 x BN x 64 (BN = 256 or 128), 512 threads = 8 waves (two per SIMD):
//   BN=256: waves 2(m) x 4(n), each 128x64 = 4x2 MFMA 32x32x16 accumulators (128 VGPRs)
//   BN=128: waves 4(m) x 2(n), each  64x64 = 2x2
// Operand tiles go L2 -> LDS directly (global_load_lds_dwordx4: 1 KiB per wave instruction, no VGPR round
// trip) into a double buffer (2 x 64 KiB): the DMA of K-tile t+1 is issued right after the barrier that
// publishes tile t and lands under tile t's MFMAs.  Measured on MI355X (tools/ubench/dma_bw.hip): an
// LDS-DMA stream of >=128-byte row pieces sustains ~84 GB/s per CU (21 TB/s chip) from L2, the same
// stream in 64-byte pieces only 40 GB/s - hence BK = 64 (128-byte rows of a k-contiguous operand) and the
// 256x256 tile (64 KiB per 2048 MFMA-cycles keeps the DMA engine at ~75 % while the matrix pipe is full).
// An LDS-DMA write is lane-linear (wave base + lane*16), so the bank-conflict swizzle is applied to the
// per-lane SOURCE address and again on the fragment read:
//   k-contiguous operand: [rows][64 k] 128-B rows, 16-B chunk ^ ((row>>1)&7)  -> conflict-free ds_read_b128
//   k-strided operand   : [64 k][rows] rows*2-B k-rows, 16-B chunk ^ ((k&3)<<2) -> conflict-free
//                         ds_read_b64_tr_b16 (the 4 k-rows of a transpose read land in 4 bank quadrants)
// MFMA operand fragments are register double-buffered (chunk kc+1 is requested before the MFMAs of kc).
// Full tiles leave through the LDS-staged coalesced epilogue (gemm_bf16_common.h).
// Rows beyond M / N are clamped to valid addresses (their products are never stored); K must be a
// multiple of 64 (the host falls back to the register-staged kernel otherwise).
#include <stdlib.h>

#include "gemm_bf16_common.h"

namespace {

constexpr int BK = 64;

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ void dma16(const bf16_t* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)lds_wave_base, 16, 0, 0);
}

// k-contiguous operand with ROWS rows: X(row,k) = X[row*ld + k]; [ROWS][128 B]; ROWS/64 pieces per wave.
template <int ROWS, int NWV>
__device__ __forceinline__ void issue_direct(const bf16_t* __restrict__ X, int64_t ld, int64_t row0, int64_t nrows,
                                             int64_t k0, char* lds, int wave, int lane) {
  constexpr int PER_WAVE = ROWS / (8 * NWV);
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int idx = wave * PER_WAVE + i;        // 1-KiB piece = 8 rows of 128 B
    const int row = idx * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int64_t r = row0 + row;
    r = r < nrows ? r : nrows - 1;
    dma16(X + r * ld + k0 + chunk * 8, lds + idx * 1024);
  }
}
// k-strided operand with ROWS rows: X(row,k) = X[k*ld + row]; LDS image [64 k][ROWS], k-row = ROWS*2 bytes.
template <int ROWS, int NWV>
__device__ __forceinline__ void issue_ks(const bf16_t* __restrict__ X, int64_t ld, int64_t row0, int64_t nrows,
                                         int64_t k0, char* lds, int wave, int lane) {
  constexpr int PER_WAVE = ROWS / (8 * NWV);    // 64 k-rows * ROWS*2 B / 1 KiB / NWV waves
  constexpr int CHUNKS = ROWS / 8;              // 16-B chunks per k-row
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int idx = wave * PER_WAVE + i;
    const int krow = idx * (512 / ROWS) + lane / CHUNKS;
    const int chunk = (lane % CHUNKS) ^ ((krow & 3) << 2);
    int64_t r = row0 + chunk * 8;
    r = r + 8 <= nrows ? r : nrows - 8;
    dma16(X + (k0 + krow) * ld + r, lds + idx * 1024);
  }
}

// lane l -> row rbase + (l&31), k = kc*16 + 8*(l>>5) .. +7
__device__ __forceinline__ bf16x8_t frag_direct(const char* lds, int rbase, int kc, int lane) {
  const int r = rbase + (lane & 31);
  const int c = kc * 2 + (lane >> 5);
  return *reinterpret_cast<const bf16x8_t*>(lds + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
}
template <int ROWS>
__device__ __forceinline__ bf16x8_t frag_ks(const char* lds, int rbase, int kc, int lane) {
  const int g4 = lane >> 4, q = lane & 15;
  const int krow = kc * 16 + 8 * (g4 >> 1) + (q >> 2);
  const int col = rbase + 16 * (g4 & 1) + 4 * (q & 3);
  const char* p = lds + krow * (ROWS * 2) + ((((col >> 3) ^ ((krow & 3) << 2))) << 4) + ((col & 7) << 1);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * (ROWS * 2)));
  s16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, r);
}

template <int N> __device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else static_assert(N < 0, "unsupported vmcnt");
}

// <BM, BN, NWV>: <256,256,8> and <256,128,8> own a CU (139 KiB of LDS: operand ring / epilogue patches);
// <128,128,4> (waves 2x2, 64x64 each) needs 68 KiB, so two workgroups share a CU and one's output phase overlaps
// the other's MFMA phase.
template <bool A_KS, bool B_KS, int BM, int BN, int NWV, int STAGES = 2>
__global__ __launch_bounds__(NWV * 64, NWV == 4 ? 2 : 1) void gemm_bf16_dma_kernel(Args g) {
  constexpr int SZA = BM * BK * 2;
  constexpr int SZB = BN * BK * 2;
  constexpr int SZS = SZA + SZB;
  constexpr int WN = BN / 64;                          // waves along n (64 columns each)
  constexpr int WM = NWV / WN;                         // waves along m
  constexpr int TM = BM / (WM * 32);                   // 32-row MFMA tiles per wave along m
  constexpr int WROWS = TM * 32;                       // rows per wave
  static_assert(TM == 2 || TM == 4, "wave tile must be 64x64 or 128x64");
  constexpr int LDS_BYTES = STAGES * SZS > NWV * EPI_WAVE_BYTES ? STAGES * SZS : NWV * EPI_WAVE_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int64_t n0 = (int64_t)(wg % g.nbx) * BN, m0 = (int64_t)(wg / g.nbx) * BM;
  const int64_t z = blockIdx.z, z1 = z / g.nb2, z2 = z % g.nb2;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A) + z1 * g.bsA1 + z2 * g.bsA2;
  const bf16_t* B = g.B + z1 * g.bsB1 + z2 * g.bsB2;
  const int64_t coff = z1 * g.bsC1 + z2 * g.bsC2;
  const int64_t roff = z1 * g.bsR1 + z2 * g.bsR2;

  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int64_t kbeg = (int64_t)blockIdx.y * g.kper;
  const int64_t kend = kbeg + g.kper < g.K ? kbeg + g.kper : g.K;
  const int nk = (int)((kend - kbeg) / BK);

  auto issue = [&](int t) {
    char* la = smem + (t % STAGES) * SZS;
    char* lb = la + SZA;
    const int64_t k0 = kbeg + (int64_t)t * BK;
    if constexpr (A_KS) issue_ks<BM, NWV>(A, g.lda, m0, g.M, k0, la, wave, lane);
    else issue_direct<BM, NWV>(A, g.lda, m0, g.M, k0, la, wave, lane);
    if constexpr (B_KS) issue_ks<BN, NWV>(B, g.ldb, n0, g.N, k0, lb, wave, lane);
    else issue_direct<BN, NWV>(B, g.ldb, n0, g.N, k0, lb, wave, lane);
  };

  // The first round of workgroups (one per CU) starts with a bounded, staggered delay: tiles all take the same
  // time, so without it every CU reaches its store phase at the same moment and the HBM write burst (not
  // overlapped with any MFMA: one workgroup per CU) is paid in full by every tile round (+5..10 % measured).
  if (NWV == 8 && bid < 256 && gridDim.x * gridDim.y * gridDim.z > 256) {
    const long long t_tile = (long long)nk * 4000 + 20000;
    const long long unit = t_tile / 8 < 5000 ? t_tile / 8 : 5000;
    const long long wait = ((bid >> 3) & 7) * unit;
    const long long t0 = clock64();
    while (clock64() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }
  issue(0);
  if constexpr (STAGES == 3) {
    if (nk > 1) issue(1);
  }
  constexpr int DMA_PER_STAGE = (BM + BN) / (8 * NWV);  // DMA instructions per wave and k-tile
  for (int t = 0; t < nk; ++t) {
    // this wave's share of tile t has landed (with three stages the DMA of tile t+1 may still be in flight)
    if (STAGES == 3 && t + 1 < nk) wait_vm<DMA_PER_STAGE>();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // ... and everybody else's; buffer (t-1)%STAGES is free
    __builtin_amdgcn_sched_barrier(0);
    if (t + STAGES - 1 < nk) issue(t + STAGES - 1);
    const char* la = smem + (t % STAGES) * SZS;
    const char* lb = la + SZA;
    // fragments of k16-chunk kc+1 are requested before the MFMAs of chunk kc (register double buffer)
    bf16x8_t fa[2][TM], fb[2][2];
    auto ldfrag = [&](int kc, bf16x8_t (&a)[TM], bf16x8_t (&b)[2]) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = A_KS ? frag_ks<BM>(la, wm * WROWS + i * 32, kc, lane) : frag_direct(la, wm * WROWS + i * 32, kc, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[j] = B_KS ? frag_ks<BN>(lb, wn * 64 + j * 32, kc, lane) : frag_direct(lb, wn * 64 + j * 32, kc, lane);
    };
    ldfrag(0, fa[0], fb[0]);
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      // hipcc waits with lgkmcnt(0) at the first MFMA that uses a fragment, i.e. also for whatever was requested
      // after it.  Consuming chunk kc's registers here puts that wait BEFORE the request of chunk kc+1, whose LDS
      // latency is then covered by the MFMAs of chunk kc instead of being exposed.
#pragma unroll
      for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(__builtin_bit_cast(u32x4, fa[kc & 1][i])));
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(__builtin_bit_cast(u32x4, fb[kc & 1][j])));
      if (kc < 3) ldfrag(kc + 1, fa[(kc + 1) & 1], fb[(kc + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch above the MFMAs (else hipcc re-serialises it)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kc & 1][i], fb[kc & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  const int64_t mw = m0 + wm * WROWS;  // this wave's WROWS x 64 sub-tile
  const int64_t nw = n0 + wn * 64;
  if (g.splits > 1) {
    const int li = lane & 31, lk = lane >> 5;
    float* slab = g.slab + ((int64_t)blockIdx.y * gridDim.z + z) * g.M * g.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t n = nw + j * 32 + li;
        if (n >= g.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t m = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          if (m < g.M) __builtin_nontemporal_store(acc[i][j][r], &slab[m * g.N + n]);
        }
      }
    return;
  }
  const bool full = m0 + BM <= g.M && n0 + BN <= g.N;  // uniform over the workgroup
#ifdef SEGCLIP_NO_EPILOGUE
  if (acc[0][0][0] != 12345.678f) return;
#endif
  const bool vec = full && g.vec_epi;
  if (vec) __syncthreads();  // every wave is done with the operand ring: the LDS is reused as 8 private patches
  float* t = reinterpret_cast<float*>(smem + wave * EPI_WAVE_BYTES);
#define RUN_EPI(SUB, MROW)                                                                              \
  do {                                                                                                  \
    if (vec) {                                                                                          \
      if (g.c_dtype == SEGCLIP_BF16) epilogue_lds_mode<bf16_t>(g, SUB, t, MROW, nw, lane, coff, roff);  \
      else epilogue_lds_mode<float>(g, SUB, t, MROW, nw, lane, coff, roff);                             \
    } else {                                                                                            \
      if (g.c_dtype == SEGCLIP_BF16) epilogue_mode<bf16_t, false>(g, SUB, MROW, nw, 0, 0, lane, coff, roff); \
      else epilogue_mode<float, false>(g, SUB, MROW, nw, 0, 0, lane, coff, roff);                       \
    }                                                                                                   \
  } while (0)
  if constexpr (TM == 2) {
    RUN_EPI(acc, mw);
  } else {
    f32x16 lo[2][2] = {{acc[0][0], acc[0][1]}, {acc[1][0], acc[1][1]}};
    RUN_EPI(lo, mw);
    __builtin_amdgcn_wave_barrier();
    f32x16 hi[2][2] = {{acc[2][0], acc[2][1]}, {acc[3][0], acc[3][1]}};
    RUN_EPI(hi, mw + 64);
  }
#undef RUN_EPI
}

}  // namespace

// Compiled once per DMA_PART (build.sh), like gemm_bf16_p8.hip: parts 0..3 hold the kernel instances of one operand
// layout <A_KS = part>>1, B_KS = part&1> behind a launcher, part 4 the host-side dispatcher.
#ifndef DMA_PART
#error "compile with -DDMA_PART=0..4 (see build.sh)"
#endif
// variant: 0 = 256x128 tile, 1 = 256x256 tile, 2 = 128x128 tile / 4 waves, 3 = 256x128 tile with a 3-stage ring (2, 3:
// only with -DSEGCLIP_GEMM_EXPERIMENTS)
#ifdef SEGCLIP_GEMM_EXPERIMENTS
#define DMA_EXP(AK, BKS)                                                                                        \
  if (variant == 2) hipLaunchKernelGGL((gemm_bf16_dma_kernel<AK, BKS, 128, 128, 4>), grid, dim3(256), 0, stream, g);      \
  else if (variant == 3) hipLaunchKernelGGL((gemm_bf16_dma_kernel<AK, BKS, 256, 128, 8, 3>), grid, dim3(512), 0, stream, g); \
  else
#else   // production: the 128x128 / 4-wave tile for problems too small to fill the chip with 256-row tiles (round 4)
#define DMA_EXP(AK, BKS)                                                                                        \
  if (variant == 2) hipLaunchKernelGGL((gemm_bf16_dma_kernel<AK, BKS, 128, 128, 4>), grid, dim3(256), 0, stream, g);      \
  else
#endif
#define DMA_LAUNCHER(NAME, AK, BKS)                                                                             \
  void NAME(int variant, dim3 grid, hipStream_t stream, const void* args) {                                     \
    const Args g = *reinterpret_cast<const Args*>(args);                                                        \
    DMA_EXP(AK, BKS)                                                                                            \
    if (variant == 1) hipLaunchKernelGGL((gemm_bf16_dma_kernel<AK, BKS, 256, 256, 8>), grid, dim3(512), 0, stream, g); \
    else hipLaunchKernelGGL((gemm_bf16_dma_kernel<AK, BKS, 256, 128, 8>), grid, dim3(512), 0, stream, g);       \
  }
#if DMA_PART == 0
DMA_LAUNCHER(segclip_dma_launch_ff, false, false)
#elif DMA_PART == 1
DMA_LAUNCHER(segclip_dma_launch_fk, false, true)
#elif DMA_PART == 2
DMA_LAUNCHER(segclip_dma_launch_kf, true, false)
#elif DMA_PART == 3
DMA_LAUNCHER(segclip_dma_launch_kk, true, true)
#endif

#if DMA_PART == 4
void segclip_dma_launch_ff(int, dim3, hipStream_t, const void*);
void segclip_dma_launch_fk(int, dim3, hipStream_t, const void*);
void segclip_dma_launch_kf(int, dim3, hipStream_t, const void*);
void segclip_dma_launch_kk(int, dim3, hipStream_t, const void*);

// tile-count heuristic: the 256x256 tile unless it leaves the 256 CUs badly quantised and 256x128 does not
static int pick_bn(const segclip_gemm_desc* d, int64_t nbatch_splits) {
  constexpr int BM = 256;
  if (d->N <= 128) return 128;
  auto eff = [&](int bn) {
    const double tiles = (double)cdiv(d->M, BM) * cdiv(d->N, bn) * nbatch_splits;
    const double rounds = tiles / 256.0;
    const double q = rounds / (double)(int64_t)(rounds + 0.999999);  // quantisation efficiency
    const double pad = (double)d->N / (cdiv(d->N, bn) * bn);
    return q * pad * (bn == 256 ? 1.0 : 0.72);                       // 256x128 runs at ~0.72 of the 256x256 rate
  };
  return eff(256) >= eff(128) ? 256 : 128;
}

// tile width the dispatcher would use (the phase-pipelined 256x256 kernel of gemm_bf16_p8.hip takes the 256-wide cases)
int segclip_gemm_bf16_dma_pick_bn(const segclip_gemm_desc* d, int64_t nbatch_splits) { return pick_bn(d, nbatch_splits); }

// Launch the LDS-DMA kernel.  `args_` is prepared by the caller (gemm_bf16.hip); returns false when the shape
// does not meet this kernel's preconditions (the caller then uses the register-staged kernel).
bool segclip_gemm_bf16_dma_try(const segclip_gemm_desc* d, const void* args_, int splits, int64_t kper, int64_t nb,
                               hipStream_t stream) {
  Args g = *reinterpret_cast<const Args*>(args_);
  const bool a_ks = d->sak != 1, b_ks = d->sbk != 1;
  if (d->a_dtype != SEGCLIP_BF16 || d->b_dtype != SEGCLIP_BF16) return false;
  if (d->K % BK != 0 || kper % BK != 0 || d->K < BK) return false;
  if (a_ks && (d->M % 8 != 0 || d->M < 8)) return false;
  if (b_ks && (d->N % 8 != 0 || d->N < 8)) return false;
  if (d->M < 64 || d->N < 16) return false;  // tiny problems: the 128x128 kernel wastes less
  // 128x128 tiles, 4 waves, 68 KiB of LDS -> two workgroups per CU: the per-tile prologue / output phase of one
  // overlaps the MFMA phase of the other.  In isolation (tools/bench_gemm.py, MI355X) this wins 10-25 % on the
  // forward GEMMs with fewer than 4 rounds of 256x256 tiles (N=768 of the vision tower, the whole text tower) and
  // loses 5-15 % on long-K dgrads and every split-K wgrad; inside the training step (text tower concurrent on a
  // second stream) a shape-based choice measured 58.4 vs 58.2 ms, i.e. no gain, so the 256-wide tiles stay the
  // default and SEGCLIP_GEMM_TILE=128 selects this variant for experiments.
  // Round 4: problems whose 256-row tiles cannot fill the 256 CUs (the center stage's q-side linears: M = 8 B = 2048 rows,
  // 48-192 tiles) take the 128x128 tile: four times the workgroups, two per CU, half the K-loop time per tile
  // (SEGCLIP_GEMM_SMALL_TILES=0 switches the rule off; a forward + backward pass of the center stage: see DESIGN 4.4).
  static const int small_rule = [] { const char* e = segclip_tuning_env("SEGCLIP_GEMM_SMALL_TILES"); return e ? atoi(e) : 1; }();
  const int bn_big = pick_bn(d, nb * splits);
  const bool auto_small = small_rule && d->M >= 128 && d->N >= 128 && !(d->aux_kind == 2 && d->aux) && !g.colsum_part &&
                          cdiv(d->M, 256) * cdiv(d->N, bn_big) * nb * splits < 256;
#ifdef SEGCLIP_GEMM_EXPERIMENTS  // build.sh -DSEGCLIP_GEMM_EXPERIMENTS: the 3-stage-ring instances (+ hipcc time)
  static const int force_tile = [] { const char* e = segclip_tuning_env("SEGCLIP_GEMM_TILE"); return e ? atoi(e) : 0; }();
  const bool small = force_tile == 128 || (force_tile == 0 && auto_small);
  const bool three = force_tile == 3;  // experiment: 256x128 tiles with a 3-stage ring (96 KiB in flight)
#else
  const bool small = auto_small;
  constexpr bool three = false;
#endif
  const int bn = (small || three) ? 128 : bn_big;
  const int bm = small ? 128 : 256;
  g.nbx = (int)cdiv(d->N, bn);
  g.nby = (int)cdiv(d->M, bm);
  g.splits = splits;
  g.kper = kper;
  if (g.colsum_part && bn != 256 && d->N % 128 != 0) return false;
  {
    auto al = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const int64_t ce = d->c_dtype == SEGCLIP_BF16 ? 8 : 4, re = d->r_dtype == SEGCLIP_BF16 ? 8 : 4;
    g.vec_epi = al(d->C) && al(d->aux) && al(d->residual) && al(d->bias) && d->ldc % ce == 0 &&
                (!d->aux || d->ldaux % ce == 0) && (!d->residual || d->ldr % re == 0) && d->bsC1 % ce == 0 &&
                d->bsC2 % ce == 0 && (!d->residual || (d->bsR1 % re == 0 && d->bsR2 % re == 0));
    if (d->residual && d->r_dtype != d->c_dtype) g.vec_epi = 0;   // the LDS epilogue holds side operands in the output type
    if (g.colsum_part && !g.vec_epi) return false;
    // the one-byte derivative of the erf-GELU is produced by gemm_bf16_pq.hip only (this epilogue: QuickGELU); decoding is generic
    if (d->aux_kind == 2 && d->aux && !d->mul_dact && d->act != SEGCLIP_ACT_QUICK_GELU) return false;
    // aux_kind 2 (one byte per saved derivative) exists in the staged epilogue of FULL tiles only (8-byte aligned rows)
    if (d->aux_kind == 2 && d->aux &&
        !(g.vec_epi && d->M % 256 == 0 && d->N % 256 == 0 && d->ldaux % 8 == 0 && d->c_dtype == SEGCLIP_BF16 && splits == 1))
      return false;
  }
  dim3 grid((unsigned)(g.nbx * g.nby), (unsigned)splits, (unsigned)nb);
  const int variant = small ? 2 : three ? 3 : bn == 256 ? 1 : 0;
  if (!a_ks && !b_ks) segclip_dma_launch_ff(variant, grid, stream, &g);
  else if (!a_ks && b_ks) segclip_dma_launch_fk(variant, grid, stream, &g);
  else if (a_ks && b_ks) segclip_dma_launch_kk(variant, grid, stream, &g);
  else segclip_dma_launch_kf(variant, grid, stream, &g);
  return true;
}
#endif  // DMA_PART == 4
