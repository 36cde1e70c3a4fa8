// bf16 MFMA GEMM, 256x256x64 tile, phase-pipelined LDS-DMA ring (the fast path for the large aligned shapes of the
// training step; same contract as gemm_bf16_dma.hip: C[z](m,n) = epi(alpha * sum_k A(m,k) * B(n,k)), A and B bf16, each
// k-contiguous or k-strided, K a multiple of 64).
//
// What bounds the older kernel (gemm_bf16_dma.hip, one barrier per K-tile): the 64 KiB DMA burst of K-tile t+1 is issued
// after barrier t and must have landed by barrier t+1, so a K-step lasts one loaded L2->LDS round trip (1.65 us measured
// against 1.0 us of MFMA time) and the queue drains every step.  Here the tile's operands are four 16-KiB HALF-TILES
//      A0 = rows 0-127, A1 = rows 128-255, B0 = columns 0-127, B1 = columns 128-255     (each [128][64 k])
// and a wave (wr = wave/4, wc = wave%4) owns a 2x2 set of 64x32 blocks: rows {h*128 + wr*64 ..+63}, columns
// {h*128 + wc*32 ..+31}.  A K-tile is four PHASES, one C quadrant each - (A0,B0) (A0,B1) (A1,B1) (A1,B0) - so every
// half-tile is read from LDS in exactly ONE phase (A0,B0: phase 1, B1: phase 2, A1: phase 3; phase 4 re-uses registers)
// and its slot can be refilled two phases later.  A phase is an R segment (LDS fragment reads, the DMA issue of one
// half-tile = 2 instructions per wave, one counted vmcnt wait) and an M segment (8 MFMAs, s_setprio 1) separated by
// barriers.  The two wave groups (wr = 0 / 1; one wave of each per SIMD) run half a phase apart: while one group is in
// M the other is in R, so the matrix pipe of every SIMD is fed alternately by its two waves and the issue cost of the
// loads (measured: ~16 cycles per ds_read_b128, ~80 per LDS-DMA piece; an in-order wave cannot hide them in its own
// MFMA gaps - a variant with the loads inside M ran 18 % slower) is paid under the partner's MFMAs.  For that the R
// segments must all be shorter than an M segment (256 cycles), i.e. the load work must be spread evenly:
//      R1: read A0(t)   (8) + DMA A1(t+1)        M1: quadrant (A0,B0)
//      R2: read B1(t)   (4) + DMA B0(t+2)        M2: quadrant (A0,B1)
//      R3: read A1(t)   (8) + DMA A0(t+2)        M3: quadrant (A1,B1)
//      R4: read B0(t+1) (4) + DMA B1(t+2)        M4: quadrant (A1,B0)      (the two B register sets alternate per tile)
// Every half-tile is issued 5 phases ahead of the R segment that reads it, at the first moment its slot is free, and is
// retired by a COUNTED s_waitcnt vmcnt(10) one phase before that read, so 80 KiB stay in flight across the barriers at
// all times (2 buffers x 4 half-tiles = 128 KiB of LDS, as in the older kernel).
// Ordering rules (MI355X guide, LDS-DMA): data is read one phase after the vmcnt that retires it (wait in R(q), every
// wave passes a barrier, read in R(q+1)); a slot is refilled >= 2 phases after its last read (the reads' lgkmcnt(0) sits
// after the barrier that follows them).
#include <stdlib.h>

#include <type_traits>

#include "gemm_bf16_common.h"

namespace {

constexpr int BK = 64, BT = 256, NWV = 8;
constexpr int UNIT = 128 * BK * 2;            // one half-tile: 16 KiB
constexpr int BUF = 4 * UNIT;                 // A0 A1 B0 B1
constexpr int RING = 2 * BUF;                 // 128 KiB

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// Two LDS-DMA pieces (1 KiB each: lane l writes 16 B at lds + l*16) of one half-tile, issued from INLINE ASM on purpose:
// hipcc (ROCm 7.2) tracks an outstanding __builtin_amdgcn_global_load_lds as a pending LDS write and puts
// `s_waitcnt vmcnt(0)` in front of the next ds_read_b64_tr_b16 (the transpose-read builtin carries no alias
// information), which drains the whole ring every phase - the k-strided variants of gemm_bf16_dma.hip have exactly that
// wait in every K-step.  Hidden from the compiler, the DMA is ordered only by the counted vmcnt waits and barriers below.
//   s_nop 4: SGPR base written by SALU -> read by VMEM; s_nop 0: M0 written by SALU -> read by the LDS-DMA.
// M0 is not used by anything else in this kernel (no other LDS-DMA / movrel / GWS), so it is not preserved.
__device__ __forceinline__ void dma16x2(const char* base_uniform, uint32_t off0, uint32_t off1, uint32_t lds0) {
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %2\n\t"
      "s_add_u32 m0, %3, 0x400\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2"
      :
      : "v"(off0), "v"(off1), "s"(base_uniform), "s"(lds0)
      : "memory", "scc");
}

// Per-lane byte offset (relative to the tile's first element of the operand) of DMA piece `idx` (1 KiB) of half-tile h.
// k-contiguous operand X(row,k) = X[row*ld + k]: image [128 rows][128 B], piece = 8 rows, 16-B chunk ^ ((row>>1)&7).
__device__ __forceinline__ uint32_t off_direct(int h, int idx, int lane, int64_t ld, int64_t row0, int64_t nrows, int rmask = 0) {
  const int u = idx * 8 + (lane >> 3);
  const int chunk = (lane & 7) ^ ((u >> 1) & 7);
  int64_t r = row0 + h * 128 + (u & ~rmask);   // rmask: experiment (fewer distinct cache lines per DMA instruction)
  r = r < nrows ? r : nrows - 1;
  return (uint32_t)(((r - row0) * ld + chunk * 8) * 2);
}
// k-strided operand X(row,k) = X[k*ld + row]: image [64 k][128 rows] (256-B k-rows), piece = 4 k-rows,
// 16-B chunk (8 rows) ^ ((k&3)<<2).
__device__ __forceinline__ uint32_t off_ks(int h, int idx, int lane, int64_t ld, int64_t row0, int64_t nrows) {
  const int krow = idx * 4 + (lane >> 4);
  const int chunk = (lane & 15) ^ ((krow & 3) << 2);
  int64_t r = row0 + h * 128 + chunk * 8;
  r = r + 8 <= nrows ? r : nrows - 8;
  return (uint32_t)((krow * ld + (r - row0)) * 2);
}

// ---- operand fragments.  The K loop reads 24 fragments per K-tile and wave; with the address arithmetic done per read
// (row * pitch + swizzle) it spent ~94 VALU instructions per K-tile on addresses - in the R segments, which must stay
// shorter than the partner group's 8 MFMAs.  Every fragment address is   lane base (VGPR, computed ONCE per kernel)
// + compile-time constant (ring buffer, half-tile unit, 32-row block, k chunk), i.e. `ds_read ... offset:imm`:
//   k-contiguous image [128 rows][128 B], 16-B chunk c stored at c ^ ((row>>1)&7): the chunk index c = 2 kc + (l>>5) enters
//     through the XOR, so there is one base per kc (4 VGPRs per operand and ring buffer; the 16-bit DS offset cannot
//     reach the second 64-KiB buffer from the first one's base);
//   k-strided image [64 k][256 B], chunk (row>>3) stored at ^ ((k&3)<<2): kc is a pure row offset (kc * 16 * 256), the
//     32-row block index flips a swizzled bit: one base per 32-row block.
typedef __attribute__((address_space(3))) const char lds_cchar;
typedef __attribute__((address_space(3))) const bf16x8_t lds_bf16x8;

// lane l -> row rbase + (l&31), k = kc*16 + 8*(l>>5) .. +7;  rbase must be a multiple of 16 (it does not enter the swizzle)
__device__ __forceinline__ uint32_t fragbase_direct(int rbase, int kc, int lane) {
  const int r = rbase + (lane & 31);
  const int c = kc * 2 + (lane >> 5);
  return (uint32_t)(r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
}
__device__ __forceinline__ uint32_t fragbase_ks(int rbase, int lane) {   // kc = 0; + kc * 4096 for the others
  const int g4 = lane >> 4, q = lane & 15;
  const int krow = 8 * (g4 >> 1) + (q >> 2);
  const int col = rbase + 16 * (g4 & 1) + 4 * (q & 3);
  return (uint32_t)(krow * 256 + ((((col >> 3) ^ ((krow & 3) << 2))) << 4) + ((col & 7) << 1));
}
template <int OFF> __device__ __forceinline__ bf16x8_t frag_direct_at(lds_cchar* base) {
  return *reinterpret_cast<lds_bf16x8*>(base + OFF);
}
template <int OFF> __device__ __forceinline__ bf16x8_t frag_ks_at(lds_cchar* base) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + OFF));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + OFF + 4 * 256));
  s16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, r);
}

template <int N> __device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else static_assert(N < 0, "unsupported vmcnt");
}

#define P8_BAR()                         \
  do {                                   \
    __builtin_amdgcn_sched_barrier(0);   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)

// ABL (experiments, SEGCLIP_P8_ABL, k-contiguous layout only): 1 no MFMAs, 2 no DMA, 3 no LDS reads, 4 no stagger
// between the wave groups - each leaves the rest of the schedule in place, results are garbage.
template <bool A_KS, bool B_KS, int ABL = 0>
__global__ __launch_bounds__(NWV * 64) void gemm_bf16_p8_kernel(Args g) {
  constexpr int LDS_BYTES = RING > NWV * EPI_WAVE_BYTES ? RING : NWV * EPI_WAVE_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES + 256];   // + junk slot of the side-tile touches
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // Workgroups go to the XCDs round-robin in dispatch order (x fastest, then y): the (tile, K-split) space is walked
  // split-major and cut into 8 contiguous chunks, one per XCD, so that the tiles that share an A row slab / B column
  // slab - and, with split-K, the SAME K range - sit behind one L2.
  const int nt = gridDim.x;
  const int nwg = nt * gridDim.y, bid = blockIdx.x + nt * blockIdx.y;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int unit_id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int wg = unit_id % nt, ksplit = unit_id / nt;
  // Tile order inside the sequence that is cut into the 8 XCD chunks.  Default: row-major over all nbx column tiles.
  // Experiment (g.colgroups = G > 1): the sequence walks column group 0 (all rows, its nbx/G columns, row-major), then
  // group 1, ...: an XCD's chunk lies in one group (or two), its weight working set is 1/G of W; every A slab is read by
  // G XCDs instead of one.  Hypothesis: row-major makes every XCD sweep the whole weight matrix once per round of its 32
  // workgroups while the A slabs stream through the same 4-MiB L2 (the weights cost more K-loop time than the
  // activations, tools/debug/gemm_latency_probe.py).  Measured: no gain (see the dispatcher).
  int tcol, trow;
  {
    const int G = g.colgroups;
    if (G <= 1) {
      tcol = wg % g.nbx; trow = wg / g.nbx;
    } else {
      const int cq = g.nbx / G, cr = g.nbx % G;      // the first cr groups have cq + 1 columns
      int rem = wg, c0 = 0, gi = 0, w = cq + (cr > 0 ? 1 : 0);
      while (gi + 1 < G && rem >= w * g.nby) { rem -= w * g.nby; c0 += w; ++gi; w = cq + (gi < cr ? 1 : 0); }
      tcol = c0 + rem % w; trow = rem / w;
    }
  }
  const int64_t n0 = (int64_t)tcol * BT, m0 = (int64_t)trow * BT;
  const int64_t z = blockIdx.z, z1 = z / g.nb2, z2 = z % g.nb2;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A) + z1 * g.bsA1 + z2 * g.bsA2;
  const bf16_t* B = g.B + z1 * g.bsB1 + z2 * g.bsB2;
  const int64_t coff = z1 * g.bsC1 + z2 * g.bsC2;
  const int64_t roff = z1 * g.bsR1 + z2 * g.bsR2;

  const int64_t kbeg = (int64_t)ksplit * g.kper;
  const int64_t kend = kbeg + g.kper < g.K ? kbeg + g.kper : g.K;
  const int nk = (int)((kend - kbeg) / BK);

  // uniform tile bases (bytes) + per-lane 32-bit offsets: the DMA address is SGPR base + VGPR offset
  // experiments (SEGCLIP_P8_EPI_ABL = 2..5, wrong results): operand rows taken from a small window that stays cache-resident
  const int64_t m0a = g.abl == 2 || g.abl == 4 ? (m0 & 511) : g.abl == 3 ? (m0 & 4095) : g.abl == 5 ? (m0 & 32767) : m0;
  const int64_t n0b = g.abl == 4 ? (n0 & 511) : n0;
  const char* baseA = reinterpret_cast<const char*>(A_KS ? A + kbeg * g.lda + m0a : A + m0a * g.lda + kbeg);
  const char* baseB = reinterpret_cast<const char*>(B_KS ? B + kbeg * g.ldb + n0b : B + n0b * g.ldb + kbeg);
  const int64_t stepA = A_KS ? (int64_t)BK * g.lda * 2 : BK * 2;   // bytes per K-tile
  const int64_t stepB = B_KS ? (int64_t)BK * g.ldb * 2 : BK * 2;
  // experiments 6..11: 4 / 2 / 1 distinct rows (cache lines) per DMA instruction, A and B (6-8) or B only (9-11)
  const int rm_ = g.abl >= 6 && g.abl <= 11 ? (2 << ((g.abl - 6) % 3)) - 1 : 0;
  const int rmaskA = g.abl >= 9 ? 0 : rm_, rmaskB = rm_;
  uint32_t offA[2][2], offB[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      offA[h][i] = A_KS ? off_ks(h, wave * 2 + i, lane, g.lda, m0, g.M) : off_direct(h, wave * 2 + i, lane, g.lda, m0, g.M, rmaskA);
      offB[h][i] = B_KS ? off_ks(h, wave * 2 + i, lane, g.ldb, n0, g.N) : off_direct(h, wave * 2 + i, lane, g.ldb, n0, g.N, rmaskB);
    }
  // half-tile `u` (0: A0, 1: A1, 2: B0, 3: B1) of K-tile t -> ring buffer t&1
  const uint32_t lds_ring = (uint32_t)(uintptr_t)((lds_void*)smem) + wave * 2048;
  auto stage = [&](int u, int t) {
    if constexpr (ABL == 2) return;
    const uint32_t dst = lds_ring + (t & 1) * BUF + u * UNIT;
    if (u < 2) dma16x2(baseA + (int64_t)t * stepA, offA[u][0], offA[u][1], dst);
    else dma16x2(baseB + (int64_t)t * stepB, offB[u - 2][0], offB[u - 2][1], dst);
  };

  // EXPERIMENT, off by default (SEGCLIP_P8_TOUCH=1): side tile of the epilogue (saved activation of the act' dgrad, or
  // the residual): every 64-byte sector of the tile's 256 rows is touched once now by a 4-byte LDS-DMA into a junk LDS
  // slot (no VGPR destination), so that the lines travel HBM -> MALL/L2 while the K loop runs.  The touches are the
  // oldest entries of the in-order VM queue: the first counted wait of the prologue covers them.  Measured on MI355X
  // (tools/bench_epi.py, M = 50176): SLOWER - out_proj +fp32 residual 89.8 -> 114.1 us, c_proj +residual 253 -> 268,
  // act' dgrad 326 -> 340: the epilogue is not waiting for the latency of these loads.
  if ((g.touch & 1) && g.splits == 1 && (g.mul_dact || g.residual) && n0 + BT <= g.N) {
    const int esz = g.mul_dact ? (g.c_dtype == SEGCLIP_BF16 ? 2 : 4) : (g.r_dtype == SEGCLIP_BF16 ? 2 : 4);
    const int64_t pitch = (g.mul_dact ? g.ldaux : g.ldr) * esz;
    const char* sp = g.mul_dact ? reinterpret_cast<const char*>(g.aux) + (coff + m0 * g.ldaux + n0) * esz
                                : reinterpret_cast<const char*>(g.residual) + (roff + m0 * g.ldr + n0) * esz;
    const int spr_log = esz == 2 ? 3 : 4;               // 64-byte sectors per tile row: 8 (bf16) / 16 (fp32)
    const int total = BT << spr_log;
    const int64_t rmax = g.M - 1 - m0;
    const uint32_t junk = (uint32_t)(uintptr_t)((lds_void*)smem) + LDS_BYTES;
    for (int sidx = tid; sidx < total; sidx += NWV * 64) {
      int64_t row = sidx >> spr_log;
      row = row < rmax ? row : rmax;
      const uint32_t off = (uint32_t)(row * pitch + ((sidx & ((1 << spr_log) - 1)) << 6));
      asm volatile(
          "s_nop 4\n\t"
          "s_mov_b32 m0, %2\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dword %0, %1"
          :
          : "v"(off), "s"(sp), "s"(junk)
          : "memory");
    }
  }

  f32x16 acc[2][2][2];  // [A half][32-row tile][B half]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][ri][j][r] = 0.f;

  // The first round of workgroups (one per CU) starts with a bounded, staggered delay so that the CUs do not all reach
  // their store phase at the same moment in every later round (kept from gemm_bf16_dma.hip, where it measured +5..10 %).
  if (!(g.touch & 2) && bid < 256 && gridDim.x * gridDim.y * gridDim.z > 256) {
    const long long t_tile = (long long)nk * 3000 + 20000;
    const long long cap = (long long)(g.touch >> 2);          // first-round stagger unit cap in cycles (default 2000)
    const long long unit = t_tile / 8 < cap ? t_tile / 8 : cap;
    // tiles of one block row share their A rows through the XCD's L2: they get the SAME delay and stay in lockstep
    const long long wait = (trow & 7) * unit;
    const long long t0 = clock64();
    while (clock64() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }

  // prologue: the issue order continues into the steady state (... B0 A0 B1 A1 ...): K-tile 0 and, of K-tile 1,
  // everything but A1 (issued in R1 of tile 0)
  stage(2, 0);
  stage(0, 0);
  stage(3, 0);
  stage(1, 0);
  if (nk > 1) {
    stage(2, 1);
    stage(0, 1);
    stage(3, 1);
    wait_vm<10>();   // B0(0), A0(0) have landed
  } else {
    wait_vm<4>();
  }
  P8_BAR();

  bf16x8_t fa[2][4], fbx[4], fby[4];
  if constexpr (ABL == 3) {
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      fa[0][kc] = fa[1][kc] = fbx[kc] = fby[kc] = __builtin_bit_cast(bf16x8_t, u32x4{(unsigned)lane, 1u, 2u, 3u});
    }
  }
  // lane bases of the fragment reads (see the fragment helpers): [ring buffer][kc] for a k-contiguous operand,
  // [ring buffer][32-row block] for a k-strided one; made opaque so that the compiler keeps them in registers instead
  // of re-deriving them in the K loop
  lds_cchar* const sm3 = (lds_cchar*)smem;
  lds_cchar* abase[2][4];
  lds_cchar* bbase[2][4];
#pragma unroll
  for (int bf = 0; bf < 2; ++bf)
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      uint32_t oa = A_KS ? fragbase_ks(wr * 64 + (x & 1) * 32, lane) : fragbase_direct(wr * 64, x, lane);
      uint32_t ob = B_KS ? fragbase_ks(wc * 32, lane) : fragbase_direct(wc * 32, x, lane);
      oa += bf * BUF; ob += bf * BUF;
      if ((!A_KS || x < 2)) asm volatile("" : "+v"(oa));
      if ((!B_KS || x < 1)) asm volatile("" : "+v"(ob));
      abase[bf][x] = sm3 + oa;
      bbase[bf][x] = sm3 + ob;
    }
  // U: byte offset of the half-tile unit inside its ring buffer
  auto read_a = [&](auto bfc, auto uc) {
    constexpr int BF_ = decltype(bfc)::value, U = decltype(uc)::value;
    if constexpr (ABL == 3) return;
#pragma unroll
    for (int ri = 0; ri < 2; ++ri) {
      if constexpr (A_KS) {
        fa[ri][0] = frag_ks_at<U + 0 * 4096>(abase[BF_][ri]);
        fa[ri][1] = frag_ks_at<U + 1 * 4096>(abase[BF_][ri]);
        fa[ri][2] = frag_ks_at<U + 2 * 4096>(abase[BF_][ri]);
        fa[ri][3] = frag_ks_at<U + 3 * 4096>(abase[BF_][ri]);
      } else {
        if (ri == 0) {
          fa[0][0] = frag_direct_at<U>(abase[BF_][0]); fa[0][1] = frag_direct_at<U>(abase[BF_][1]);
          fa[0][2] = frag_direct_at<U>(abase[BF_][2]); fa[0][3] = frag_direct_at<U>(abase[BF_][3]);
        } else {
          fa[1][0] = frag_direct_at<U + 4096>(abase[BF_][0]); fa[1][1] = frag_direct_at<U + 4096>(abase[BF_][1]);
          fa[1][2] = frag_direct_at<U + 4096>(abase[BF_][2]); fa[1][3] = frag_direct_at<U + 4096>(abase[BF_][3]);
        }
      }
    }
  };
  auto read_b = [&](auto bfc, auto uc, bf16x8_t (&fb)[4]) {
    constexpr int BF_ = decltype(bfc)::value, U = decltype(uc)::value;
    if constexpr (ABL == 3) return;
    if constexpr (B_KS) {
      fb[0] = frag_ks_at<U + 0 * 4096>(bbase[BF_][0]); fb[1] = frag_ks_at<U + 1 * 4096>(bbase[BF_][0]);
      fb[2] = frag_ks_at<U + 2 * 4096>(bbase[BF_][0]); fb[3] = frag_ks_at<U + 3 * 4096>(bbase[BF_][0]);
    } else {
      fb[0] = frag_direct_at<U>(bbase[BF_][0]); fb[1] = frag_direct_at<U>(bbase[BF_][1]);
      fb[2] = frag_direct_at<U>(bbase[BF_][2]); fb[3] = frag_direct_at<U>(bbase[BF_][3]);
    }
  };
  auto quadrant = [&](f32x16 (&c)[2][2], int j, const bf16x8_t (&fb)[4]) {
    if constexpr (ABL == 1) {  // keep the fragments live without issuing matrix instructions
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        asm volatile("" ::"v"(__builtin_bit_cast(u32x4, fa[0][kc])), "v"(__builtin_bit_cast(u32x4, fa[1][kc])),
                     "v"(__builtin_bit_cast(u32x4, fb[kc])));
      }
      return;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
      for (int ri = 0; ri < 2; ++ri) c[ri][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ri][kc], fb[kc], c[ri][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  using std::integral_constant;
  typedef integral_constant<int, 0> I0;
  typedef integral_constant<int, 1> I1;
  read_b(I0{}, integral_constant<int, 2 * UNIT>{}, fbx);   // B0 of K-tile 0 (in the loop this read sits in R4 of the previous tile)
  if (ABL != 4 && wr == 1) P8_BAR();     // group 1 runs one barrier interval behind group 0

  // one K-tile: fbp holds B0(t) on entry and fbq is free; on exit fbq holds B0(t+1)
  // (bc = t & 1 as a type: the ring-buffer offset of every fragment read is a compile-time constant)
  auto ktile = [&](auto bc, int t, bf16x8_t (&fbp)[4], bf16x8_t (&fbq)[4]) {
    typedef decltype(bc) CB;
    typedef integral_constant<int, 1 - CB::value> NB;
    const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
    // ---- phase 1: quadrant (A0, B0)
    read_a(CB{}, integral_constant<int, 0>{});
    if (n1) { stage(1, t + 1); wait_vm<10>(); } else { wait_vm<2>(); }          // retires B1(t), read in R2
    P8_BAR();
    quadrant(acc[0], 0, fbp);
    P8_BAR();
    // ---- phase 2: quadrant (A0, B1)
    read_b(CB{}, integral_constant<int, 3 * UNIT>{}, fbq);
    if (n2) { stage(2, t + 2); wait_vm<10>(); } else if (n1) { wait_vm<8>(); } else { wait_vm<0>(); }   // retires A1(t)
    P8_BAR();
    quadrant(acc[0], 1, fbq);
    P8_BAR();
    // ---- phase 3: quadrant (A1, B1)
    read_a(CB{}, integral_constant<int, UNIT>{});
    if (n2) { stage(0, t + 2); wait_vm<10>(); } else if (n1) { wait_vm<6>(); }   // retires B0(t+1), read in R4
    P8_BAR();
    quadrant(acc[1], 1, fbq);
    P8_BAR();
    // ---- phase 4: quadrant (A1, B0); B1's registers are free: B0(t+1) goes there
    if (n1) read_b(NB{}, integral_constant<int, 2 * UNIT>{}, fbq);
    if (n2) { stage(3, t + 2); wait_vm<10>(); } else if (n1) { wait_vm<4>(); }   // retires A0(t+1), read in R1
    P8_BAR();
    quadrant(acc[1], 0, fbp);
    P8_BAR();
  };
  int t = 0;
  for (; t + 1 < nk; t += 2) {
    ktile(I0{}, t, fbx, fby);
    ktile(I1{}, t + 1, fby, fbx);
  }
  if (t < nk) ktile(I0{}, t, fbx, fby);
  if (ABL != 4 && wr == 0) P8_BAR();  // group 0 catches up: both groups have executed the same number of barriers

  const int64_t nw = n0 + wc * 32;  // this wave's first column (second strip at +128)
  if (g.abl == 1) {   // experiment: no epilogue at all (one conditional store keeps the accumulators alive)
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc += acc[i][ri][j][r];
    if (sacc == 1.2345678f) reinterpret_cast<float*>(g.C)[lane] = sacc;
    return;
  }
  if (g.splits > 1) {
    const int li = lane & 31, lk = lane >> 5;
    float* slab = g.slab + ((int64_t)ksplit * gridDim.z + z) * g.M * g.N;
    // full tiles: the raw fp32 partial tile goes through the staged epilogue (park, 16-byte row stores: 32 store
    // instructions per wave instead of 128 one-dword stores); SEGCLIP_P8_SLAB_STAGED=0 keeps the per-element stores
    if (g.slab_staged && m0 + BT <= g.M && n0 + BT <= g.N && g.N % 4 == 0) {
      Args g2 = g;
      g2.C = slab; g2.ldc = g.N; g2.c_dtype = SEGCLIP_F32; g2.bias = nullptr; g2.residual = nullptr; g2.aux = nullptr;
      g2.act = SEGCLIP_ACT_NONE; g2.mul_dact = 0; g2.colsum_part = nullptr; g2.alpha = 1.0f;
      __syncthreads();  // every wave is done with the operand ring
      float* tps = reinterpret_cast<float*>(smem + wave * EPI_WAVE_BYTES);
      epilogue_lds2<float, EPI_PLAIN, 128>(g2, acc[0], acc[1], tps, m0 + wr * 64, m0 + 128 + wr * 64, nw, lane, 0, 0);
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int64_t n = nw + j * 128 + li;
          if (n >= g.N) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + i * 128 + wr * 64 + ri * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (m < g.M) __builtin_nontemporal_store(acc[i][ri][j][r], &slab[m * g.N + n]);
          }
        }
    return;
  }
  const bool full = m0 + BT <= g.M && n0 + BT <= g.N;  // uniform over the workgroup
  const bool vec = full && g.vec_epi;
  if (vec) __syncthreads();  // every wave is done with the operand ring: the LDS is reused as 8 private patches
  float* tp = reinterpret_cast<float*>(smem + wave * EPI_WAVE_BYTES);
  const int64_t mw0 = m0 + wr * 64, mw1 = m0 + 128 + wr * 64;
  if (vec) {
#ifdef P8_TEST_SINGLE
    if (g.c_dtype == SEGCLIP_BF16) { epilogue_lds_mode<bf16_t, 128>(g, acc[0], tp, mw0, nw, lane, coff, roff); __builtin_amdgcn_wave_barrier(); epilogue_lds_mode<bf16_t, 128>(g, acc[1], tp, mw1, nw, lane, coff, roff); }
    else { epilogue_lds_mode<float, 128>(g, acc[0], tp, mw0, nw, lane, coff, roff); __builtin_amdgcn_wave_barrier(); epilogue_lds_mode<float, 128>(g, acc[1], tp, mw1, nw, lane, coff, roff); }
#else
    if (g.c_dtype == SEGCLIP_BF16)
      epilogue_lds2_mode<bf16_t, 128>(g, acc[0], acc[1], tp, mw0, mw1, nw, lane, coff, roff,
                                      g.xw_epi ? reinterpret_cast<const float*>(smem) : nullptr, wr, wc, n0);
    else epilogue_lds2_mode<float, 128>(g, acc[0], acc[1], tp, mw0, mw1, nw, lane, coff, roff);
#endif
#ifndef P8_TEST_NORETURN
    return;
#endif
  }
#ifdef P8_TEST_NORETURN
  else {
#endif
  // two explicit copies (a loop over the A half would index the accumulators at run time -> scratch)
#define P8_EPI(I)                                                                                              \
  do {                                                                                                         \
    const int64_t mw = m0 + (I) * 128 + wr * 64;                                                               \
    if (g.c_dtype == SEGCLIP_BF16) epilogue_mode<bf16_t, false>(g, acc[I], mw, nw, 0, 0, lane, coff, roff, 128); \
    else epilogue_mode<float, false>(g, acc[I], mw, nw, 0, 0, lane, coff, roff, 128);                          \
  } while (0)
  P8_EPI(0);
  __builtin_amdgcn_wave_barrier();
  P8_EPI(1);
#undef P8_EPI
#ifdef P8_TEST_NORETURN
  }
#endif
}

}  // namespace

// The file is compiled once per P8_PART (segclip_amd/csrc/build.sh) so that the four operand-layout instances - each
// carries every epilogue variant and takes over a minute of hipcc time - build in parallel:
//   P8_PART 0..3 : kernel instance <A_KS = part>>1, B_KS = part&1> and its launcher
//   P8_PART 4    : the host-side dispatcher below (no device code)
//   P8_PART 5    : the main-loop ablation instances (only with -DSEGCLIP_P8_ABLATIONS; tools/bench_gemm_abl.py)
#ifndef P8_PART
#error "compile with -DP8_PART=0..5 (see build.sh)"
#endif
#define P8_LAUNCHER(NAME, ...)                                                      \
  void NAME(dim3 grid, hipStream_t stream, const void* args) {                      \
    const Args g = *reinterpret_cast<const Args*>(args);                            \
    hipLaunchKernelGGL((gemm_bf16_p8_kernel<__VA_ARGS__>), grid, dim3(512), 0, stream, g); \
  }
#if P8_PART == 0
P8_LAUNCHER(segclip_p8_launch_ff, false, false)
#elif P8_PART == 1
P8_LAUNCHER(segclip_p8_launch_fk, false, true)
#elif P8_PART == 2
P8_LAUNCHER(segclip_p8_launch_kf, true, false)
#elif P8_PART == 3
P8_LAUNCHER(segclip_p8_launch_kk, true, true)
#elif P8_PART == 5
P8_LAUNCHER(segclip_p8_launch_abl1, false, false, 1)
P8_LAUNCHER(segclip_p8_launch_abl2, false, false, 2)
P8_LAUNCHER(segclip_p8_launch_abl3, false, false, 3)
P8_LAUNCHER(segclip_p8_launch_abl4, false, false, 4)
#endif

#if P8_PART == 4
void segclip_p8_launch_ff(dim3, hipStream_t, const void*);
void segclip_p8_launch_fk(dim3, hipStream_t, const void*);
void segclip_p8_launch_kf(dim3, hipStream_t, const void*);
void segclip_p8_launch_kk(dim3, hipStream_t, const void*);
#ifdef SEGCLIP_P8_ABLATIONS
void segclip_p8_launch_abl1(dim3, hipStream_t, const void*);
void segclip_p8_launch_abl2(dim3, hipStream_t, const void*);
void segclip_p8_launch_abl3(dim3, hipStream_t, const void*);
void segclip_p8_launch_abl4(dim3, hipStream_t, const void*);
#endif

// Launch the phase-pipelined kernel for problems tiled 256x256.  `args_` is prepared by the caller (gemm_bf16.hip);
// returns false when the shape does not meet this kernel's preconditions.
bool segclip_gemm_bf16_p8_try(const segclip_gemm_desc* d, const void* args_, int splits, int64_t kper, int64_t nb,
                              hipStream_t stream) {
  static const int disabled = [] { const char* e = segclip_tuning_env("SEGCLIP_GEMM_P8"); return e ? atoi(e) == 0 : 0; }();
  if (disabled) return false;
  Args g = *reinterpret_cast<const Args*>(args_);
  const bool a_ks = d->sak != 1, b_ks = d->sbk != 1;
  if (d->a_dtype != SEGCLIP_BF16 || d->b_dtype != SEGCLIP_BF16) return false;
  if (d->K % BK != 0 || kper % BK != 0 || d->K < BK) return false;
  if (a_ks && (d->M % 8 != 0 || d->M < 8)) return false;
  if (b_ks && (d->N % 8 != 0 || d->N < 8)) return false;
  if (d->M < 64 || d->N <= 128) return false;
  // 32-bit DMA offsets: 256 rows (or 64 k-rows) of the leading dimension must stay below 4 GiB
  if ((a_ks ? 64 : 256) * (a_ks ? d->sak : d->sam) * 2 >= (int64_t)1 << 31) return false;
  if ((b_ks ? 64 : 256) * (b_ks ? d->sbk : d->sbn) * 2 >= (int64_t)1 << 31) return false;
  static const int touch = [] { const char* e = segclip_tuning_env("SEGCLIP_P8_TOUCH"); return e ? atoi(e) : 0; }();   // measured slower (see the kernel)
  static const int stagger = [] { const char* e = segclip_tuning_env("SEGCLIP_P8_STAGGER"); return e ? atoi(e) : 2000; }();   // unit cap in cycles; 0 = off.  In the step (two runs each): 5000: 43.74 ms, 2000: 43.50, 1000: 43.47, 0: 43.47, 12000: 43.98
  static const int epi_abl = segclip_ablation_env("SEGCLIP_P8_EPI_ABL");
  g.abl = epi_abl;
  g.touch = (touch ? 1 : 0) | (stagger > 0 ? 0 : 2) | (stagger << 2);   // bit 0: side-tile touch experiment, bit 1: no first-round stagger
  g.nbx = (int)cdiv(d->N, BT);
  g.nby = (int)cdiv(d->M, BT);
  // column groups of the tile order (experiment, SEGCLIP_P8_COLGROUPS = 2..4; default 1 = row-major): measured at
  // M = 50176 (tools/bench_epi.py): within +-3 % of row-major on every shape (N = 3072: -4 % with 3 groups; N = 768 with
  // 3 groups loses the A sharing: +18 %) - the weight refetch model in the kernel comment is not what bounds the K loop
  static const int cg_env = [] { const char* e = segclip_tuning_env("SEGCLIP_P8_COLGROUPS"); return e ? atoi(e) : 0; }();
  g.colgroups = splits > 1 ? 1 : (cg_env > 0 ? cg_env : 1);
  if (g.colgroups > g.nbx) g.colgroups = g.nbx;
  g.splits = splits;
  g.kper = kper;
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
#ifdef SEGCLIP_P8_ABLATIONS
  static const int abl = segclip_ablation_env("SEGCLIP_P8_ABL");
  if (!a_ks && !b_ks && abl >= 1 && abl <= 4) {
    (abl == 1 ? segclip_p8_launch_abl1 : abl == 2 ? segclip_p8_launch_abl2 : abl == 3 ? segclip_p8_launch_abl3
                                                                             : segclip_p8_launch_abl4)(grid, stream, &g);
    return true;
  }
#endif
  if (!a_ks && !b_ks) segclip_p8_launch_ff(grid, stream, &g);
  else if (!a_ks && b_ks) segclip_p8_launch_fk(grid, stream, &g);
  else if (a_ks && b_ks) segclip_p8_launch_kk(grid, stream, &g);
  else segclip_p8_launch_kf(grid, stream, &g);
  return true;
}
#endif  // P8_PART == 4
