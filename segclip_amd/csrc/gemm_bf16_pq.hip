// bf16 MFMA GEMM, 256x256x64 tiles, PERSISTENT workgroups with an overlapped output path (round 4).
//
// Same contract as gemm_bf16_p8.hip for the shapes it takes: C(m,n) = epi(sum_k A(m,k) B(n,k)), A bf16 k-contiguous,
// B bf16 k-contiguous (forward, weights (N,K)) or k-strided (data gradient, weights (K,N) read as B(n,k) = W[k][n]),
// C bf16, M and N multiples of 256, K a multiple of 64, no split-K, no batch.  The K loop is the 8-phase LDS-DMA ring of
// gemm_bf16_p8.hip (see there for the schedule); what is new is everything around it:
//
//  * The accumulators live in the ACCUMULATOR half of the unified register file (a[0:127]) and are only touched by inline
//    asm (v_mfma ... a[..], v_accvgpr_read): hipcc never sees 128 live accumulator values, so the tile loop that encloses
//    the K loop costs no spills (three attempts with compiler-managed accumulators ended with 30-70 spill reloads inside
//    the K loop).  With 2 waves per SIMD the budget is 128 architectural + 128 accumulator registers.
//  * The MFMA operands are SWAPPED (srcA = B fragment, srcB = A fragment): the 32x32 result block is then held transposed,
//    lane = output ROW, 16 registers = 4 groups of 4 CONSECUTIVE columns.  bf16 pairs are packed in-lane
//    (v_cvt_pk_bf16_f32 on adjacent registers, no lane exchange) and a group leaves as ONE ds_write_b64.
//  * Epilogue = four QUADRANTS (128 x 128) through one 32-KiB bf16 patch (the LDS the 128-KiB ring leaves free),
//    XOR-swizzled so that parking (ds_write_b64) and the row pass (ds_read_b128, then 16-byte stores of whole 256-byte
//    row segments) are both conflict-free: 32 + 16 LDS instructions per wave and tile instead of 128 + 32, half the bytes.
//  * A workgroup walks a static list of tiles (its XCD's contiguous chunk of the tile sequence, so the 32 workgroups
//    of an XCD always work on neighbouring tiles) and issues the NEXT tile's first 7 half-tile DMAs before it starts the
//    epilogue of the current one: the load round trip, the workgroup turnaround and the store drain overlap.
//    vmcnt is one in-order counter for loads and stores: the counted waits of the K loop stay correct (they are only
//    more conservative while output stores are still in flight).
//
// Epilogue modes: PQ_PLAIN (bias), PQ_RES (bias + bf16 residual), PQ_ACT8 (bias, QuickGELU, saved derivative as one byte),
// PQ_DACT8 (* saved derivative, + column sums of the output).  Side operands are read in the accumulator layout.
#include <stdlib.h>

#include <type_traits>

#include "gemm_bf16_common.h"

namespace {

constexpr int BK = 64, BT = 256, NWV = 8;
constexpr int UNIT = 128 * BK * 2;            // one half-tile: 16 KiB
constexpr int BUF = 4 * UNIT;                 // A0 A1 B0 B1
constexpr int RING = 2 * BUF;                 // 128 KiB
constexpr int PATCH = RING;                   // byte offset of the epilogue patch
constexpr int PATCH_BYTES = 32768;            // one quadrant, bf16
constexpr int LDS_BYTES = RING + PATCH_BYTES; // 160 KiB

enum { PQ_PLAIN = 0, PQ_RES = 1, PQ_ACT8 = 2, PQ_DACT8 = 3 };

struct PQArgs {
  const bf16_t* A; const bf16_t* B; bf16_t* C; const float* bias;
  const void* side;      // PQ_RES: bf16 residual (ld = lds); PQ_DACT8: uint8 saved derivative (ld = lds)
  uint8_t* aux;          // PQ_ACT8: uint8 saved derivative out (ld = ldaux)
  float* colsum_part;    // PQ_DACT8, optional: [M/64][N] partial column sums of the stored output
  int64_t lda, ldb, ldc, lds, ldaux;
  int N, K, nbx, ntiles;
  int abl;               // timing experiments (SEGCLIP_PQ_ABL, results garbage): 1 = no output stores, 2 = no epilogue at all
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) const char lds_cchar;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) const bf16x8_t lds_bf16x8;
typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
typedef __attribute__((address_space(3))) const u32x4 lds_cu32x4;

#define PQ_AGPRS                                                                                                          \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17",     \
  "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34",  \
  "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51",  \
  "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68",  \
  "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85",  \
  "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101",       \
  "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", \
  "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"

// accumulator block ACC (16 registers) += srcA x srcB; ZERO: = srcA x srcB (first K chunk of a tile: no zero-fill pass)
template <int ACC, bool ZERO>
__device__ __forceinline__ void mfma_acc(const bf16x8_t& a, const bf16x8_t& b) {
  if constexpr (ZERO)
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, 0" : : "v"(__builtin_bit_cast(u32x4, a)), "v"(__builtin_bit_cast(u32x4, b)), "i"(ACC * 16), "i"(ACC * 16 + 15) : PQ_AGPRS);
  else
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(__builtin_bit_cast(u32x4, a)), "v"(__builtin_bit_cast(u32x4, b)), "i"(ACC * 16), "i"(ACC * 16 + 15) : PQ_AGPRS);
}
// 8 accumulator registers a[B .. B+7] -> v[0..7]
template <int B> __device__ __forceinline__ void acc_read8(float* v) {
  asm volatile(
      "v_accvgpr_read_b32 %0, a%c8\n\tv_accvgpr_read_b32 %1, a%c9\n\tv_accvgpr_read_b32 %2, a%c10\n\t"
      "v_accvgpr_read_b32 %3, a%c11\n\tv_accvgpr_read_b32 %4, a%c12\n\tv_accvgpr_read_b32 %5, a%c13\n\t"
      "v_accvgpr_read_b32 %6, a%c14\n\tv_accvgpr_read_b32 %7, a%c15"
      : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7])
      : "i"(B), "i"(B + 1), "i"(B + 2), "i"(B + 3), "i"(B + 4), "i"(B + 5), "i"(B + 6), "i"(B + 7));
}

// Two LDS-DMA pieces (1 KiB each) of one half-tile; see gemm_bf16_p8.hip (inline asm: hidden from hipcc's waitcnt pass).
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

// per-lane DMA source offsets (bytes from the tile's first element), as in gemm_bf16_p8.hip; full tiles only
__device__ __forceinline__ uint32_t off_direct(int h, int idx, int lane, int64_t ld) {
  const int u = idx * 8 + (lane >> 3);
  const int chunk = (lane & 7) ^ ((u >> 1) & 7);
  return (uint32_t)(((int64_t)(h * 128 + u) * ld + chunk * 8) * 2);
}
__device__ __forceinline__ uint32_t off_ks(int h, int idx, int lane, int64_t ld) {
  const int krow = idx * 4 + (lane >> 4);
  const int chunk = (lane & 15) ^ ((krow & 3) << 2);
  return (uint32_t)((krow * ld + h * 128 + chunk * 8) * 2);
}
__device__ __forceinline__ uint32_t fragbase_direct(int rbase, int kc, int lane) {
  const int r = rbase + (lane & 31);
  const int c = kc * 2 + (lane >> 5);
  return (uint32_t)(r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
}
__device__ __forceinline__ uint32_t fragbase_ks(int rbase, int lane) {
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
  else if constexpr (N == 26) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
  else static_assert(N < 0, "unsupported vmcnt");
}

#define PQ_BAR()                         \
  do {                                   \
    __builtin_amdgcn_sched_barrier(0);   \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)
// barrier of the epilogue: this wave's LDS operations have completed before it arrives
#define PQ_BAR_LDS()                                         \
  do {                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
    PQ_BAR();                                                \
  } while (0)

// 16-byte store of whole row segments, issued from inline asm (not on hipcc's vmcnt scoreboard: the epilogue's barriers and
// the next K loop's counted waits must not turn into vmcnt(0)); s_nop 1: the data registers are read after issue
__device__ __forceinline__ void store16_nt(const void* base_uniform, uint32_t off, const u32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(off), "v"(v), "s"(base_uniform) : "memory");
}
__device__ __forceinline__ void store8_nt(const void* base_uniform, uint32_t off, const u32x2& v) {
  asm volatile("global_store_dwordx2 %0, %1, %2 nt\n\ts_nop 1" : : "v"(off), "v"(v), "s"(base_uniform) : "memory");
}

// ---- one quadrant (I = A half, J = B half) of the tile: accumulators -> [bias, side operand, activation] -> bf16 patch
// Accumulator layout (swapped MFMA): block (I, ri, J) = a[((I*2+ri)*2+J)*16 ..+15]; lane l holds row
// I*128 + wr*64 + ri*32 + (l&31), register r column J*128 + wc*32 + 8*(r>>2) + 4*(l>>5) + (r&3).
// Patch: [128 rows][256 B]; 16-byte chunk c of row R is stored at chunk c ^ (R&7), and its two 8-byte halves are swapped
// when (R>>3)&1: the 16 lanes a ds_write_b64 services together (16 consecutive rows, one column group) hit 16 different
// 8-byte slots of the 128-byte bank window.
template <int MODE, int I, int J, int RI>
__device__ __forceinline__ void pq_park_block(lds_char* sm, const f32x4 (&bias)[4], uint32_t base0, int x) {
  float v[16];
  acc_read8<((I * 2 + RI) * 2 + J) * 16>(v);
  acc_read8<((I * 2 + RI) * 2 + J) * 16 + 8>(v + 8);
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    float w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = v[gq * 4 + k] + bias[gq][k];
    u32x2 p;
    p[0] = pack2bf(w[0], w[1]);
    p[1] = pack2bf(w[2], w[3]);
    *reinterpret_cast<lds_u32x2*>(sm + base0 + RI * 8192 + ((gq ^ x) << 4)) = p;
  }
}
template <int MODE, int I, int J>
__device__ __forceinline__ void pq_park(const PQArgs& g, lds_char* sm, const f32x4 (&bias)[4], int lane, int wr, int wc,
                                        int64_t m0, int64_t n0) {
  const int li = lane & 31, lk = lane >> 5;
  const uint32_t base0 = PATCH + (uint32_t)(wr * 64 + li) * 256 + ((((uint32_t)wc * 4) ^ (li & 4)) << 4) +
                         ((lk ^ ((li >> 3) & 1)) << 3);
  pq_park_block<MODE, I, J, 0>(sm, bias, base0, li & 3);
  pq_park_block<MODE, I, J, 1>(sm, bias, base0, li & 3);
}

// row pass of one quadrant: wave w moves rows w*16 .. w*16+15 of the patch (4 rows = 4 x 256 B per instruction)
template <int I, int J>
__device__ __forceinline__ void pq_rows(const PQArgs& g, lds_cchar* sm, int lane, int wave, const bf16_t* cq /* uniform: C + (m0 + I*128)*ldc + n0 + J*128 */) {
  const int p = lane & 15, rs = lane >> 4;
  const int c0 = p ^ rs;
  const uint32_t rd = PATCH + (uint32_t)wave * 4096 + (uint32_t)lane * 16;
  const int64_t ldcb = g.ldc * 2;
  const uint32_t o_even = (uint32_t)((wave * 16 + rs) * ldcb + c0 * 16);
  const uint32_t o_odd = (uint32_t)((wave * 16 + rs) * ldcb + (c0 ^ 4) * 16);
  u32x4 d[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) d[it] = *reinterpret_cast<lds_cu32x4*>(sm + rd + it * 1024);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    if (it & 2) d[it] = u32x4{d[it][2], d[it][3], d[it][0], d[it][1]};   // rows with bit 3 set keep their 8-byte halves swapped
    const char* bq = reinterpret_cast<const char*>(cq) + (int64_t)it * 4 * ldcb;
    if (!(g.abl & 1)) store16_nt(bq, (it & 1) ? o_odd : o_even, d[it]);
  }
}

template <bool B_KS, int MODE>
__global__ __launch_bounds__(NWV * 64) void gemm_bf16_pq_kernel(PQArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  int lane = tid & 63;

  // ---- static unit walk: the tile sequence (row-major, columns fastest) is cut into 8 contiguous chunks, one per XCD
  // (workgroup b runs on XCD b % 8); the workgroups of an XCD take the tiles of its chunk round-robin, so at any moment
  // they work on up to 32 neighbouring tiles that share A row slabs / B column slabs through the XCD's L2.
  const int G = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int nslot = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
  const int q8 = g.ntiles >> 3, r8 = g.ntiles & 7;
  const int cbeg = xcd * q8 + (xcd < r8 ? xcd : r8), clen = q8 + (xcd < r8 ? 1 : 0);
  if (slot >= clen) return;
  const int nk = g.K / BK;
  const bool slowwait = (g.abl & 4) != 0;   // experiment: keep the conservative vmcnt(10) waits behind an epilogue
  const int64_t stepA = BK * 2;
  const int64_t stepB = B_KS ? (int64_t)BK * g.ldb * 2 : BK * 2;

  // per-lane DMA offsets and fragment bases: functions of the lane and the leading dimensions only (tile-invariant)
  uint32_t offA[2][2], offB[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      offA[h][i] = off_direct(h, wave * 2 + i, lane, g.lda);
      offB[h][i] = B_KS ? off_ks(h, wave * 2 + i, lane, g.ldb) : off_direct(h, wave * 2 + i, lane, g.ldb);
    }
  const uint32_t lds_ring = (uint32_t)(uintptr_t)((lds_void*)smem) + wave * 2048;
  lds_cchar* const sm3 = (lds_cchar*)smem;
  lds_cchar* abase[2][4];
  lds_cchar* bbase[2][4];
#pragma unroll
  for (int bf = 0; bf < 2; ++bf)
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      uint32_t oa = fragbase_direct(wr * 64, x, lane);
      uint32_t ob = B_KS ? fragbase_ks(wc * 32, lane) : fragbase_direct(wc * 32, x, lane);
      oa += bf * BUF; ob += bf * BUF;
      asm volatile("" : "+v"(oa));
      if ((!B_KS || x < 1)) asm volatile("" : "+v"(ob));
      abase[bf][x] = sm3 + oa;
      bbase[bf][x] = sm3 + ob;
    }

  auto stage = [&](const char* baseA, const char* baseB, int u, int t) {
    const uint32_t dst = lds_ring + (t & 1) * BUF + u * UNIT;
    if (u < 2) dma16x2(baseA + (int64_t)t * stepA, offA[u][0], offA[u][1], dst);
    else dma16x2(baseB + (int64_t)t * stepB, offB[u - 2][0], offB[u - 2][1], dst);
  };
  // first 7 half-tiles of a tile (K-tile 0 and, of K-tile 1, everything but A1, which R1 of K-tile 0 issues)
  auto prologue = [&](const char* baseA, const char* baseB) {
    stage(baseA, baseB, 2, 0);
    stage(baseA, baseB, 0, 0);
    stage(baseA, baseB, 3, 0);
    stage(baseA, baseB, 1, 0);
    if (nk > 1) {
      stage(baseA, baseB, 2, 1);
      stage(baseA, baseB, 0, 1);
      stage(baseA, baseB, 3, 1);
    }
  };
  auto tile_bases = [&](int unit, int64_t& m0, int64_t& n0, const char*& baseA, const char*& baseB) {
    const int tcol = unit % g.nbx, trow = unit / g.nbx;
    m0 = (int64_t)trow * BT; n0 = (int64_t)tcol * BT;
    baseA = reinterpret_cast<const char*>(g.A + m0 * g.lda);
    baseB = reinterpret_cast<const char*>(B_KS ? g.B + n0 : g.B + n0 * g.ldb);
  };

  bf16x8_t fa[2][4], fbx[4], fby[4];
  using std::integral_constant;
  typedef integral_constant<int, 0> I0;
  typedef integral_constant<int, 1> I1;
  auto read_a = [&](auto bfc, auto uc) {
    constexpr int BF_ = decltype(bfc)::value, U = decltype(uc)::value;
    fa[0][0] = frag_direct_at<U>(abase[BF_][0]); fa[0][1] = frag_direct_at<U>(abase[BF_][1]);
    fa[0][2] = frag_direct_at<U>(abase[BF_][2]); fa[0][3] = frag_direct_at<U>(abase[BF_][3]);
    fa[1][0] = frag_direct_at<U + 4096>(abase[BF_][0]); fa[1][1] = frag_direct_at<U + 4096>(abase[BF_][1]);
    fa[1][2] = frag_direct_at<U + 4096>(abase[BF_][2]); fa[1][3] = frag_direct_at<U + 4096>(abase[BF_][3]);
  };
  auto read_b = [&](auto bfc, auto uc, bf16x8_t (&fb)[4]) {
    constexpr int BF_ = decltype(bfc)::value, U = decltype(uc)::value;
    if constexpr (B_KS) {
      fb[0] = frag_ks_at<U + 0 * 4096>(bbase[BF_][0]); fb[1] = frag_ks_at<U + 1 * 4096>(bbase[BF_][0]);
      fb[2] = frag_ks_at<U + 2 * 4096>(bbase[BF_][0]); fb[3] = frag_ks_at<U + 3 * 4096>(bbase[BF_][0]);
    } else {
      fb[0] = frag_direct_at<U>(bbase[BF_][0]); fb[1] = frag_direct_at<U>(bbase[BF_][1]);
      fb[2] = frag_direct_at<U>(bbase[BF_][2]); fb[3] = frag_direct_at<U>(bbase[BF_][3]);
    }
  };
  // 8 MFMAs of one C quadrant (A half I, B half J) and one K-tile; srcA = B fragment: the result block is transposed
  auto quadrant = [&](auto ic, auto jc, auto zc, const bf16x8_t (&fb)[4]) {
    constexpr int I = decltype(ic)::value, J = decltype(jc)::value;
    constexpr bool Z = decltype(zc)::value != 0;
    __builtin_amdgcn_s_setprio(1);
    mfma_acc<(I * 2 + 0) * 2 + J, Z>(fb[0], fa[0][0]);
    mfma_acc<(I * 2 + 1) * 2 + J, Z>(fb[0], fa[1][0]);
#pragma unroll
    for (int kc = 1; kc < 4; ++kc) {
      mfma_acc<(I * 2 + 0) * 2 + J, false>(fb[kc], fa[0][kc]);
      mfma_acc<(I * 2 + 1) * 2 + J, false>(fb[kc], fa[1][kc]);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  int64_t m0, n0;
  const char *baseA, *baseB;
  int iu = slot;
  bool post_store = false;   // see ktile(): the previous epilogue's stores are still in the VM queue
  tile_bases(cbeg + iu, m0, n0, baseA, baseB);
  prologue(baseA, baseB);

  while (true) {
    // one K-tile: fbp holds B0(t) on entry and fbq is free; on exit fbq holds B0(t+1)
    // ps: this tile follows an epilogue of this workgroup, whose PQ_NSTORE output stores sit in the VM queue BEHIND the 14
    // prologue pieces issued before them.  vmcnt retires loads and stores in issue order, so the first six waits of the
    // tile (K-tile 0 and R1 of K-tile 1, which retire exactly those 14 pieces) may leave the stores in flight as well:
    // vmcnt(10 + 16).  From R2 of K-tile 1 on the retired piece is younger than the stores and the count is 10 again.
    auto ktile = [&](auto bc, auto zc, int t, bool ps, bf16x8_t (&fbp)[4], bf16x8_t (&fbq)[4]) {
      typedef decltype(bc) CB;
      typedef integral_constant<int, 1 - CB::value> NB;
      typedef decltype(zc) Z;
      const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
      const bool ps0 = ps && t == 0, ps1 = ps && t <= 1;
      auto wait_steady = [&](bool p) { if (p) wait_vm<26>(); else wait_vm<10>(); };
      // ---- phase 1: quadrant (A0, B0)
      read_a(CB{}, integral_constant<int, 0>{});
      if (n1) { stage(baseA, baseB, 1, t + 1); wait_steady(ps1); } else { wait_vm<2>(); }          // retires B1(t), read in R2
      PQ_BAR();
      quadrant(I0{}, I0{}, Z{}, fbp);
      PQ_BAR();
      // ---- phase 2: quadrant (A0, B1)
      read_b(CB{}, integral_constant<int, 3 * UNIT>{}, fbq);
      if (n2) { stage(baseA, baseB, 2, t + 2); wait_steady(ps0); } else if (n1) { wait_vm<8>(); } else { wait_vm<0>(); }   // retires A1(t)
      PQ_BAR();
      quadrant(I0{}, I1{}, Z{}, fbq);
      PQ_BAR();
      // ---- phase 3: quadrant (A1, B1)
      read_a(CB{}, integral_constant<int, UNIT>{});
      if (n2) { stage(baseA, baseB, 0, t + 2); wait_steady(ps0); } else if (n1) { wait_vm<6>(); }   // retires B0(t+1), read in R4
      PQ_BAR();
      quadrant(I1{}, I1{}, Z{}, fbq);
      PQ_BAR();
      // ---- phase 4: quadrant (A1, B0); B1's registers are free: B0(t+1) goes there
      if (n1) read_b(NB{}, integral_constant<int, 2 * UNIT>{}, fbq);
      if (n2) { stage(baseA, baseB, 3, t + 2); wait_steady(ps0); } else if (n1) { wait_vm<4>(); }   // retires A0(t+1), read in R1
      PQ_BAR();
      quadrant(I1{}, I0{}, Z{}, fbp);
      PQ_BAR();
    };

    if (post_store) wait_vm<26>(); else if (nk > 1) wait_vm<10>(); else wait_vm<4>();   // B0(0), A0(0) have landed
    PQ_BAR();
    read_b(I0{}, integral_constant<int, 2 * UNIT>{}, fbx);
    if (wr == 1) PQ_BAR();     // group 1 runs one barrier interval behind group 0
    ktile(I0{}, I1{}, 0, post_store, fbx, fby);     // first K-tile: the MFMAs overwrite the accumulators (C = 0)
    int t = 1;
    for (; t + 1 < nk; t += 2) {
      ktile(I1{}, I0{}, t, post_store, fby, fbx);
      ktile(I0{}, I0{}, t + 1, false, fbx, fby);
    }
    if (t < nk) ktile(I1{}, I0{}, t, false, fby, fbx);
    if (wr == 0) PQ_BAR();     // group 0 catches up: every wave is done with the operand ring

    // ---- the next tile's first DMA rounds go out before this tile's output
    const int inext = iu + nslot;
    const bool have_next = inext < clen;
    const int64_t cm0 = m0, cn0 = n0;
    asm volatile("" : "+v"(lane));   // lane-dependent epilogue addresses are re-derived per tile, not kept live across the K loop
    const int li = lane & 31, lk = lane >> 5;
    (void)li;
    // bias: this wave's 2 x 32 columns travel by ONE 4-byte LDS-DMA instruction (no VGPR destination: a VGPR-destination asm
    // load let hipcc copy the registers before the data had landed) into a wave-private 256-byte slot of the ring's A1 unit
    // of buffer 1 - the one half-tile slot the next tile's prologue leaves alone - and is the oldest entry of the VM queue
    // when the prologue pieces follow it; the L2 round trip passes under their issue time.
    const bool has_bias = g.bias != nullptr && MODE != PQ_DACT8;
    constexpr uint32_t BIAS_SLOT = BUF + UNIT;   // byte offset of A1(1)
    if (has_bias) {
      const float* bsrc = g.bias + cn0 + wc * 32;                  // uniform
      const uint32_t boff = (uint32_t)(((lane >> 5) * 128 + (lane & 31)) * 4);
      const uint32_t bdst = (uint32_t)(uintptr_t)((lds_void*)smem) + BIAS_SLOT + wave * 256;
      asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(boff), "s"(bsrc), "s"(bdst) : "memory");
    }
    if (have_next) {
      iu = inext;
      tile_bases(cbeg + iu, m0, n0, baseA, baseB);
      prologue(baseA, baseB);
    }
    // the bias piece is older than every DMA piece just issued: 14 (nk > 1) or 8 of those may stay in flight
    if (!have_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    f32x4 bias[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        bias[j][q] = has_bias ? *reinterpret_cast<__attribute__((address_space(3))) const f32x4*>(
                                    sm3 + BIAS_SLOT + wave * 256 + (j * 32 + q * 8 + lk * 4) * 4)
                              : f32x4{0.f, 0.f, 0.f, 0.f};
    // MFMA results -> v_accvgpr_read: the last MFMA was issued a barrier ago; 16-pass XDL needs 18 wait states
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");

    lds_char* const smw = (lds_char*)smem;
    const bf16_t* ct = g.C + cm0 * g.ldc + cn0;
#define PQ_QUADRANT(I, J)                                                                      \
  do {                                                                                         \
    pq_park<MODE, I, J>(g, smw, bias[J], lane, wr, wc, cm0, cn0);                              \
    PQ_BAR_LDS();                                                                              \
    pq_rows<I, J>(g, sm3, lane, wave, ct + (int64_t)(I) * 128 * g.ldc + (J) * 128);            \
    PQ_BAR_LDS();                                                                              \
  } while (0)
    if (!(g.abl & 2)) {
      PQ_QUADRANT(0, 0);
      PQ_QUADRANT(0, 1);
      PQ_QUADRANT(1, 0);
      PQ_QUADRANT(1, 1);
    }
#undef PQ_QUADRANT
    if (!have_next) break;
    // K-tiles 0, 1 must be steady-state K-tiles (n2) for the counted form; the ablation without stores counts none
    post_store = nk >= 4 && !(g.abl & 3) && !(slowwait);
  }
}

}  // namespace

// The file is compiled once per PQ_PART (build.sh): 0 = forward layout, 1 = data-gradient layout, 2 = host dispatcher
#ifndef PQ_PART
#error "compile with -DPQ_PART=0..2 (see build.sh)"
#endif
#define PQ_LAUNCHER(NAME, BKS)                                                                        \
  void NAME(int mode, dim3 grid, hipStream_t stream, const void* args) {                              \
    const PQArgs g = *reinterpret_cast<const PQArgs*>(args);                                          \
    hipLaunchKernelGGL((gemm_bf16_pq_kernel<BKS, PQ_PLAIN>), grid, dim3(NWV * 64), 0, stream, g);     \
  }
#if PQ_PART == 0
PQ_LAUNCHER(segclip_pq_launch_f, false)
#elif PQ_PART == 1
PQ_LAUNCHER(segclip_pq_launch_k, true)
#endif

#if PQ_PART == 2
void segclip_pq_launch_f(int, dim3, hipStream_t, const void*);
void segclip_pq_launch_k(int, dim3, hipStream_t, const void*);

// Launch the persistent kernel when the problem meets its preconditions (see the top of the file); false = not taken.
bool segclip_gemm_bf16_pq_try(const segclip_gemm_desc* d, const void* args_, int splits, int64_t nb, hipStream_t stream) {
  // SEGCLIP_GEMM_PQ: 1 (default) on, 0 off, 2 = consult SEGCLIP_GEMM_PQ_NOW (0/1) at every call (A/B tests in one process)
  static const int mode = [] { const char* e = getenv("SEGCLIP_GEMM_PQ"); return e ? atoi(e) : 1; }();
  if (mode == 0) return false;
  if (mode == 2) { const char* e = getenv("SEGCLIP_GEMM_PQ_NOW"); if (e && atoi(e) == 0) return false; }
  const Args& a = *reinterpret_cast<const Args*>(args_);
  const bool a_ks = d->sak != 1, b_ks = d->sbk != 1;
  if (a_ks || splits != 1 || nb != 1) return false;
  if (d->a_dtype != SEGCLIP_BF16 || d->b_dtype != SEGCLIP_BF16 || d->c_dtype != SEGCLIP_BF16) return false;
  if (d->M % BT != 0 || d->N % BT != 0 || d->K % BK != 0 || d->K < BK) return false;
  if (d->alpha != 1.0f || a.colsum_part != nullptr) return false;
  if (d->residual != nullptr || d->aux != nullptr || d->act != SEGCLIP_ACT_NONE || d->mul_dact) return false;   // PQ_PLAIN only so far
  auto al = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!al(d->A) || !al(d->B) || !al(d->C) || !al(d->bias) || d->ldc % 8 != 0) return false;
  const int64_t lda = d->sam, ldb = b_ks ? d->sbk : d->sbn;
  if (lda % 8 != 0 || ldb % 8 != 0) return false;
  // 32-bit per-lane offsets: 256 rows (64 k-rows) of an operand and 256 rows of the output stay below 2 GiB
  if (256 * lda * 2 >= (int64_t)1 << 31 || (b_ks ? 64 : 256) * ldb * 2 >= (int64_t)1 << 31) return false;
  if (256 * d->ldc * 2 >= (int64_t)1 << 31) return false;
  PQArgs g;
  g.A = reinterpret_cast<const bf16_t*>(d->A); g.B = reinterpret_cast<const bf16_t*>(d->B);
  g.C = reinterpret_cast<bf16_t*>(d->C); g.bias = d->bias;
  g.side = nullptr; g.aux = nullptr; g.colsum_part = nullptr;
  g.lda = lda; g.ldb = ldb; g.ldc = d->ldc; g.lds = 0; g.ldaux = 0;
  g.N = (int)d->N; g.K = (int)d->K; g.nbx = (int)(d->N / BT); g.ntiles = (int)((d->M / BT) * (d->N / BT));
  // one workgroup per CU (160 KiB of LDS); SEGCLIP_PQ_PERSIST=0: one tile per workgroup (the same kernel, no tile loop taken)
  static const int persist0 = [] { const char* e = getenv("SEGCLIP_PQ_PERSIST"); return e ? atoi(e) : 1; }();
  int persist = persist0;
  if (mode == 2) { const char* e = getenv("SEGCLIP_PQ_PERSIST_NOW"); if (e) persist = atoi(e); }
  static const int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  { const char* e = mode == 2 ? getenv("SEGCLIP_PQ_ABL") : nullptr; g.abl = e ? atoi(e) : 0; }
  const int grid = persist && g.ntiles > ncu ? ncu : g.ntiles;
  (b_ks ? segclip_pq_launch_k : segclip_pq_launch_f)(PQ_PLAIN, dim3((unsigned)grid), stream, &g);
  return true;
}
#endif  // PQ_PART == 2
