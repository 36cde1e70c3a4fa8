// bf16 MFMA GEMM, 256x256x64 tiles, accumulators in the accumulator register file, transposed result blocks and a
// packed-bf16 output path (round 4).  Takes over from gemm_bf16_p8.hip the shapes that carry the training step:
// C(m,n) = epi(sum_k A(m,k) B(n,k)), A bf16 k-contiguous, B bf16 k-contiguous (forward, weights (N,K)) or k-strided (data
// gradient, weights (K,N) read as B(n,k) = W[k][n]), C bf16, M and N multiples of 256, K a multiple of 64, no split-K, no
// batch.  The K loop is the 8-phase LDS-DMA ring of gemm_bf16_p8.hip (see there for the schedule); new is the rest:
//
//  * The 128 accumulators live in a[0:127] and are only touched by inline asm (v_mfma ... a[..], v_accvgpr_read): hipcc
//    sees a kernel with ~100 live registers, compiles it in seconds instead of minutes and spills nothing (the compiler-
//    managed form of gemm_bf16_p8.hip sits at 256 VGPRs with 31 spilled).  Budget with 2 waves per SIMD: 128 + 128.
//  * The MFMA operands are SWAPPED (srcA = B fragment, srcB = A fragment): the 32x32 result block is held transposed,
//    lane = output ROW, 16 registers = 4 groups of 4 CONSECUTIVE columns.  Everything elementwise (bias, QuickGELU and
//    its saved derivative, act' multiplier) happens in that layout in fp32; bf16 pairs are packed in-lane
//    (v_cvt_pk_bf16_f32 on adjacent registers, no lane exchange) and a group leaves as ONE ds_write_b64.
//  * Output = two HALVES (128 rows) of two QUADRANTS (128 x 128) each, parked as bf16 in XOR-swizzled 32-KiB patches
//    (conflict-free ds_write_b64 on the way in, ds_read_b128 on the way out) and stored as whole 128-byte lines: 32 + 16
//    LDS instructions per wave and tile instead of 128 + 32 on fp32 data, 3 workgroup barriers instead of 12.
//  * Side operands never occupy registers during the K loop: the bias (256 B per wave) and the saved-derivative bytes of the
//    first half travel by LDS-DMA into the 32 KiB the ring leaves free, the second half's follow into ring space once the
//    K loop is over; they are re-read in the accumulator layout with ds_read.
//
// Measured (MI355X, M = 50176, tools/bench_pq.py, bit-identical outputs): forward N = 2304, K = 768 195 -> 158 us, data
// gradients 4-9 % faster than gemm_bf16_p8.hip; per tile (tools/debug/pq_ksweep.py): K loop 1.43 us per K-tile, fixed cost
// 1.5 us + output VALU/LDS 1.4 us + output stores 2.0-2.3 us (8-phase kernel: 5.9 us of output path).
// A PERSISTENT form of this kernel (static tile walk per XCD, the next tile's first 7 half-tile DMAs issued before the
// epilogue, the K loop's counted waits extended past the 16 output stores: vmcnt(26)) was built and measured first
// (git history: dc78ad3): bit-exact, and NOT faster than one tile per workgroup (N = 2304, K = 768: 153.8 vs 155.8 us;
// in the training step 43.23 vs 43.33 ms): with the epilogue removed both forms take the same 130 us, i.e. the
// workgroup turnaround was never exposed, and the output stores cost their 2 us per tile as memory-system throughput
// wherever they are issued (the same time with and without counted waits past them).  The one-tile form keeps the
// hardware's dynamic workgroup scheduling, which the two concurrent towers rely on, and frees the whole ring for the output.
#include <stdlib.h>

#include <type_traits>

#include "gemm_bf16_common.h"

namespace {

constexpr int BK = 64, BT = 256, NWV = 8;
constexpr int UNIT = 128 * BK * 2;            // one half-tile: 16 KiB
constexpr int BUF = 4 * UNIT;                 // A0 A1 B0 B1
constexpr int RING = 2 * BUF;                 // 128 KiB
constexpr int EXTRA = RING;                   // 32 KiB beyond the ring: free during the K loop
constexpr int LDS_BYTES = RING + 32768;       // 160 KiB
// epilogue map (the ring is free by then)
constexpr int H_OFF = 0;                      // bf16 output patches of the current half: J * 32 KiB
constexpr int AO_OFF = 65536;                 // uint8 saved-derivative patches (PQ_ACT8): J * 16 KiB
constexpr int SI1_OFF = 98304;                // side-in of the second half (PQ_DACT8: uint8, J * 16 KiB)
constexpr int SI0_OFF = EXTRA;                // side-in of the first half, fetched during the K loop
constexpr int BIAS_OFF = EXTRA;               // bias: wave * 256 B (modes with a bias have no side-in)
// half-tile tail (128 x 256 workgroups, see pq_main): a 3-slot ring of {A0, B0, B1} = 3 x 48 KiB, the bias behind it
constexpr int HSLOT = 3 * UNIT;
constexpr int BIAS_OFF_H = 3 * HSLOT;         // 144 KiB

enum { PQ_PLAIN = 0, PQ_RES = 1, PQ_ACT8 = 2, PQ_DACT8 = 3, PQ_SLAB = 4, PQ_RES32 = 5, PQ_ACT8E = 6 };
// PQ_ACT8E: PQ_ACT8 with the erf-GELU of the MAE decoders (modules/module_mae.py:110-134: timm Block, nn.GELU) instead of QuickGELU
template <int MODE> constexpr bool pq_is_act8() { return MODE == PQ_ACT8 || MODE == PQ_ACT8E; }

struct PQArgs {
  const bf16_t* A; const bf16_t* B; bf16_t* C; const float* bias;
  const void* side;      // PQ_RES: bf16 / PQ_RES32: fp32 residual (ld = lds); PQ_DACT8: uint8 saved derivative (ld = lds)
  uint8_t* aux;          // PQ_ACT8: uint8 saved derivative out (ld = ldaux)
  float* colsum_part;    // PQ_DACT8, optional: [M/64][N] partial column sums of the stored output
  int64_t lda, ldb, ldc, lds, ldaux;
  int N, K, nbx, ntiles;
  float* Cf;             // PQ_SLAB / PQ_RES32: fp32 output (split-K: slab of K range s at Cf + s * slab_stride), row pitch ldc
  int64_t kper, slab_stride;   // PQ_SLAB: K elements per split (a multiple of 64)
  // tail split (every mode but PQ_SLAB): the tail_r tiles of the last, partial round of 256 workgroups are cut into tail_S K
  // ranges; partial accumulators meet in tail_ws [tail_r][tail_S][128 registers][512 lanes] fp32 and the LAST workgroup of a
  // tile to arrive (tail_cnt[tile], zeroed by the host) sums them in K-range order and runs the epilogue
  int tail_S, tail_r, nfull;
  float* tail_ws; int* tail_cnt;
  int rmod;              // PQ_RES32: > 0 -> residual row = output row % rmod (a table broadcast over the samples)
  // half-tile tail (every mode but PQ_SLAB): the half_r tiles of the last, partial round of 256 workgroups run as
  // 2 * half_r workgroups of 128 x 256 (workgroups nfull ..: the upper / lower 128 rows of tile nfull + j / 2)
  int half_r;
  // remainder row: M = 256 q + 128 -> half_x = nbx more half-tile workgroups (the upper halves of tile row q), behind the tail's
  int half_x;
  int abl;               // timing experiments (SEGCLIP_PQ_ABL, results garbage): 1 = no output stores, 2 = no epilogue at all
};

// Grouped weight gradients (PQ_SLAB): several C_p = A_p^T B_p problems with the same K (token rows) and the same number of K
// ranges run as ONE launch - the (K range, tile) units of problem p are [unit_end[p-1], unit_end[p]).  A residual block has four
// weight gradients of 9-36 tiles each: launched one by one every one of them needs 7-28 K ranges to fill the 256 CUs (64 MB
// of fp32 partial tiles written and read back per gradient); the gradients of several blocks together fill the chip with 1-4.
struct PQProb {
  const bf16_t* A; const bf16_t* B; float* Cf;
  int64_t slab_stride;
  int lda, ldb, ldc, nbx, ntiles, unit_end;
};
constexpr int PQ_GROUP_MAX = 48;
struct PQGroup { int nprob; int pad; PQProb prob[PQ_GROUP_MAX]; };

typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) const char lds_cchar;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) const bf16x8_t lds_bf16x8;
typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
typedef __attribute__((address_space(3))) const u32x4 lds_cu32x4;
typedef __attribute__((address_space(3))) const f32x4 lds_cf32x4;

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

template <int B> __device__ __forceinline__ void acc_write8(const float* v) {
  asm volatile(
      "v_accvgpr_write_b32 a%c8, %0\n\tv_accvgpr_write_b32 a%c9, %1\n\tv_accvgpr_write_b32 a%c10, %2\n\t"
      "v_accvgpr_write_b32 a%c11, %3\n\tv_accvgpr_write_b32 a%c12, %4\n\tv_accvgpr_write_b32 a%c13, %5\n\t"
      "v_accvgpr_write_b32 a%c14, %6\n\tv_accvgpr_write_b32 a%c15, %7"
      :
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "i"(B), "i"(B + 1), "i"(B + 2),
        "i"(B + 3), "i"(B + 4), "i"(B + 5), "i"(B + 6), "i"(B + 7)
      : PQ_AGPRS);
}

// ---- tail split: exchange of partial accumulators.  Element (register r, thread t) of a partial tile lives at
// [(r >> 2) * 512 + t] * 4 + (r & 3): a 16-byte store / load per lane and register quad, 1 KiB contiguous per wave.
template <int C> __device__ __forceinline__ void tail_store_chunk(float* wp, int tid) {
  float v[8];
  acc_read8<C * 8>(v);
  *reinterpret_cast<f32x4*>(wp + ((int64_t)(2 * C) * 512 + tid) * 4) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(wp + ((int64_t)(2 * C + 1) * 512 + tid) * 4) = f32x4{v[4], v[5], v[6], v[7]};
  if constexpr (C + 1 < 16) tail_store_chunk<C + 1>(wp, tid);
}
// accumulators = partial[0] + partial[1] + ... + partial[S-1] in THIS order whichever workgroup arrives last (its own
// partial is taken from its registers): the result does not depend on the arrival order
template <int C> __device__ __forceinline__ void tail_combine_chunk(const float* wt, int tid, int own, int S) {
  float t[8], x[8];
  auto get = [&](int sp, float* o) {
    if (sp == own) { acc_read8<C * 8>(o); return; }
    const float* p = wt + (int64_t)sp * (128 * 512);
    const f32x4 a = *reinterpret_cast<const f32x4*>(p + ((int64_t)(2 * C) * 512 + tid) * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + ((int64_t)(2 * C + 1) * 512 + tid) * 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  };
  get(0, t);
  for (int sp = 1; sp < S; ++sp) {
    get(sp, x);
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] += x[k];
  }
  acc_write8<C * 8>(t);
  if constexpr (C + 1 < 16) tail_combine_chunk<C + 1>(wt, tid, own, S);
}

// LDS-DMA pieces, issued from INLINE ASM (hidden from hipcc's waitcnt pass, see gemm_bf16_p8.hip).  Nothing here has a
// VGPR destination: an asm load into registers lets hipcc copy those registers before the data has landed.
//   s_nop 4: SGPR base written by SALU -> read by VMEM; s_nop 0: M0 written by SALU -> read by the LDS-DMA.
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
__device__ __forceinline__ void dma16(const void* base_uniform, uint32_t off, uint32_t lds0) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base_uniform), "s"(lds0) : "memory");
}
__device__ __forceinline__ void dma4(const void* base_uniform, uint32_t off, uint32_t lds0) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(off), "s"(base_uniform), "s"(lds0) : "memory");
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

// 16-byte store of whole lines, issued from inline asm (not on hipcc's vmcnt scoreboard: the waits between the halves are
// counted by hand); s_nop 1: the data registers are read after issue
__device__ __forceinline__ void store16_nt(const void* base_uniform, uint32_t off, const u32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(off), "v"(v), "s"(base_uniform) : "memory");
}

// ---- output path.  Accumulator layout (swapped MFMA): block (I, ri, J) = a[((I*2+ri)*2+J)*16 ..+15]; lane l holds row
// I*128 + wr*64 + ri*32 + (l&31), register r column J*128 + wc*32 + 8*(r>>2) + 4*(l>>5) + (r&3).
// bf16 patch of a quadrant: [128 rows][256 B]; 16-byte chunk c of row R is stored at chunk c ^ (R&7), and its two 8-byte
// halves are swapped when (R>>3)&1: the 16 lanes a ds_write_b64 services together (16 consecutive rows, one column group)
// hit 16 different 8-byte slots of the 128-byte bank window.
// uint8 patch of a quadrant: [128 rows][128 B]; chunk c at c ^ (R&7) and (patches written by ds_write_b32) dword d of a
// chunk at d ^ ((R>>3)&3): 32 rows x one dword -> 32 different banks.
template <int MODE, int I, int J, int RI>
__device__ __forceinline__ void pq_block(const PQArgs& g, lds_char* sm, const f32x4 (&bias)[4], int li, int lk, int wr,
                                         int wc, int si_off) {
  float v[16];
  acc_read8<((I * 2 + RI) * 2 + J) * 16>(v);
  acc_read8<((I * 2 + RI) * 2 + J) * 16 + 8>(v + 8);
  const uint32_t hbase = H_OFF + J * 32768 + (uint32_t)(wr * 64 + RI * 32 + li) * 256 + ((((uint32_t)wc * 4) ^ (li & 4)) << 4) +
                         ((lk ^ ((li >> 3) & 1)) << 3);
  const uint32_t row128 = (uint32_t)(wr * 64 + RI * 32 + li) * 128;
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    float w[4];
    // uint8 patches: byte column wc*32 + 8 gq + 4 lk -> chunk wc*2 + (gq>>1), dword (gq&1)*2 + lk
    const uint32_t c8 = ((uint32_t)wc * 2 + (gq >> 1)) ^ (li & 7);
    if constexpr (MODE == PQ_DACT8) {
      const uint32_t q = *reinterpret_cast<lds_cu32*>(sm + si_off + J * 16384 + row128 + (c8 << 4) + (((gq & 1) * 2 + lk) << 2));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float qf = (float)((q >> (8 * k)) & 0xffu);   // v_cvt_f32_ubyteN
        w[k] = v[gq * 4 + k] * (qf * (1.0f / AUX8_SCALE) - AUX8_OFF);
      }
    } else if constexpr (MODE == PQ_RES) {   // bias and residual join in the row pass
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = v[gq * 4 + k];
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = v[gq * 4 + k] + bias[gq][k];
    }
    if constexpr (MODE == PQ_ACT8E) {
      // erf-GELU and its derivative on pairs: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7: one rcp, one exp2 - the
      // exponential the derivative's Gaussian needs anyway - and a degree-5 polynomial), Phi(x) = 1 - q / 2 (x >= 0) | q / 2
      uint32_t q = 0;
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const v2f x = v2f{w[2 * k2], w[2 * k2 + 1]};
        const v2f ax = v2f{__builtin_fabsf(x[0]), __builtin_fabsf(x[1])};
        const v2f td = ax * v2f{0.2316418882f, 0.2316418882f} + v2f{1.0f, 1.0f};          // 1 + 0.3275911 |x| / sqrt 2
        const v2f t = v2f{__builtin_amdgcn_rcpf(td[0]), __builtin_amdgcn_rcpf(td[1])};
        const v2f ee = (x * v2f{-0.7213475204f, -0.7213475204f}) * x;                      // -x^2 / 2 * log2 e
        const v2f E = v2f{__builtin_amdgcn_exp2f(ee[0]), __builtin_amdgcn_exp2f(ee[1])};   // exp(-x^2 / 2)
        v2f pl = t * v2f{1.061405429f, 1.061405429f} + v2f{-1.453152027f, -1.453152027f};
        pl = pl * t + v2f{1.421413741f, 1.421413741f};
        pl = pl * t + v2f{-0.284496736f, -0.284496736f};
        pl = pl * t + v2f{0.254829592f, 0.254829592f};
        const v2f hq = ((pl * t) * E) * v2f{0.5f, 0.5f};                                   // (1 - erf(|x| / sqrt 2)) / 2
        const v2f up = v2f{1.0f, 1.0f} - hq;
        const v2f phi = v2f{x[0] >= 0.f ? up[0] : hq[0], x[1] >= 0.f ? up[1] : hq[1]};
        const v2f y = x * phi;
        const v2f sd = (x * E) * v2f{0.3989422804f, 0.3989422804f} + phi;                  // act'(u) = Phi + x phi(x)
        const v2f qf = sd * v2f{AUX8_SCALE, AUX8_SCALE} + v2f{AUX8_OFF * AUX8_SCALE, AUX8_OFF * AUX8_SCALE};
        w[2 * k2] = y[0]; w[2 * k2 + 1] = y[1];
        q = __builtin_amdgcn_cvt_pk_u8_f32(qf[0], 2 * k2, q);
        q = __builtin_amdgcn_cvt_pk_u8_f32(qf[1], 2 * k2 + 1, q);
      }
      *reinterpret_cast<lds_u32*>(sm + AO_OFF + J * 16384 + row128 + (c8 << 4) +
                                  ((((gq & 1) * 2 + lk) ^ ((li >> 3) & 3)) << 2)) = q;
    }
    if constexpr (MODE == PQ_ACT8) {
      // QuickGELU and its derivative on PAIRS (v_pk_mul / v_pk_add / v_pk_fma_f32: the same IEEE operations as the scalar
      // form of sigmoid1702 / act_with_side, two elements per instruction; exp2 and rcp stay scalar): the epilogue of the
      // c_fc forward is VALU work in a phase the matrix pipe idles in
      uint32_t q = 0;
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const v2f x = v2f{w[2 * k2], w[2 * k2 + 1]};
        const v2f e = x * v2f{-2.4554669595930157f, -2.4554669595930157f};
        const v2f d = v2f{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + v2f{1.0f, 1.0f};
        const v2f sg = v2f{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        const v2f t = (x * v2f{1.702f, 1.702f}) * (v2f{1.0f, 1.0f} - sg) + v2f{1.0f, 1.0f};
        const v2f sd = sg * t;                                   // act'(u) = s (1 + 1.702 u (1 - s))
        const v2f y = x * sg;
        const v2f qf = sd * v2f{AUX8_SCALE, AUX8_SCALE} + v2f{AUX8_OFF * AUX8_SCALE, AUX8_OFF * AUX8_SCALE};
        w[2 * k2] = y[0]; w[2 * k2 + 1] = y[1];
        q = __builtin_amdgcn_cvt_pk_u8_f32(qf[0], 2 * k2, q);
        q = __builtin_amdgcn_cvt_pk_u8_f32(qf[1], 2 * k2 + 1, q);
      }
      *reinterpret_cast<lds_u32*>(sm + AO_OFF + J * 16384 + row128 + (c8 << 4) +
                                  ((((gq & 1) * 2 + lk) ^ ((li >> 3) & 3)) << 2)) = q;
    }
    u32x2 p;
    p[0] = pack2bf(w[0], w[1]);
    p[1] = pack2bf(w[2], w[3]);
    *reinterpret_cast<lds_u32x2*>(sm + hbase + ((gq ^ (li & 3)) << 4)) = p;
  }
}

// row pass of one half (two quadrants): wave w moves 64 rows x 64 columns (= one 128-byte line per row) of quadrant
// J = w>>2: rows ((w>>1)&1)*64 .., columns (w&1)*64 ..; 8 rows per instruction.  A lane keeps ONE logical 16-byte chunk
// over its 8 rows (the physical chunk c ^ (R&7) depends on lane>>3 only), so the column sums of the 64 rows are wave-local.
template <int MODE, int I>
__device__ __forceinline__ void pq_rows_half(const PQArgs& g, lds_cchar* sm, int lane, int wave, int64_t m0, int64_t n0) {
  const int J = wave >> 2, rh = (wave >> 1) & 1, ch = wave & 1;
  const int rs = lane >> 3, cl = lane & 7;
  const int c = ch * 8 + cl;                                                    // logical chunk: columns c*8 .. +7
  const uint32_t rd = H_OFF + J * 32768 + (uint32_t)(rh * 64 + rs) * 256 + ((uint32_t)(c ^ rs) << 4);
  const int64_t ldcb = g.ldc * 2;
  const char* cq = reinterpret_cast<const char*>(g.C + (m0 + I * 128 + rh * 64) * g.ldc + n0 + J * 128);   // uniform
  const uint32_t go = (uint32_t)(rs * ldcb + c * 16);
  float cs[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) cs[k] = 0.f;
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    u32x4 d[4];
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) d[i4] = *reinterpret_cast<lds_cu32x4*>(sm + rd + (h2 * 4 + i4) * 2048);
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const int it = h2 * 4 + i4;                                               // rows it*8 + rs: bit 3 of the row = it & 1
      if (it & 1) d[i4] = u32x4{d[i4][2], d[i4][3], d[i4][0], d[i4][1]};
      if (MODE == PQ_DACT8 && g.colsum_part != nullptr) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          cs[2 * k] += __uint_as_float(d[i4][k] << 16);
          cs[2 * k + 1] += __uint_as_float(d[i4][k] & 0xffff0000u);
        }
      }
      if (!(g.abl & 1)) store16_nt(cq + (int64_t)it * 8 * ldcb, go, d[i4]);
    }
  }
  if (MODE == PQ_DACT8 && g.colsum_part != nullptr) {   // column sums of the stored (rounded) values over this wave's 64 rows
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float x = cs[k];
      x += __shfl_xor(x, 8, 64);
      x += __shfl_xor(x, 16, 64);
      x += __shfl_xor(x, 32, 64);
      cs[k] = x;
    }
    if (rs == 0) {
      float* dst = g.colsum_part + ((m0 + I * 128 + rh * 64) >> 6) * g.N + n0 + J * 128 + c * 8;
      *reinterpret_cast<f32x4*>(dst) = f32x4{cs[0], cs[1], cs[2], cs[3]};
      *reinterpret_cast<f32x4*>(dst + 4) = f32x4{cs[4], cs[5], cs[6], cs[7]};
    }
  }
}
// PQ_RES: the bf16 residual rows of the wave's row-pass patch (the store mapping of pq_rows_half), 8 x 16 bytes per lane
// and half, fetched with ordinary loads long before they are used (the registers of the operand fragments are free)
template <int I>
__device__ __forceinline__ void pq_res_issue(const PQArgs& g, int lane, int wave, int64_t m0, int64_t n0, u32x4 (&res)[8]) {
  const int J = wave >> 2, rh = (wave >> 1) & 1, ch = wave & 1;
  const int rs = lane >> 3, c = ch * 8 + (lane & 7);
  const int64_t ldsb = g.lds * 2;
  const char* rq = reinterpret_cast<const char*>(g.side) + ((m0 + I * 128 + rh * 64 + rs) * g.lds + n0 + J * 128 + c * 8) * 2;
#pragma unroll
  for (int it = 0; it < 8; ++it) res[it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(rq + (int64_t)it * 8 * ldsb));
}
// row pass of one half with the residual (and the bias) added: out = bf16(parked bf16(acc) + bias + residual)
template <int I>
__device__ __forceinline__ void pq_rows_half_res(const PQArgs& g, lds_cchar* sm, int lane, int wave, int64_t m0, int64_t n0,
                                                 const u32x4 (&res)[8], const float (&bv)[8]) {
  const int J = wave >> 2, rh = (wave >> 1) & 1, ch = wave & 1;
  const int rs = lane >> 3, cl = lane & 7;
  const int c = ch * 8 + cl;
  const uint32_t rd = H_OFF + J * 32768 + (uint32_t)(rh * 64 + rs) * 256 + ((uint32_t)(c ^ rs) << 4);
  const int64_t ldcb = g.ldc * 2;
  const char* cq = reinterpret_cast<const char*>(g.C + (m0 + I * 128 + rh * 64) * g.ldc + n0 + J * 128);   // uniform
  const uint32_t go = (uint32_t)(rs * ldcb + c * 16);
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    u32x4 d[4];
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) d[i4] = *reinterpret_cast<lds_cu32x4*>(sm + rd + (h2 * 4 + i4) * 2048);
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const int it = h2 * 4 + i4;
      if (it & 1) d[i4] = u32x4{d[i4][2], d[i4][3], d[i4][0], d[i4][1]};
      u32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float lo = __uint_as_float(d[i4][k] << 16) + __uint_as_float(res[it][k] << 16) + bv[2 * k];
        const float hi = __uint_as_float(d[i4][k] & 0xffff0000u) + __uint_as_float(res[it][k] & 0xffff0000u) + bv[2 * k + 1];
        o[k] = pack2bf(lo, hi);
      }
      if (!(g.abl & 1)) store16_nt(cq + (int64_t)it * 8 * ldcb, go, o);
    }
  }
}
// uint8 saved-derivative patches of one half -> global: wave w moves rows (w&3)*32 .. +31 of quadrant J = w>>2
template <int I>
__device__ __forceinline__ void pq_rows_aux(const PQArgs& g, lds_cchar* sm, int lane, int wave, int64_t m0, int64_t n0) {
  const int J = wave >> 2, rb = (wave & 3) * 32;
  const int rs = lane >> 3, pc = lane & 7;
  const uint32_t rd = AO_OFF + J * 16384 + (uint32_t)rb * 128 + (uint32_t)lane * 16;
  const char* aq = reinterpret_cast<const char*>(g.aux + (m0 + I * 128 + rb) * g.ldaux + n0 + J * 128);   // uniform
  const uint32_t go = (uint32_t)(rs * g.ldaux + ((pc ^ rs) << 4));
  u32x4 d[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) d[it] = *reinterpret_cast<lds_cu32x4*>(sm + rd + it * 1024);
#pragma unroll
  for (int it = 0; it < 4; ++it) {   // rows rb + it*8 + rs: (R>>3)&3 = it -> logical dword j sits at j ^ it
    const u32x4 o = u32x4{d[it][0 ^ it], d[it][1 ^ it], d[it][2 ^ it], d[it][3 ^ it]};
    if (!(g.abl & 1)) store16_nt(aq + (int64_t)it * 8 * g.ldaux, go, o);
  }
}

// ---- fp32 output (PQ_SLAB: weight gradients, raw split-K partial tiles).  fp32 patch of a quadrant: [128 rows][512 B];
// 16-byte chunk c (4 columns) of row R is stored at chunk c ^ (R & 15): the 16 lanes a ds_write_b128 services together
// (16 consecutive rows, one column group) hit the 16 different 16-byte slots of a 256-byte bank window, and so do the 16
// lanes of a ds_read_b128 of the row pass (16 consecutive chunks of one row).  Two quadrants = the whole 128-KiB ring.
template <int I, int J, int RI>
__device__ __forceinline__ void pq_block_f32(lds_char* sm, int li, int lk, int wr, int wc) {
  float v[16];
  acc_read8<((I * 2 + RI) * 2 + J) * 16>(v);
  acc_read8<((I * 2 + RI) * 2 + J) * 16 + 8>(v + 8);
  const uint32_t row = (uint32_t)(wr * 64 + RI * 32 + li);
  const uint32_t base = J * 65536 + row * 512;
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const uint32_t chunk = (uint32_t)wc * 8 + 2 * gq + lk;
    *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(sm + base + ((chunk ^ (row & 15)) << 4)) =
        f32x4{v[gq * 4], v[gq * 4 + 1], v[gq * 4 + 2], v[gq * 4 + 3]};
  }
}
__device__ __forceinline__ void store16(const void* base_uniform, uint32_t off, const u32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(off), "v"(v), "s"(base_uniform) : "memory");
}
// row pass: wave w moves rows (w&3)*32 .. +31 of quadrant J = w>>2, two whole 512-byte rows per instruction (16 instructions;
// Q = 0 / 1: the first / second 8 of them).  RES: out = (parked acc + bias) + residual, all fp32 - the arithmetic and the
// rounding of the 8-phase kernel's fp32-residual epilogue
template <int I, int Q, bool RES>
__device__ __forceinline__ void pq_rows_q_f32(const PQArgs& g, float* out, lds_cchar* sm, int lane, int wave, int64_t m0,
                                              int64_t n0, const f32x4 (&res)[8], const f32x4& bias4) {
  const int J = wave >> 2, rb = (wave & 3) * 32;
  const int r2 = lane >> 5, c = lane & 31;
  const uint32_t rd = J * 65536 + (uint32_t)(rb + r2) * 512;
  const uint32_t cx = (uint32_t)(c ^ r2);                      // (R & 15) = ((2 it) & 15) | r2 for R = rb + 2 it + r2
  const int64_t ldcb = g.ldc * 4;
  const char* cq = reinterpret_cast<const char*>(out + (m0 + I * 128 + rb) * g.ldc + n0 + J * 128);   // uniform
  const uint32_t go = (uint32_t)(r2 * ldcb + c * 16);
#pragma unroll
  for (int h4 = 0; h4 < 2; ++h4) {
    f32x4 d[4];
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const int it = Q * 8 + h4 * 4 + i4;
      d[i4] = *reinterpret_cast<lds_cf32x4*>(sm + rd + it * 1024 + ((cx ^ (uint32_t)((2 * it) & 15)) << 4));
    }
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const int it = Q * 8 + h4 * 4 + i4;
      if constexpr (RES) d[i4] = (d[i4] + bias4) + res[h4 * 4 + i4];
      if (!(g.abl & 1)) store16(cq + (int64_t)it * 2 * ldcb, go, __builtin_bit_cast(u32x4, d[i4]));
    }
  }
}
// PQ_RES32: the fp32 residual rows of instructions Q*8 .. +7 of the row pass above (ordinary loads, issued early)
template <int I, int Q>
__device__ __forceinline__ void pq_res32_issue(const PQArgs& g, int lane, int wave, int64_t m0, int64_t n0, f32x4 (&res)[8]) {
  const int J = wave >> 2, rb = (wave & 3) * 32;
  const int r2 = lane >> 5, c = lane & 31;
  if (g.rmod > 0) {   // broadcast table: its rows are re-read by every sample - ordinary (cached) loads
    const float* rt = reinterpret_cast<const float*>(g.side) + n0 + J * 128 + c * 4;
    const int64_t row0 = m0 + I * 128 + rb + r2;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      res[k] = *reinterpret_cast<const f32x4*>(rt + ((row0 + (Q * 8 + k) * 2) % g.rmod) * g.lds);
    return;
  }
  const float* rq = reinterpret_cast<const float*>(g.side) + (m0 + I * 128 + rb + r2) * g.lds + n0 + J * 128 + c * 4;
#pragma unroll
  for (int k = 0; k < 8; ++k) res[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rq + (int64_t)(Q * 8 + k) * 2 * g.lds));
}

// unit_given >= 0 (grouped launch): this workgroup's unit inside the problem described by g
template <bool A_KS, bool B_KS, int MODE>
__device__ __forceinline__ void pq_main(const PQArgs& g, const int unit_given) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  int lane = tid & 63;

  // workgroup b runs on XCD b % 8: the tile sequence (row-major, columns fastest) is cut into 8 contiguous chunks, one
  // per XCD, so that the tiles sharing an A row slab / B column slab sit behind one L2 (as in gemm_bf16_p8.hip)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  // PQ_SLAB (split-K): the (K range, tile) units are ordered K-range-major, so that the tiles of ONE K range - which share
  // its operand rows - sit behind one L2
  int unit = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  if (MODE == PQ_SLAB && unit_given >= 0) unit = unit_given;
  // tail split: workgroups nfull .. are the K ranges of the tail tiles; the tail_S ranges of a tile share bid & 7 (one XCD)
  const bool has_tail = MODE != PQ_SLAB && g.tail_S > 1;
  bool tail = false;
  int ttile = 0, tsplit = 0;
  if (has_tail) {
    if (bid >= g.nfull) {
      const int j = bid - g.nfull, qq = j >> 3;
      ttile = (qq / g.tail_S) * 8 + (j & 7);
      tsplit = qq % g.tail_S;
      if (ttile >= g.tail_r) return;
      tail = true;
      unit = g.nfull + ttile;
    } else {
      const int qf = g.nfull >> 3;               // nfull is a multiple of 256
      unit = xcd * qf + (bid >> 3);
    }
  }
  // half-tile tail: workgroups nfull .. are the 128-row halves of the tail tiles; the half sequence (tile-major) is cut into
  // 8 contiguous chunks, one per XCD, like the full tiles in front of it
  constexpr bool HALF_OK = MODE != PQ_SLAB;
  bool half = false;
  int hsel = 0;
  if (HALF_OK && !A_KS && (g.half_r > 0 || g.half_x > 0)) {
    if (bid >= g.nfull) {
      const int j = bid - g.nfull, n2 = 2 * g.half_r + g.half_x, q8h = n2 >> 3, r8h = n2 & 7, xh = j & 7;
      const int hu = (xh < r8h ? xh * (q8h + 1) : r8h * (q8h + 1) + (xh - r8h) * q8h) + (j >> 3);
      if (hu < 2 * g.half_r) {
        unit = g.nfull + (hu >> 1);
        hsel = hu & 1;
      } else {
        unit = g.nfull + g.half_r + (hu - 2 * g.half_r);   // tile row M / 256: only its upper 128 rows exist
      }
      half = true;
    } else {
      const int q8f = g.nfull >> 3, r8f = g.nfull & 7;
      unit = (xcd < r8f ? xcd * (q8f + 1) : r8f * (q8f + 1) + (xcd - r8f) * q8f) + (bid >> 3);
    }
  }
  const int ksplit = MODE == PQ_SLAB ? unit / g.ntiles : 0;
  const int tile = MODE == PQ_SLAB ? unit - ksplit * g.ntiles : unit;
  const int tcol = tile % g.nbx, trow = tile / g.nbx;
  const int64_t m0 = (int64_t)trow * BT + hsel * 128, n0 = (int64_t)tcol * BT;
  const int nkt = g.K / BK;
  const int tk0 = tail ? tsplit * nkt / g.tail_S : 0;
  const int64_t kb = MODE == PQ_SLAB ? (int64_t)ksplit * g.kper : (int64_t)tk0 * BK;
  const char* baseA = reinterpret_cast<const char*>(A_KS ? g.A + m0 + kb * g.lda : g.A + m0 * g.lda + kb);
  const char* baseB = reinterpret_cast<const char*>(B_KS ? g.B + n0 + kb * g.ldb : g.B + n0 * g.ldb + kb);
  const int nk = MODE == PQ_SLAB ? (int)((g.K - kb < g.kper ? g.K - kb : g.kper) / BK)
                                 : (tail ? (tsplit + 1) * nkt / g.tail_S - tk0 : nkt);
  const int64_t stepA = A_KS ? (int64_t)BK * g.lda * 2 : BK * 2;
  const int64_t stepB = B_KS ? (int64_t)BK * g.ldb * 2 : BK * 2;
  const uint32_t lds0 = (uint32_t)(uintptr_t)((lds_void*)smem);

  // ---- side operands that travel during the K loop (oldest entries of the VM queue: the first counted wait covers them)
  constexpr bool HAS_BIAS = MODE == PQ_PLAIN || pq_is_act8<MODE>();   // bias added in the accumulator layout (PQ_RES: in the row pass)
  const bool has_bias = HAS_BIAS && g.bias != nullptr;
  if (has_bias)   // this wave's 2 x 32 columns: lane l -> column (l>>5)*128 + wc*32 + (l&31)
    dma4(g.bias + n0 + wc * 32, (uint32_t)(((lane >> 5) * 128 + (lane & 31)) * 4), lds0 + (half ? BIAS_OFF_H : BIAS_OFF) + wave * 256);
  // PQ_DACT8: saved-derivative bytes of one half (2 quadrants x [128 rows][128 B]) = 32 pieces of 1 KiB (8 rows each), 4 per
  // wave; lane l of piece p: row (p&15)*8 + (l>>3), physical chunk l&7 <- logical chunk (l&7) ^ (row&7)
  auto side_half = [&](int I, uint32_t dst) {
    const char* sb = reinterpret_cast<const char*>(g.side) + (m0 + I * 128) * g.lds + n0;   // uniform
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = wave * 4 + i, J = p >> 4, r = (p & 15) * 8 + (lane >> 3);
      dma16(sb, (uint32_t)(r * g.lds + J * 128 + (((lane & 7) ^ (r & 7)) << 4)), lds0 + dst + p * 1024);
    }
  };
  if constexpr (MODE == PQ_DACT8) { if (!half) side_half(0, SI0_OFF); }   // half-tile: the 3-slot ring covers SI0_OFF, fetched after the K loop

  uint32_t offA[2][2], offB[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      offA[h][i] = A_KS ? off_ks(h, wave * 2 + i, lane, g.lda) : off_direct(h, wave * 2 + i, lane, g.lda);
      offB[h][i] = B_KS ? off_ks(h, wave * 2 + i, lane, g.ldb) : off_direct(h, wave * 2 + i, lane, g.ldb);
    }
  const uint32_t lds_ring = lds0 + wave * 2048;
  lds_cchar* const sm3 = (lds_cchar*)smem;
  bf16x8_t fa[2][4], fbx[4], fby[4];
  using std::integral_constant;
  typedef integral_constant<int, 0> I0;
  typedef integral_constant<int, 1> I1;
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


  if (HALF_OK && !A_KS && half) {
    // ---- half-tile K loop: a 128 x 256 output (A0 only), two phases per K-tile - L1: A0(t), B0(t) fragments, stage A0(t+2);
    // L2: B1(t) fragments, stage B0(t+2), B1(t+2) - on a 3-slot ring (tile t in slot t % 3), so that a unit travels for two
    // K-tiles (8 barrier intervals) as in the full-tile schedule.  A unit of tile t-1 is overwritten two intervals after the
    // lagging wave group consumed it.  VM queue per wave: ... B1(t) | A0(t+1) B0(t+1) B1(t+1) A0(t+2) | -> vmcnt(8) retires B1(t)
    // at the end of L1(t); ... A0(t+1) B0(t+1) | B1(t+1) A0(t+2) B0(t+2) B1(t+2) -> vmcnt(8) retires both at the end of L2(t).
    if constexpr (HALF_OK && !A_KS) {
      lds_cchar* abh[3][4];
      lds_cchar* bbh[3][4];
#pragma unroll
      for (int sl = 0; sl < 3; ++sl)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          uint32_t oa = fragbase_direct(wr * 64, x, lane);
          uint32_t ob = B_KS ? fragbase_ks(wc * 32, lane) : fragbase_direct(wc * 32, x, lane);
          oa += sl * HSLOT; ob += sl * HSLOT + UNIT;
          asm volatile("" : "+v"(oa));
          if ((!B_KS || x < 1)) asm volatile("" : "+v"(ob));
          abh[sl][x] = sm3 + oa;
          bbh[sl][x] = sm3 + ob;
        }
      typedef integral_constant<int, 2> I2;
      auto stage_h = [&](int u, int t, int sl) {   // u: 0 = A0, 1 = B0, 2 = B1
        const uint32_t dst = lds_ring + sl * HSLOT + u * UNIT;
        if (u == 0) dma16x2(baseA + (int64_t)t * stepA, offA[0][0], offA[0][1], dst);
        else dma16x2(baseB + (int64_t)t * stepB, offB[u - 1][0], offB[u - 1][1], dst);
      };
      auto read_a_h = [&](auto sc) {
        constexpr int SL = decltype(sc)::value;
        fa[0][0] = frag_direct_at<0>(abh[SL][0]); fa[0][1] = frag_direct_at<0>(abh[SL][1]);
        fa[0][2] = frag_direct_at<0>(abh[SL][2]); fa[0][3] = frag_direct_at<0>(abh[SL][3]);
        fa[1][0] = frag_direct_at<4096>(abh[SL][0]); fa[1][1] = frag_direct_at<4096>(abh[SL][1]);
        fa[1][2] = frag_direct_at<4096>(abh[SL][2]); fa[1][3] = frag_direct_at<4096>(abh[SL][3]);
      };
      auto read_b_h = [&](auto sc, auto uc, bf16x8_t (&fb)[4]) {
        constexpr int SL = decltype(sc)::value, U = decltype(uc)::value;
        if constexpr (B_KS) {
          fb[0] = frag_ks_at<U + 0 * 4096>(bbh[SL][0]); fb[1] = frag_ks_at<U + 1 * 4096>(bbh[SL][0]);
          fb[2] = frag_ks_at<U + 2 * 4096>(bbh[SL][0]); fb[3] = frag_ks_at<U + 3 * 4096>(bbh[SL][0]);
        } else {
          fb[0] = frag_direct_at<U>(bbh[SL][0]); fb[1] = frag_direct_at<U>(bbh[SL][1]);
          fb[2] = frag_direct_at<U>(bbh[SL][2]); fb[3] = frag_direct_at<U>(bbh[SL][3]);
        }
      };
      stage_h(1, 0, 0);
      stage_h(0, 0, 0);
      stage_h(2, 0, 0);
      if (nk > 1) {
        stage_h(1, 1, 1);
        stage_h(0, 1, 1);
        stage_h(2, 1, 1);
        wait_vm<8>();    // B0(0), A0(0) have landed
      } else {
        wait_vm<2>();
      }
      PQ_BAR();
      if (wr == 1) PQ_BAR();
      auto ktile_h = [&](auto sc, auto zc, int t) {
        typedef decltype(sc) SL;
        typedef integral_constant<int, (SL::value + 2) % 3> NS;
        typedef decltype(zc) Z;
        const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
        // ---- phase 1: quadrant (A0, B0)
        read_a_h(SL{});
        read_b_h(SL{}, integral_constant<int, 0>{}, fbx);
        if (n2) { stage_h(0, t + 2, NS::value); wait_vm<8>(); } else if (n1) { wait_vm<6>(); } else { wait_vm<0>(); }   // retires B1(t)
        PQ_BAR();
        quadrant(I0{}, I0{}, Z{}, fbx);
        PQ_BAR();
        // ---- phase 2: quadrant (A0, B1)
        read_b_h(SL{}, integral_constant<int, UNIT>{}, fby);
        if (n2) { stage_h(1, t + 2, NS::value); stage_h(2, t + 2, NS::value); wait_vm<8>(); } else if (n1) { wait_vm<2>(); }   // retires A0(t+1), B0(t+1)
        PQ_BAR();
        quadrant(I0{}, I1{}, Z{}, fby);
        PQ_BAR();
      };
      ktile_h(I0{}, I1{}, 0);
      int t = 1;
      while (t < nk) {
        ktile_h(I1{}, I0{}, t);
        if (++t >= nk) break;
        ktile_h(I2{}, I0{}, t);
        if (++t >= nk) break;
        ktile_h(I0{}, I0{}, t);
        ++t;
      }
    }
  } else {
  lds_cchar* abase[2][4];
  lds_cchar* bbase[2][4];
#pragma unroll
  for (int bf = 0; bf < 2; ++bf)
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      // k-strided A: one base per 32-row block (x = 0, 1), the k chunk is an immediate; direct A: one base per k chunk
      uint32_t oa = A_KS ? fragbase_ks(wr * 64 + (x & 1) * 32, lane) : fragbase_direct(wr * 64, x, lane);
      uint32_t ob = B_KS ? fragbase_ks(wc * 32, lane) : fragbase_direct(wc * 32, x, lane);
      oa += bf * BUF; ob += bf * BUF;
      if (!A_KS || x < 2) asm volatile("" : "+v"(oa));
      if ((!B_KS || x < 1)) asm volatile("" : "+v"(ob));
      abase[bf][x] = sm3 + oa;
      bbase[bf][x] = sm3 + ob;
    }

  auto stage = [&](int u, int t) {
    const uint32_t dst = lds_ring + (t & 1) * BUF + u * UNIT;
    if (u < 2) dma16x2(baseA + (int64_t)t * stepA, offA[u][0], offA[u][1], dst);
    else dma16x2(baseB + (int64_t)t * stepB, offB[u - 2][0], offB[u - 2][1], dst);
  };

  auto read_a = [&](auto bfc, auto uc) {
    constexpr int BF_ = decltype(bfc)::value, U = decltype(uc)::value;
    if constexpr (A_KS) {
      fa[0][0] = frag_ks_at<U + 0 * 4096>(abase[BF_][0]); fa[0][1] = frag_ks_at<U + 1 * 4096>(abase[BF_][0]);
      fa[0][2] = frag_ks_at<U + 2 * 4096>(abase[BF_][0]); fa[0][3] = frag_ks_at<U + 3 * 4096>(abase[BF_][0]);
      fa[1][0] = frag_ks_at<U + 0 * 4096>(abase[BF_][1]); fa[1][1] = frag_ks_at<U + 1 * 4096>(abase[BF_][1]);
      fa[1][2] = frag_ks_at<U + 2 * 4096>(abase[BF_][1]); fa[1][3] = frag_ks_at<U + 3 * 4096>(abase[BF_][1]);
    } else {
      fa[0][0] = frag_direct_at<U>(abase[BF_][0]); fa[0][1] = frag_direct_at<U>(abase[BF_][1]);
      fa[0][2] = frag_direct_at<U>(abase[BF_][2]); fa[0][3] = frag_direct_at<U>(abase[BF_][3]);
      fa[1][0] = frag_direct_at<U + 4096>(abase[BF_][0]); fa[1][1] = frag_direct_at<U + 4096>(abase[BF_][1]);
      fa[1][2] = frag_direct_at<U + 4096>(abase[BF_][2]); fa[1][3] = frag_direct_at<U + 4096>(abase[BF_][3]);
    }
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
  // prologue: K-tile 0 and, of K-tile 1, everything but A1 (issued in R1 of K-tile 0)
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
  PQ_BAR();
  read_b(I0{}, integral_constant<int, 2 * UNIT>{}, fbx);
  if (wr == 1) PQ_BAR();     // group 1 runs one barrier interval behind group 0

  // one K-tile: fbp holds B0(t) on entry and fbq is free; on exit fbq holds B0(t+1)
  auto ktile = [&](auto bc, auto zc, int t, bf16x8_t (&fbp)[4], bf16x8_t (&fbq)[4]) {
    typedef decltype(bc) CB;
    typedef integral_constant<int, 1 - CB::value> NB;
    typedef decltype(zc) Z;
    const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
    // ---- phase 1: quadrant (A0, B0)
    read_a(CB{}, integral_constant<int, 0>{});
    if (n1) { stage(1, t + 1); wait_vm<10>(); } else { wait_vm<2>(); }          // retires B1(t), read in R2
    PQ_BAR();
    quadrant(I0{}, I0{}, Z{}, fbp);
    PQ_BAR();
    // ---- phase 2: quadrant (A0, B1)
    read_b(CB{}, integral_constant<int, 3 * UNIT>{}, fbq);
    if (n2) { stage(2, t + 2); wait_vm<10>(); } else if (n1) { wait_vm<8>(); } else { wait_vm<0>(); }   // retires A1(t)
    PQ_BAR();
    quadrant(I0{}, I1{}, Z{}, fbq);
    PQ_BAR();
    // ---- phase 3: quadrant (A1, B1)
    read_a(CB{}, integral_constant<int, UNIT>{});
    if (n2) { stage(0, t + 2); wait_vm<10>(); } else if (n1) { wait_vm<6>(); }   // retires B0(t+1), read in R4
    PQ_BAR();
    quadrant(I1{}, I1{}, Z{}, fbq);
    PQ_BAR();
    // ---- phase 4: quadrant (A1, B0); B1's registers are free: B0(t+1) goes there
    if (n1) read_b(NB{}, integral_constant<int, 2 * UNIT>{}, fbq);
    if (n2) { stage(3, t + 2); wait_vm<10>(); } else if (n1) { wait_vm<4>(); }   // retires A0(t+1), read in R1
    PQ_BAR();
    quadrant(I1{}, I0{}, Z{}, fbp);
    PQ_BAR();
  };
  ktile(I0{}, I1{}, 0, fbx, fby);     // first K-tile: the MFMAs overwrite the accumulators (C = 0)
  int t = 1;
  for (; t + 1 < nk; t += 2) {
    ktile(I1{}, I0{}, t, fby, fbx);
    ktile(I0{}, I0{}, t + 1, fbx, fby);
  }
  if (t < nk) ktile(I1{}, I0{}, t, fby, fbx);
  }
  if (wr == 0) PQ_BAR();     // group 0 catches up: every wave is done with the operand ring, the VM queue is empty

  if (g.abl & 2) return;
  if (has_tail && tail) {
    // partial tile -> workspace; the last workgroup of the tile to arrive sums the tail_S partials and goes on to the epilogue
    asm volatile("" : "+v"(tid));
    float* const wt = g.tail_ws + (int64_t)ttile * g.tail_S * (128 * 512);
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");     // last MFMA -> v_accvgpr_read
    tail_store_chunk<0>(wt + (int64_t)tsplit * (128 * 512), tid);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");     // EVERY wave: its partial is visible device-wide before it arrives
    __syncthreads();
    int* const flag = reinterpret_cast<int*>(smem);        // the operand ring is free by now
    if (tid == 0) *flag = __hip_atomic_fetch_add(g.tail_cnt + ttile, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int arrived = *reinterpret_cast<volatile int*>(flag);
    __syncthreads();
    if (arrived != g.tail_S - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    tail_combine_chunk<0>(wt, tid, tsplit, g.tail_S);
  }
  // ---- output
  asm volatile("" : "+v"(lane));   // lane-dependent epilogue addresses are derived here, not kept live across the K loop
  const int li = lane & 31, lk = lane >> 5;
  lds_char* const smw = (lds_char*)smem;
  if constexpr (MODE == PQ_SLAB || MODE == PQ_RES32) {
    // fp32 output through fp32 patches: the partial tile of this K range -> its slab (PQ_SLAB), or
    // (acc + bias) + fp32 residual -> fp32 (PQ_RES32: out_proj / c_proj on the fp32 residual stream)
    constexpr bool RES = MODE == PQ_RES32;
    float* const out = g.Cf + (int64_t)ksplit * g.slab_stride;
    f32x4 ra[8], rb8[8], bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (RES) {
      pq_res32_issue<0, 0>(g, lane, wave, m0, n0, ra);
      pq_res32_issue<0, 1>(g, lane, wave, m0, n0, rb8);
      if (g.bias != nullptr) bias4 = *reinterpret_cast<const f32x4*>(g.bias + n0 + (wave >> 2) * 128 + (lane & 31) * 4);
    }
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#define PQ_BLOCKS32(I)                                                   \
    do {                                                                 \
      pq_block_f32<I, 0, 0>(smw, li, lk, wr, wc);                        \
      pq_block_f32<I, 0, 1>(smw, li, lk, wr, wc);                        \
      pq_block_f32<I, 1, 0>(smw, li, lk, wr, wc);                        \
      pq_block_f32<I, 1, 1>(smw, li, lk, wr, wc);                        \
      PQ_BAR_LDS();                                                      \
    } while (0)
    PQ_BLOCKS32(0);
    pq_rows_q_f32<0, 0, RES>(g, out, sm3, lane, wave, m0, n0, ra, bias4);
    if constexpr (RES) { if (!half) pq_res32_issue<1, 0>(g, lane, wave, m0, n0, ra); }   // the second half's rows travel under the rest
    pq_rows_q_f32<0, 1, RES>(g, out, sm3, lane, wave, m0, n0, rb8, bias4);
    if (HALF_OK && half) return;                                            // half-tile workgroup: 128 rows only
    if constexpr (RES) pq_res32_issue<1, 1>(g, lane, wave, m0, n0, rb8);
    PQ_BAR_LDS();
    PQ_BLOCKS32(1);
    pq_rows_q_f32<1, 0, RES>(g, out, sm3, lane, wave, m0, n0, ra, bias4);
    pq_rows_q_f32<1, 1, RES>(g, out, sm3, lane, wave, m0, n0, rb8, bias4);
#undef PQ_BLOCKS32
    return;
  }
  if constexpr (MODE == PQ_DACT8) {
    if (half) { side_half(0, SI1_OFF); wait_vm<0>(); PQ_BAR(); }   // half-tile: its only side-in, into ring space (exposed once per tail workgroup)
    else side_half(1, SI1_OFF);                                     // lands under the first half's arithmetic
  }
  f32x4 bias[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bias[j][q] = has_bias ? *reinterpret_cast<lds_cf32x4*>(sm3 + (half ? BIAS_OFF_H : BIAS_OFF) + wave * 256 + (j * 32 + q * 8 + lk * 4) * 4)
                            : f32x4{0.f, 0.f, 0.f, 0.f};
  // MFMA results -> v_accvgpr_read: the last MFMA was issued a barrier ago; 16-pass XDL needs 18 wait states
  asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");

  // PQ_RES: residual rows (both halves) and the bias of the row-pass chunk in registers
  u32x4 res0[8], res1[8];
  float bv[8];
  if constexpr (MODE == PQ_RES) {
    pq_res_issue<0>(g, lane, wave, m0, n0, res0);
    const int cb = (wave >> 2) * 128 + ((wave & 1) * 8 + (lane & 7)) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) bv[k] = g.bias != nullptr ? g.bias[n0 + cb + k] : 0.f;
  }
#define PQ_HALF(I, SI, RESV)                                                   \
  do {                                                                         \
    pq_block<MODE, I, 0, 0>(g, smw, bias[0], li, lk, wr, wc, SI);              \
    pq_block<MODE, I, 0, 1>(g, smw, bias[0], li, lk, wr, wc, SI);              \
    pq_block<MODE, I, 1, 0>(g, smw, bias[1], li, lk, wr, wc, SI);              \
    pq_block<MODE, I, 1, 1>(g, smw, bias[1], li, lk, wr, wc, SI);              \
    PQ_BAR_LDS();                                                              \
    if constexpr (MODE == PQ_RES) {                                            \
      if (I == 0 && !half) pq_res_issue<1>(g, lane, wave, m0, n0, res1);       \
      pq_rows_half_res<I>(g, sm3, lane, wave, m0, n0, RESV, bv);               \
    } else {                                                                   \
      pq_rows_half<MODE, I>(g, sm3, lane, wave, m0, n0);                       \
    }                                                                          \
    if constexpr (pq_is_act8<MODE>()) pq_rows_aux<I>(g, sm3, lane, wave, m0, n0); \
  } while (0)
  PQ_HALF(0, (MODE == PQ_DACT8 && half ? SI1_OFF : SI0_OFF), res0);
  if (HALF_OK && half) return;   // half-tile workgroup: 128 rows only
  // second half: its side-in pieces (4 per wave) are older than the first half's 8 output stores of this wave
  if constexpr (MODE == PQ_DACT8) { if (g.abl & 1) wait_vm<0>(); else wait_vm<8>(); }
  PQ_BAR_LDS();   // the patches are free again (and every wave's side-in pieces have landed)
  PQ_HALF(1, SI1_OFF, res1);
#undef PQ_HALF
}

template <bool A_KS, bool B_KS, int MODE>
__global__ __launch_bounds__(NWV * 64) void gemm_bf16_pq_kernel(PQArgs g) {
  pq_main<A_KS, B_KS, MODE>(g, -1);
}

// grouped weight gradients: the XCD-chunked unit sequence runs over ALL problems (a chunk = a few K ranges of a few problems)
__global__ __launch_bounds__(NWV * 64) void gemm_bf16_pq_group_kernel(PQArgs g, PQGroup grp) {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int unit = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  int p = 0;
  while (p + 1 < grp.nprob && unit >= grp.prob[p].unit_end) ++p;
  const int u0 = p ? grp.prob[p - 1].unit_end : 0;
  g.A = grp.prob[p].A; g.B = grp.prob[p].B; g.Cf = grp.prob[p].Cf;
  g.slab_stride = grp.prob[p].slab_stride;
  g.lda = grp.prob[p].lda; g.ldb = grp.prob[p].ldb; g.ldc = grp.prob[p].ldc;
  g.nbx = grp.prob[p].nbx; g.ntiles = grp.prob[p].ntiles;
  pq_main<true, true, PQ_SLAB>(g, unit - u0);
}

}  // namespace

// The file is compiled once per PQ_PART (build.sh): 0 = forward layout, 1 = data-gradient layout, 2 = host dispatcher,
// 3 = weight-gradient layout
#ifndef PQ_PART
#error "compile with -DPQ_PART=0..3 (see build.sh)"
#endif
#define PQ_LAUNCH(AKS, BKS, MODE) hipLaunchKernelGGL((gemm_bf16_pq_kernel<AKS, BKS, MODE>), grid, dim3(NWV * 64), 0, stream, g)
#if PQ_PART == 0
void segclip_pq_launch_f(int mode, dim3 grid, hipStream_t stream, const void* args) {   // forward: plain / QuickGELU + saved derivative / + residual
  const PQArgs g = *reinterpret_cast<const PQArgs*>(args);
  if (mode == PQ_ACT8) PQ_LAUNCH(false, false, PQ_ACT8);
  else if (mode == PQ_ACT8E) PQ_LAUNCH(false, false, PQ_ACT8E);
  else if (mode == PQ_RES) PQ_LAUNCH(false, false, PQ_RES);
  else if (mode == PQ_RES32) PQ_LAUNCH(false, false, PQ_RES32);
  else PQ_LAUNCH(false, false, PQ_PLAIN);
}
#elif PQ_PART == 1
void segclip_pq_launch_k(int mode, dim3 grid, hipStream_t stream, const void* args) {   // data gradient: plain / x saved derivative
  const PQArgs g = *reinterpret_cast<const PQArgs*>(args);
  if (mode == PQ_DACT8) PQ_LAUNCH(false, true, PQ_DACT8);
  else PQ_LAUNCH(false, true, PQ_PLAIN);
}
#elif PQ_PART == 3
void segclip_pq_launch_w(int mode, dim3 grid, hipStream_t stream, const void* args) {   // weight gradient: both operands k-strided, fp32 slabs
  const PQArgs g = *reinterpret_cast<const PQArgs*>(args);
  (void)mode;
  PQ_LAUNCH(true, true, PQ_SLAB);
}
void segclip_pq_launch_group(dim3 grid, hipStream_t stream, const void* args, const void* group) {
  const PQArgs g = *reinterpret_cast<const PQArgs*>(args);
  const PQGroup grp = *reinterpret_cast<const PQGroup*>(group);
  hipLaunchKernelGGL(gemm_bf16_pq_group_kernel, grid, dim3(NWV * 64), 0, stream, g, grp);
}
#endif

#if PQ_PART == 2
// Tail split plan of a launch with `ntiles` full tiles and nkt K-tiles: r = tiles of the last partial round of 256 workgroups,
// S = K ranges per tail tile (0 = no tail split).  The idea: 588 tiles (N = 768 at M = 50176) are 2.3 rounds of 256 CUs and
// take the time of 3; with the 76 tail tiles cut into 3 K ranges the third round would cost a third of the K loop + the
// exchange.  MEASURED (MI355X, round 4, tools/bench_pq.py with SEGCLIP_PQ_TAIL=0/2/3; results equal to the unsplit kernel up
// to the order of the partial sums, deterministic): SLOWER - K = 3072: 220 -> 248 us (S = 2) / 271 us (S = 3), K = 2304:
// 162 -> 203 / 229 us, the training step 42.1 -> 43.5 / 44.3 ms.  The exchange needs an agent-scope release (buffer_wbl2 sc1:
// the XCD's L2 writes back every dirty output line) per arriving workgroup and an acquire (buffer_inv sc1) in the last one,
// whose 2 x 256 KiB of partials then arrive at one CU's load rate.  OFF by default (SEGCLIP_PQ_TAIL=2..4 enables it for A/B).
static int pq_tail_plan(int64_t ntiles, int64_t nkt, int* r_out) {
  static const int env = [] { const char* e = segclip_tuning_env("SEGCLIP_PQ_TAIL"); return e ? atoi(e) : 0; }();
  static const int min_nkt = [] { const char* e = segclip_tuning_env("SEGCLIP_PQ_TAIL_MINK"); return e ? atoi(e) : 24; }();
  *r_out = 0;
  if (!env || ntiles <= 256) return 0;
  const int r = (int)(ntiles % 256);
  if (r == 0 || r > 128) return 0;
  int S = 256 / r;
  if (S > 4) S = 4;
  if (env > 1 && S > env) S = env;      // SEGCLIP_PQ_TAIL = 2..4 caps the number of ranges
  if (S < 2 || nkt < min_nkt || nkt / S < 4) return 0;
  *r_out = r;
  return S;
}
// Half-tile tail plan: the r = ntiles % 256 tiles of the last, partial round run as 2 r workgroups of 128 x 256 when those fit
// one round (r <= 128): 591 tiles (N = 768 at M = 50432) are 2.3 rounds of 256 CUs and take the time of 3; the third round
// then costs a half-tile's time (about 0.6 of a tile's: 3 of 4 operand units per K-tile, half the MFMAs, half the output).
// No exchange between workgroups, results bit-identical to the full tiles'.  SEGCLIP_PQ_HALF=0 switches it off (A/B).
static bool pq_half_on() {
  static const int env = [] { const char* e = segclip_tuning_env("SEGCLIP_PQ_HALF"); return e ? atoi(e) : 1; }();
  return (env == 2 ? [] { const char* e = segclip_tuning_env("SEGCLIP_PQ_HALF_NOW"); return e ? atoi(e) : 1; }() : env) != 0;
}
static int pq_half_plan(int64_t ntiles) {
  if (!pq_half_on() || ntiles <= 256) return 0;
  const int r = (int)(ntiles % 256);
  return r > 0 && r <= 128 ? r : 0;
}
// the number of tail tiles a launch over `ntiles` 256 x 256 tiles runs as half-tiles (0 = none); its grid is ntiles + that
extern "C" int segclip_gemm_pq_half_tail(int64_t ntiles) { return ntiles > 0 ? pq_half_plan(ntiles) : 0; }
static size_t pq_tail_ws_bytes(int r, int S) { return 4096 + (size_t)r * S * 128 * 512 * sizeof(float); }
// workspace the tail split of this descriptor wants (0 = none): counters (4 KiB) + partial tiles
size_t segclip_gemm_bf16_pq_tail_ws_bytes(const segclip_gemm_desc* d) {
  if (d->a_dtype != SEGCLIP_BF16 || d->b_dtype != SEGCLIP_BF16 || d->sak != 1) return 0;
  if (d->M % BT != 0 || d->N % BT != 0 || d->K % BK != 0 || d->K < BK) return 0;
  if ((d->nb1 > 1) || (d->nb2 > 1)) return 0;
  int r = 0;
  const int S = pq_tail_plan((d->M / BT) * (d->N / BT), d->K / BK, &r);
  return S ? pq_tail_ws_bytes(r, S) : 0;
}
// fills the tail-split fields of g when the caller's workspace allows it; zeroes the arrival counters on the stream
static bool pq_tail_setup(PQArgs& g, const segclip_gemm_desc* d, hipStream_t stream, unsigned* nwg, bool allow_half) {
  int r = 0;
  const int S = pq_tail_plan(g.ntiles, g.K / BK, &r);
  *nwg = (unsigned)g.ntiles;
  const int hx = d->M % BT != 0 ? g.nbx : 0;      // M = 256 q + 128 (checked by the caller): one more row of half-tiles
  if (hx && !pq_half_on()) return false;
  if ((!S || hx) && allow_half) {
    const int hr = pq_half_plan(g.ntiles);
    if (hr > 0 || hx > 0) { g.half_r = hr; g.half_x = hx; g.nfull = g.ntiles - hr; *nwg = (unsigned)(g.nfull + 2 * hr + hx); }
    return true;
  }
  if (hx) return false;
  if (!S || d->ws == nullptr || (size_t)d->ws_bytes < pq_tail_ws_bytes(r, S) || (reinterpret_cast<uintptr_t>(d->ws) & 15) != 0) return true;
  if (hipMemsetAsync(d->ws, 0, 4096, stream) != hipSuccess) return false;
  g.tail_S = S; g.tail_r = r; g.nfull = g.ntiles - r;
  g.tail_cnt = reinterpret_cast<int*>(d->ws);
  g.tail_ws = reinterpret_cast<float*>(reinterpret_cast<char*>(d->ws) + 4096);
  *nwg = (unsigned)(g.nfull + 8 * S * ((r + 7) / 8));
  return true;
}

void segclip_pq_launch_f(int, dim3, hipStream_t, const void*);
void segclip_pq_launch_k(int, dim3, hipStream_t, const void*);
void segclip_pq_launch_w(int, dim3, hipStream_t, const void*);

// Launch this kernel when the problem meets its preconditions (see the top of the file); false = not taken.
bool segclip_gemm_bf16_pq_try(const segclip_gemm_desc* d, const void* args_, int splits, int64_t nb, hipStream_t stream) {
  // SEGCLIP_GEMM_PQ: 1 (default) on, 0 off, 2 = consult SEGCLIP_GEMM_PQ_NOW (0/1) at every call (A/B tests in one process)
  static const int mode_env = [] { const char* e = segclip_tuning_env("SEGCLIP_GEMM_PQ"); return e ? atoi(e) : 1; }();
  if (mode_env == 0) return false;
  if (mode_env == 2) { const char* e = segclip_tuning_env("SEGCLIP_GEMM_PQ_NOW"); if (e && atoi(e) == 0) return false; }
  const Args& a = *reinterpret_cast<const Args*>(args_);
  const bool a_ks = d->sak != 1, b_ks = d->sbk != 1;
  if (nb != 1 || d->a_dtype != SEGCLIP_BF16 || d->b_dtype != SEGCLIP_BF16) return false;
  // M = 256 q + 128 (the token rows of an odd multiple of 128 samples x 197 / 577 / 77 tokens): the last 128 rows run as a
  // row of half-tiles (forward / data-gradient layouts; the weight gradient's M is a weight dimension)
  if (d->M % (a_ks ? BT : 128) != 0 || d->M < 128 || d->N % BT != 0 || d->K % BK != 0 || d->K < BK) return false;
  if (a_ks) {   // weight gradient: C(m,n) = sum_k A[k][m] B[k][n], fp32 out (split-K: raw partial tiles into the slabs)
    static const int wg_env = [] { const char* e = segclip_tuning_env("SEGCLIP_GEMM_PQ_WGRAD"); return e ? atoi(e) : 1; }();
    if (!wg_env || !b_ks || d->c_dtype != SEGCLIP_F32) return false;
    if (d->bias || d->residual || d->aux || d->act != SEGCLIP_ACT_NONE || d->mul_dact || a.colsum_part) return false;
    if (splits == 1 && (d->alpha != 1.0f || d->ldc % 4 != 0 || (reinterpret_cast<uintptr_t>(d->C) & 15) != 0)) return false;
    if (splits > 1 && (a.slab == nullptr || a.kper % BK != 0 || d->N % 4 != 0 || (reinterpret_cast<uintptr_t>(a.slab) & 15) != 0)) return false;
    if ((reinterpret_cast<uintptr_t>(d->A) & 15) != 0 || (reinterpret_cast<uintptr_t>(d->B) & 15) != 0) return false;
    if (d->sak % 8 != 0 || d->sbk % 8 != 0) return false;
    if (64 * d->sak * 2 >= (int64_t)1 << 31 || 64 * d->sbk * 2 >= (int64_t)1 << 31) return false;
    PQArgs g = {};
    g.A = reinterpret_cast<const bf16_t*>(d->A); g.B = reinterpret_cast<const bf16_t*>(d->B);
    g.lda = d->sak; g.ldb = d->sbk;
    g.N = (int)d->N; g.K = (int)d->K; g.nbx = (int)(d->N / BT); g.ntiles = (int)((d->M / BT) * (d->N / BT));
    if (splits > 1) { g.Cf = a.slab; g.ldc = d->N; g.kper = a.kper; g.slab_stride = d->M * d->N; }
    else { g.Cf = reinterpret_cast<float*>(d->C); g.ldc = d->ldc; g.kper = d->K; g.slab_stride = 0; }
    if (256 * g.ldc * 4 >= (int64_t)1 << 31) return false;
    g.abl = mode_env == 2 ? segclip_ablation_env("SEGCLIP_PQ_ABL") : 0;
    segclip_pq_launch_w(PQ_SLAB, dim3((unsigned)(g.ntiles * splits)), stream, &g);
    return true;
  }
  if (splits != 1) return false;
  if (d->res_row_mod > 0 && !(d->c_dtype == SEGCLIP_F32 && d->residual != nullptr && d->r_dtype == SEGCLIP_F32)) return false;
  if (d->c_dtype == SEGCLIP_F32) {   // forward + fp32 residual -> fp32 (the fp32 residual stream): fp32 patches, fp32 row pass
    static const int r32_env = [] { const char* e = segclip_tuning_env("SEGCLIP_GEMM_PQ_RES32"); return e ? atoi(e) : 1; }();
    if (!r32_env || b_ks || d->residual == nullptr || d->r_dtype != SEGCLIP_F32 || d->alpha != 1.0f) return false;
    if (d->aux || d->act != SEGCLIP_ACT_NONE || d->mul_dact || a.colsum_part) return false;
    auto al16 = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!al16(d->A) || !al16(d->B) || !al16(d->C) || !al16(d->bias) || !al16(d->residual)) return false;
    if (d->sam % 8 != 0 || d->sbn % 8 != 0 || d->ldc % 4 != 0 || d->ldr % 4 != 0) return false;
    if (256 * d->sam * 2 >= (int64_t)1 << 31 || 256 * d->sbn * 2 >= (int64_t)1 << 31 || 256 * d->ldc * 4 >= (int64_t)1 << 31) return false;
    PQArgs g = {};
    g.A = reinterpret_cast<const bf16_t*>(d->A); g.B = reinterpret_cast<const bf16_t*>(d->B);
    g.bias = d->bias; g.side = d->residual; g.lds = d->ldr; g.rmod = d->res_row_mod > 0 ? d->res_row_mod : 0;
    g.lda = d->sam; g.ldb = d->sbn; g.Cf = reinterpret_cast<float*>(d->C); g.ldc = d->ldc;
    g.N = (int)d->N; g.K = (int)d->K; g.nbx = (int)(d->N / BT); g.ntiles = (int)((d->M / BT) * (d->N / BT));
    g.abl = mode_env == 2 ? segclip_ablation_env("SEGCLIP_PQ_ABL") : 0;
    unsigned nwg = 0;
    if (!pq_tail_setup(g, d, stream, &nwg, true)) return false;
    segclip_pq_launch_f(PQ_RES32, dim3(nwg), stream, &g);
    return true;
  }
  if (d->c_dtype != SEGCLIP_BF16) return false;
  if (d->alpha != 1.0f) return false;
  int mode = PQ_PLAIN;
  if (d->residual != nullptr) {   // + bf16 residual -> bf16 (the bf16 residual stream of the towers): forward layout only
    if (d->r_dtype != SEGCLIP_BF16 || b_ks || d->mul_dact || d->act != SEGCLIP_ACT_NONE || d->aux || d->ldr % 8 != 0) return false;
    mode = PQ_RES;
  } else
  if (d->mul_dact) {            // x act'(u), saved as one byte; optional fused column sums
    const bool gelu = d->act == SEGCLIP_ACT_QUICK_GELU || d->act == SEGCLIP_ACT_GELU_ERF;   // (the byte is decoded the same way)
    if (!(d->aux && d->aux_kind == 2 && gelu && !d->bias && b_ks && d->ldaux % 16 == 0)) return false;
    mode = PQ_DACT8;
  } else if (d->act != SEGCLIP_ACT_NONE) {   // QuickGELU / erf-GELU + saved derivative as one byte
    const bool gelu = d->act == SEGCLIP_ACT_QUICK_GELU || d->act == SEGCLIP_ACT_GELU_ERF;
    if (!(d->aux && d->aux_kind == 2 && gelu && !b_ks && d->ldaux % 16 == 0)) return false;
    mode = d->act == SEGCLIP_ACT_QUICK_GELU ? PQ_ACT8 : PQ_ACT8E;
  } else if (d->aux) {
    return false;
  }
  if (a.colsum_part != nullptr && mode != PQ_DACT8) return false;
  auto al = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!al(d->A) || !al(d->B) || !al(d->C) || !al(d->bias) || !al(d->aux) || !al(d->residual) || d->ldc % 8 != 0) return false;
  const int64_t lda = d->sam, ldb = b_ks ? d->sbk : d->sbn;
  if (lda % 8 != 0 || ldb % 8 != 0) return false;
  // 32-bit per-lane offsets: 256 rows (64 k-rows) of an operand and 256 rows of an output stay below 2 GiB
  if (256 * lda * 2 >= (int64_t)1 << 31 || (b_ks ? 64 : 256) * ldb * 2 >= (int64_t)1 << 31) return false;
  if (256 * d->ldc * 2 >= (int64_t)1 << 31 || 256 * d->ldaux >= (int64_t)1 << 31) return false;
  PQArgs g = {};
  g.A = reinterpret_cast<const bf16_t*>(d->A); g.B = reinterpret_cast<const bf16_t*>(d->B);
  g.C = reinterpret_cast<bf16_t*>(d->C); g.bias = d->bias;
  g.side = mode == PQ_DACT8 ? d->aux : (mode == PQ_RES ? d->residual : nullptr);
  g.aux = (mode == PQ_ACT8 || mode == PQ_ACT8E) ? reinterpret_cast<uint8_t*>(d->aux) : nullptr;
  g.colsum_part = a.colsum_part;
  g.lda = lda; g.ldb = ldb; g.ldc = d->ldc; g.lds = mode == PQ_RES ? d->ldr : d->ldaux; g.ldaux = d->ldaux;
  g.N = (int)d->N; g.K = (int)d->K; g.nbx = (int)(d->N / BT); g.ntiles = (int)((d->M / BT) * (d->N / BT));
  g.abl = mode_env == 2 ? segclip_ablation_env("SEGCLIP_PQ_ABL") : 0;
  unsigned nwg = 0;
  if (!pq_tail_setup(g, d, stream, &nwg, true)) return false;
  (b_ks ? segclip_pq_launch_k : segclip_pq_launch_f)(mode, dim3(nwg), stream, &g);
  return true;
}
void segclip_pq_launch_group(dim3, hipStream_t, const void*, const void*);

// K ranges for a group of `tiles` 256x256 output tiles over `ksteps` 64-row K steps: the count that minimises the modelled time
//   ceil(tiles * s / 256) rounds x (K loop of ksteps / s steps at ~1.63 us - the k-strided operand layout, fitted to
//   profiles/r04_wgrad_group.txt - + ~5 us of prologue / output) + the fp32 partial tiles
//   written and read back when s > 1 (~0.075 us each, device-wide)
extern "C" double segclip_wgrad_group_model_us(int64_t tiles, int64_t ksteps, int splits) {
  if (tiles < 1 || ksteps < 1 || splits < 1) return 0.0;
  const int64_t per = cdiv(ksteps, splits);
  const double rounds = (double)cdiv(tiles * splits, 256);
  return rounds * (per * 1.63 + 5.0) + (splits > 1 ? tiles * splits * 0.075 : 0.0);
}
extern "C" int segclip_wgrad_group_splits(int64_t tiles, int64_t ksteps) {
  if (tiles < 1 || ksteps < 1) return 1;
  int best = 1;
  double bt = 0.0;
  for (int s = 1; s <= 32 && (s == 1 || s <= ksteps / 8); ++s) {
    if ((s - 1) * cdiv(ksteps, s) >= ksteps) continue;        // an empty last range
    const double t = segclip_wgrad_group_model_us(tiles, ksteps, s);
    if (s == 1 || t < bt) { bt = t; best = s; }
  }
  return best;
}
extern "C" size_t segclip_wgrad_group_ws_bytes(const segclip_wgrad_item* it, int n, int splits) {
  if (splits <= 1) return 0;
  size_t total = 0;
  for (int i = 0; i < n; ++i) total += (size_t)splits * it[i].M * it[i].N * sizeof(float);
  return total;
}
// dw_i (M_i, N_i) fp32 = dy_i^T x_i for n problems over the same R token rows, as one launch of the weight-gradient kernel.
// splits == 1: written to dw_i directly.  splits > 1: K range s of problem i is left as a raw partial tile set at
// ws + (sum_{j<i} splits*M_j*N_j + s*M_i*N_i) floats and the CALLER combines them (segclip_reduce_multi, kind slabs) - the
// order of the partial sums is fixed, so results are reproducible.  SEGCLIP_ERR_UNSUPPORTED (nothing launched) unless every
// problem has bf16 operands on 16-byte boundaries, M_i and N_i multiples of 256 and leading dimensions multiples of 8.
extern "C" int segclip_wgrad_group(const segclip_wgrad_item* it, int n, int64_t R, int splits, void* ws, size_t ws_bytes,
                                   void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 1 || n > PQ_GROUP_MAX || R % BK != 0 || R < BK || splits < 1) { segclip_set_error("wgrad_group: unsupported group"); return SEGCLIP_ERR_UNSUPPORTED; }
  const int64_t per = cdiv(R / BK, splits) * BK;
  if ((int64_t)(splits - 1) * per >= R) { segclip_set_error("wgrad_group: empty K range"); return SEGCLIP_ERR_UNSUPPORTED; }
  if (splits > 1 && (ws == nullptr || ws_bytes < segclip_wgrad_group_ws_bytes(it, n, splits) || (reinterpret_cast<uintptr_t>(ws) & 15) != 0)) {
    segclip_set_error("wgrad_group: workspace too small");
    return SEGCLIP_ERR_INVALID;
  }
  PQGroup grp = {};
  grp.nprob = n;
  int64_t units = 0;
  float* slab = reinterpret_cast<float*>(ws);
  for (int i = 0; i < n; ++i) {
    const segclip_wgrad_item& q = it[i];
    const bool ok = q.M > 0 && q.N > 0 && q.M % BT == 0 && q.N % BT == 0 && q.ld_dy % 8 == 0 && q.ld_x % 8 == 0 &&
                    (reinterpret_cast<uintptr_t>(q.dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(q.x) & 15) == 0 &&
                    64 * q.ld_dy * 2 < (int64_t)1 << 31 && 64 * q.ld_x * 2 < (int64_t)1 << 31 &&
                    (splits > 1 || (q.dw != nullptr && q.ld_dw % 4 == 0 && (reinterpret_cast<uintptr_t>(q.dw) & 15) == 0 &&
                                    256 * q.ld_dw * 4 < (int64_t)1 << 31));
    if (!ok) { segclip_set_error("wgrad_group: problem %d is not covered (256-multiples, 16-byte alignment)", i); return SEGCLIP_ERR_UNSUPPORTED; }
    PQProb& p = grp.prob[i];
    p.A = reinterpret_cast<const bf16_t*>(q.dy); p.B = reinterpret_cast<const bf16_t*>(q.x);
    p.lda = (int)q.ld_dy; p.ldb = (int)q.ld_x;
    p.nbx = (int)(q.N / BT); p.ntiles = (int)((q.M / BT) * (q.N / BT));
    if (splits > 1) { p.Cf = slab; p.ldc = (int)q.N; p.slab_stride = q.M * q.N; slab += (int64_t)splits * q.M * q.N; }
    else { p.Cf = q.dw; p.ldc = (int)q.ld_dw; p.slab_stride = 0; }
    units += (int64_t)p.ntiles * splits;
    p.unit_end = (int)units;
  }
  PQArgs g = {};
  g.K = (int)R; g.kper = per;
  g.abl = 0;
  segclip_pq_launch_group(dim3((unsigned)units), stream, &g, &grp);
  SEGCLIP_CHECK_LAUNCH("wgrad_group");
  return 0;
}
#endif  // PQ_PART == 2
