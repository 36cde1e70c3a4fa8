// Multi-head attention core, head_dim <= 64, arbitrary (batch, token) strides for Q/K/V/O.
//
// bf16 path (flash style, never materialises the score matrix in HBM):
//   forward : block = (b, h, group of <=8 query tiles of 32 rows), one wave per query tile.  K/V of a
//             256-key chunk are staged once in LDS (swizzled 128-B rows).  Each wave computes the
//             *transposed* score tile S^T = K Q^T with v_mfma_f32_32x32x16_bf16, so a lane owns one
//             query row: the online-softmax max/sum are 16 in-lane ops + one cross-half shuffle.  P^T is
//             fed straight back as the B operand of O^T = V^T P^T; the V^T fragment comes from the LDS
//             transpose read ds_read_b64_tr_b16 with the key order permuted to match the accumulator
//             layout, so P never leaves registers.
//   backward: block = (b, h); Q, K, V, dO tiles in LDS (<=256 tokens each).  Pass A (wave owns a key
//             tile) accumulates dK, dV; pass B (wave owns a query tile) accumulates dQ.  S and dP are
//             recomputed in both passes (no atomics, deterministic).
// f32 path (exact parity mode): S = scale*Q K^T, row softmax, O = P V as three launches of the f32 MFMA
//   GEMM + a softmax kernel; P is kept for backward (5 GEMMs + one elementwise kernel).
#include <stdlib.h>

#include "common.h"

int segclip_gemm_f32_launch(const segclip_gemm_desc* d, hipStream_t stream);

namespace {

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int ROWB = 128;  // bytes per LDS row (64 bf16)

// swizzled byte offset of element (row, col) in a [rows][64] bf16 LDS tile.
// f = rotr3((row>>1)&7): ds_read_b128 of 16 rows is conflict-free, and the 4-row tr16 reads of a
// 32-lane half land in disjoint banks.
__device__ __forceinline__ int swz(int row, int col) {
  const int t = (row >> 1) & 7;
  const int f = ((t & 1) << 2) | (t >> 1);
  return row * ROWB + ((((col >> 3) ^ f)) << 4) + ((col & 7) << 1);
}

// MFMA operand fragment, k-contiguous rows: lane l -> row rbase + (l&31), cols kc*16 + 8*(l>>5) .. +7
__device__ __forceinline__ bf16x8_t frag_rows(const char* t, int rbase, int kc, int lane) {
  return *reinterpret_cast<const bf16x8_t*>(t + swz(rbase + (lane & 31), kc * 16 + 8 * (lane >> 5)));
}
// MFMA operand fragment of the TRANSPOSED tile: lane l -> "row" = column cbase + (l&31) of the tile,
// k = tile rows  rbase + 4*(l>>5) + {0,1,2,3, 8,9,10,11}   (the key order of the 32x32 accumulator)
__device__ __forceinline__ bf16x8_t frag_tr(const char* t, int rbase, int cbase, int lane) {
  const int g4 = lane >> 4, q = lane & 15;
  const int row = rbase + 4 * (g4 >> 1) + (q >> 2);
  const int col = cbase + 16 * (g4 & 1) + 4 * (q & 3);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(t + swz(row, col)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(t + swz(row + 8, col)));
  s16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, r);
}
typedef float v2f __attribute__((ext_vector_type(2)));
// the same fragment from precomputed lane offsets (rows r0 .. r0+3 and r0+8 .. r0+11 of a 16-row group)
__device__ __forceinline__ bf16x8_t frag_tr_at(const char* t, int off_lo, int off_hi) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(t + off_lo));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(t + off_hi));
  s16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(bf16x8_t, r);
}
__device__ __forceinline__ bf16x8_t pack8(const float* p) {
  u32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = pack2bf(p[2 * j], p[2 * j + 1]);
  return __builtin_bit_cast(bf16x8_t, r);
}
// accumulator row index of register r for half h (32x32 tile)
__device__ __forceinline__ int accrow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// stage `nrows_pad` rows (zero-filled beyond nvalid / beyond hd) of X[row*st + d] into a swizzled tile.
// Loads are unconditional from clamped addresses and issued in batches of 4 before any LDS write (a load
// inside a divergent branch is waited for individually by hipcc and the staging becomes latency-serial).
__device__ __forceinline__ void stage_rows(char* tile, const bf16_t* __restrict__ X, int64_t st, int row0, int nvalid,
                                           int nrows_pad, int hd, int tid, int nthreads) {
  const int total = nrows_pad * 8;
  for (int base = 0; base < total; base += 4 * nthreads) {
    u32x4 v[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = base + it * nthreads + tid;
      const int r = idx >> 3, c = idx & 7;
      const int rc = r < nvalid ? r : nvalid - 1;
      const int cc = c * 8 < hd ? c : 0;
      v[it] = *reinterpret_cast<const u32x4*>(X + (int64_t)(row0 + rc) * st + cc * 8);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = base + it * nthreads + tid;
      const int r = idx >> 3, c = idx & 7;
      if (idx < total) {
        const bool ok = r < nvalid && c * 8 < hd;
        *reinterpret_cast<u32x4*>(tile + swz(r, c * 8)) = ok ? v[it] : u32x4{0u, 0u, 0u, 0u};
      }
    }
  }
}

struct FwdArgs {
  const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O; float* lse;
  int H, Tq, Tk, hd;
  int64_t q_sb, q_st, k_sb, k_st, v_sb, v_st, o_sb, o_st;
  float scale; int causal;
  const int* klen;  // nullable: valid keys per sample
  int staged;       // bf16 forward: output rows through LDS (SEGCLIP_ATTN_FWD_STAGED)
  int lean;         // bf16 forward: the lean softmax step on unmasked key tiles (SEGCLIP_ATTN_FWD_LEAN, default on)
};

constexpr int KCHUNK = 256;

// 4 waves per SIMD (two 7-wave workgroups per CU): at most 128 VGPRs
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_fwd_bf16_kernel(FwdArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * KCHUNK * ROWB];
  char* Kt = smem;
  char* Vt = smem + KCHUNK * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
  const int q0 = (blockIdx.x * nw + wave) * 32;
  const int li = lane & 31, lh = lane >> 5;
  const bf16_t* Qp = a.Q + (int64_t)b * a.q_sb + (int64_t)h * a.hd;
  const bf16_t* Kp = a.K + (int64_t)b * a.k_sb + (int64_t)h * a.hd;
  const bf16_t* Vp = a.V + (int64_t)b * a.v_sb + (int64_t)h * a.hd;
  const int qg = q0 + li;
  const bool wave_active = q0 < a.Tq;
  const int kvalid = a.klen ? (a.klen[b] < a.Tk ? a.klen[b] : a.Tk) : a.Tk;
  const float sc2 = a.scale * 1.4426950408889634f;  // scale * log2(e): the running max m is kept in the log2 domain

  bf16x8_t qf[4];
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    const int d = kc * 16 + 8 * lh;
    const bool ok = qg < a.Tq && d < a.hd;
    const u32x4 raw = *reinterpret_cast<const u32x4*>(Qp + (int64_t)(qg < a.Tq ? qg : a.Tq - 1) * a.q_st + (d < a.hd ? d : 0));
    const u32x4 v = ok ? raw : u32x4{0u, 0u, 0u, 0u};
    qf[kc] = __builtin_bit_cast(bf16x8_t, v);
  }
  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  float m = -INFINITY, l = 0.f;
  // lane-constant LDS offsets of the K fragments (k chunk kc) and of the four 8-row groups of the transposed V fragments
  // (V: rows r0 and r0 + 8 of a 32-key tile per 32-column half; rows + 16 / + 24 are the same offsets + 2048 bytes, since
  // the swizzle only looks at bits 1-3 of the row)
  int koff[4], voff[2][2];
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) koff[kc] = swz(li, kc * 16 + 8 * lh);
  {
    const int g4 = lane >> 4, q = lane & 15;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int j = 0; j < 2; ++j) voff[dt][j] = swz(4 * (g4 >> 1) + (q >> 2) + 8 * j, dt * 32 + 16 * (g4 & 1) + 4 * (q & 3));
  }

  for (int c0 = 0; c0 < a.Tk; c0 += KCHUNK) {
    const int nvalid = a.Tk - c0 < KCHUNK ? a.Tk - c0 : KCHUNK;
    const int npad = (nvalid + 31) & ~31;
    __syncthreads();
    stage_rows(Kt, Kp, a.k_st, c0, nvalid, npad, a.hd, tid, blockDim.x);
    stage_rows(Vt, Vp, a.v_st, c0, nvalid, npad, a.hd, tid, blockDim.x);
    __syncthreads();
    if (!wave_active) continue;
    // Unmasked key tiles (all but the last one of a vision sequence) take a leaner softmax step (round 4; PMC round 3: 27 VALU
    // instructions per MFMA, MFMA busy 14.7 %): the row maximum is taken over the RAW scores (the scale is positive), so that
    // scale and shift are one fused multiply-add per score (packed: two scores per instruction); the exponentials are
    // summed pairwise and packed to bf16 as they are produced; the rescale of the 32 output accumulators is skipped when no
    // row's maximum moved (wave-uniform test); and the lane-dependent LDS offsets are computed once per kernel
    // (swz(kt + r, c) = kt * ROWB + swz(r, c) for kt a multiple of 16): ~60 VALU instructions per tile instead of ~130.
    int kt = 0;
    while (kt < npad) {
      if (a.causal && c0 + kt > q0 + 31) break;
      const bool edge = c0 + kt + 32 > kvalid || (a.causal && c0 + kt + 31 > q0);
      if (!edge && a.lean) {
        const char* kb = Kt + kt * ROWB;
        const char* vb = Vt + kt * ROWB;
        f32x16 s0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc)
          s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(kb + koff[kc]), qf[kc], s0, 0, 0, 0);
        float mx = fmaxf(s0[0], s0[1]);
#pragma unroll
        for (int r = 2; r < 16; ++r) mx = fmaxf(mx, s0[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx * sc2);            // finite: an unmasked tile has 32 real scores per row
        const float alpha = __builtin_amdgcn_exp2f(m - mn);   // m = -inf (first step) -> 0
        const v2f sc2v = v2f{sc2, sc2}, nmv = v2f{-mn, -mn};
        v2f rs2 = v2f{0.f, 0.f};
        u32x4 pk[2];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const v2f e0 = v2f{s0[2 * j], s0[2 * j + 1]} * sc2v + nmv;
          const v2f x0 = v2f{__builtin_amdgcn_exp2f(e0[0]), __builtin_amdgcn_exp2f(e0[1])};
          rs2 += x0;
          pk[j >> 2][j & 3] = pack2bf(x0[0], x0[1]);
        }
        float rs = rs2[0] + rs2[1];
        rs += __shfl_xor(rs, 32, 64);
        l = l * alpha + rs;
        m = mn;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
          const v2f av = v2f{alpha, alpha};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const v2f a0 = v2f{o[0][2 * j], o[0][2 * j + 1]} * av, a1 = v2f{o[1][2 * j], o[1][2 * j + 1]} * av;
            o[0][2 * j] = a0[0]; o[0][2 * j + 1] = a0[1]; o[1][2 * j] = a1[0]; o[1][2 * j + 1] = a1[1];
          }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_at(vb, voff[dt][0], voff[dt][1]), __builtin_bit_cast(bf16x8_t, pk[0]), o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_at(vb + 16 * ROWB, voff[dt][0], voff[dt][1]), __builtin_bit_cast(bf16x8_t, pk[1]), o[dt], 0, 0, 0);
        }
        kt += 32;
        continue;
      }
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc)
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Kt, kt, kc, lane), qf[kc], s, 0, 0, 0);
      // softmax in the log2 domain (one fma + v_exp_f32 per element); the key / causal masks are only evaluated on
      // tiles that contain masked elements (wave-uniform test)
      float p[16];
      float mx = -INFINITY;
      if (edge) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = c0 + kt + accrow(r, lh);
          const bool ok = key < kvalid && (!a.causal || key <= qg);
          p[r] = ok ? s[r] * sc2 : -INFINITY;
          mx = fmaxf(mx, p[r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) { p[r] = s[r] * sc2; mx = fmaxf(mx, p[r]); }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m, mx);
      const float msafe = mn == -INFINITY ? 0.f : mn;
      const float alpha = __builtin_amdgcn_exp2f(m - msafe);  // m = -inf -> 0
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(p[r] - msafe); rs += p[r]; }
      rs += __shfl_xor(rs, 32, 64);
      l = l * alpha + rs;
      m = mn;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
      const bf16x8_t pb0 = pack8(p), pb1 = pack8(p + 8);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Vt, kt, dt * 32, lane), pb0, o[dt], 0, 0, 0);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Vt, kt + 16, dt * 32, lane), pb1, o[dt], 0, 0, 0);
      }
      kt += 32;
    }
  }
  // Output.  The accumulators hold transposed tiles (lane = query row, registers = 4 consecutive columns): stored from
  // registers every instruction touches 32 rows with 8 bytes each.  With SEGCLIP_ATTN_FWD_STAGED (a.staged) the tile goes
  // through the wave's own 4 KB of the K tile instead (free once every wave has left the key loop): 8-byte LDS writes,
  // 16-byte reads of whole 128-byte rows, 16-byte coalesced global stores.
  if (a.staged) {
    __syncthreads();
    if (!wave_active) return;
    const float inv = l > 0.f ? 1.f / l : 0.f;
    char* ot = Kt + wave * 32 * ROWB;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        u32x2 t;
        t[0] = pack2bf(o[dt][rg * 4 + 0] * inv, o[dt][rg * 4 + 1] * inv);
        t[1] = pack2bf(o[dt][rg * 4 + 2] * inv, o[dt][rg * 4 + 3] * inv);
        *reinterpret_cast<u32x2*>(ot + swz(li, dt * 32 + 8 * rg + 4 * lh)) = t;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    bf16_t* Ob = a.O + (int64_t)b * a.o_sb + (int64_t)h * a.hd;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = it * 64 + lane, r = idx >> 3, c = idx & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(ot + swz(r, c * 8));
      if (q0 + r < a.Tq && c * 8 < a.hd) *reinterpret_cast<u32x4*>(Ob + (int64_t)(q0 + r) * a.o_st + c * 8) = v;
    }
    if (lh == 0 && qg < a.Tq) a.lse[((int64_t)b * a.H + h) * a.Tq + qg] = (l > 0.f) ? m * 0.6931471805599453f + __logf(l) : -INFINITY;
    return;
  }
  if (!wave_active || qg >= a.Tq) return;
  const float inv = l > 0.f ? 1.f / l : 0.f;
  bf16_t* Op = a.O + (int64_t)b * a.o_sb + (int64_t)qg * a.o_st + (int64_t)h * a.hd;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int d = dt * 32 + 8 * rg + 4 * lh;
      if (d < a.hd) {
        u32x2 t;
        t[0] = pack2bf(o[dt][rg * 4 + 0] * inv, o[dt][rg * 4 + 1] * inv);
        t[1] = pack2bf(o[dt][rg * 4 + 2] * inv, o[dt][rg * 4 + 3] * inv);
        *reinterpret_cast<u32x2*>(Op + d) = t;
      }
    }
  if (lh == 0) a.lse[((int64_t)b * a.H + h) * a.Tq + qg] = (l > 0.f) ? m * 0.6931471805599453f + __logf(l) : -INFINITY;
}

struct BwdArgs {
  const bf16_t* Q; const bf16_t* K; const bf16_t* V; const bf16_t* O; const bf16_t* dO; const float* lse;
  bf16_t* dQ; bf16_t* dK; bf16_t* dV;
  int H, Tq, Tk, hd;
  int64_t q_sb, q_st, k_sb, k_st, v_sb, v_st, o_sb, o_st, do_sb, do_st;
  int64_t dq_sb, dq_st, dk_sb, dk_st, dv_sb, dv_st;
  float scale; int causal;
  const int* klen;     // nullable: valid keys per sample (fused kernel only)
  float* colsum_part;  // nullable: [B][3][H*hd] token sums of dQ | dK | dV (the in_proj bias gradient, per sample)
  int abl;             // experiments (SEGCLIP_ATTN_ABL): 0 = the kernel; see attention_sp.inc
  int nitems;          // single-pass kernel: B * H items, walked by a persistent grid
  float* ws;           // attention_dqw.inc, sequences of more than 224 tokens: fp32 dQ accumulators across key chunks
};

constexpr int TMAX = 256;
// LDS bytes of the backward kernel for tiles of tp (padded) rows: two tiles (Q,dO for the dK/dV pass, then K,V for
// the dQ pass in the same space) + lse, D, cs vectors + the cross-wave reduce pad
static inline size_t bwd_lds_bytes(int tp) { return (size_t)tp * (2 * ROWB + 3 * 4) + 8 * 3 * 64 * 4; }

#include "attention_smallq.inc"

// MFMA operand fragment straight from global memory (rows that only ONE wave needs): lane l -> row row0 + (l&31),
// cols kc*16 + 8*(l>>5) .. +7; zero beyond nvalid / hd.  Unconditional load from a clamped address + select.
__device__ __forceinline__ bf16x8_t frag_rows_g(const bf16_t* __restrict__ X, int64_t st, int row0, int nvalid, int kc,
                                                int lane, int hd) {
  const int r = row0 + (lane & 31), c = kc * 16 + 8 * (lane >> 5);
  const int rc = r < nvalid ? r : nvalid - 1, cc = c < hd ? c : 0;
  u32x4 v = *reinterpret_cast<const u32x4*>(X + (int64_t)rc * st + cc);
  if (!(r < nvalid && c < hd)) v = u32x4{0u, 0u, 0u, 0u};
  return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ void store_acc_T(bf16_t* base, int64_t st, int row, int nrows, int hd, const f32x16 (&acc)[2],
                                            int lh) {
  if (row >= nrows) return;
  bf16_t* p = base + (int64_t)row * st;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int d = dt * 32 + 8 * rg + 4 * lh;
      if (d < hd) {
        u32x2 t;
        t[0] = pack2bf(acc[dt][rg * 4 + 0], acc[dt][rg * 4 + 1]);
        t[1] = pack2bf(acc[dt][rg * 4 + 2], acc[dt][rg * 4 + 3]);
        *reinterpret_cast<u32x2*>(p + d) = t;
      }
    }
}

__global__ __launch_bounds__(512) void attn_bwd_bf16_kernel(BwdArgs a) {
  // Dynamic LDS sized by the padded sequence length, and only TWO tiles: Q and dO (read by every wave in the dK/dV
  // pass), later overwritten by K and V (read by every wave in the dQ pass).  The rows only one wave needs (its own
  // key tile of K,V / query tile of Q,dO) come straight from global memory.  197 tokens -> 66 KB, 77 -> 31 KB, so
  // two (vision) or more (text) workgroups share a CU and one's load/store phases overlap the other's MFMAs.
  extern __shared__ __attribute__((aligned(16))) char smem_bwd[];
  const int tp0 = (((a.Tq > a.Tk ? a.Tq : a.Tk) + 31) & ~31);
  char* T0 = smem_bwd;               // Q, then K
  char* T1 = smem_bwd + tp0 * ROWB;  // dO, then V
  char* Qt = T0;
  char* Gt = T1;
  char* Kt = T0;
  char* Vt = T1;
  float* Ls = reinterpret_cast<float*>(smem_bwd + 2 * tp0 * ROWB);
  float* Ds = Ls + tp0;
  float* Cs = Ds + tp0;   // cs[key] = sum_q dS[q][key]
  float* Red = Cs + tp0;  // [8 waves][3][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t hoff = (int64_t)h * a.hd;
  const bf16_t* Qp = a.Q + b * a.q_sb + hoff;
  const bf16_t* Kp = a.K + b * a.k_sb + hoff;
  const bf16_t* Vp = a.V + b * a.v_sb + hoff;
  const bf16_t* Op = a.O + b * a.o_sb + hoff;
  const bf16_t* Gp = a.dO + b * a.do_sb + hoff;
  const int tqp = (a.Tq + 31) & ~31, tkp = (a.Tk + 31) & ~31;
  const int kvalid = a.klen ? (a.klen[b] < a.Tk ? a.klen[b] : a.Tk) : a.Tk;

  stage_rows(Qt, Qp, a.q_st, 0, a.Tq, tqp, a.hd, tid, blockDim.x);
  // dO tile + D[q] = sum_d dO*O (8 consecutive lanes share a row); branch-free batched loads as in stage_rows
  {
    const int total = tqp * 8;
    for (int base = 0; base < total; base += 2 * (int)blockDim.x) {
      u32x4 gv[2], ov[2];
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = base + it * (int)blockDim.x + tid;
        const int r = idx >> 3, c = idx & 7;
        const int rc = r < a.Tq ? r : a.Tq - 1;
        const int cc = c * 8 < a.hd ? c : 0;
        gv[it] = *reinterpret_cast<const u32x4*>(Gp + (int64_t)rc * a.do_st + cc * 8);
        ov[it] = *reinterpret_cast<const u32x4*>(Op + (int64_t)rc * a.o_st + cc * 8);
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = base + it * (int)blockDim.x + tid;
        const int r = idx >> 3, c = idx & 7;
        const bool ok = r < a.Tq && c * 8 < a.hd;
        const u32x4 g = ok ? gv[it] : u32x4{0u, 0u, 0u, 0u};
        const u32x4 o4 = ok ? ov[it] : u32x4{0u, 0u, 0u, 0u};
        float sdot = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sdot += __uint_as_float(g[j] << 16) * __uint_as_float(o4[j] << 16);
          sdot += __uint_as_float(g[j] & 0xffff0000u) * __uint_as_float(o4[j] & 0xffff0000u);
        }
        sdot += __shfl_xor(sdot, 1, 64); sdot += __shfl_xor(sdot, 2, 64); sdot += __shfl_xor(sdot, 4, 64);
        if (idx < total) {
          *reinterpret_cast<u32x4*>(Gt + swz(r, c * 8)) = g;
          if (c == 0) {
            Ds[r] = sdot;
            Ls[r] = r < a.Tq ? a.lse[((int64_t)b * a.H + h) * a.Tq + r] * 1.4426950408889634f : 0.f;  // log2 domain
          }
        }
      }
    }
  }
  __syncthreads();
  const float sc2 = a.scale * 1.4426950408889634f;

  // ---------------- pass A: this wave owns key tiles; dK, dV ----------------
  for (int k0 = wave * 32; k0 < tkp; k0 += nw * 32) {
    bf16x8_t kf[4], vf[4];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      kf[kc] = frag_rows_g(Kp, a.k_st, k0, a.Tk, kc, lane, a.hd);
      vf[kc] = frag_rows_g(Vp, a.v_st, k0, a.Tk, kc, lane, a.hd);
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
    const int key = k0 + li;
    float csl = 0.f;
    for (int q0 = 0; q0 < tqp; q0 += 32) {
      if (a.causal && q0 + 31 < k0) continue;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Qt, q0, kc, lane), kf[kc], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Gt, q0, kc, lane), vf[kc], dp, 0, 0, 0);
      }
      // P = exp2(S * scale*log2e - lse*log2e): one fma + v_exp_f32; dS is kept WITHOUT the softmax scale (applied once
      // to dK, the column sums and dQ at the end); masks only on tiles that contain masked elements (wave-uniform)
      const bool edge = k0 + 32 > kvalid || q0 + 32 > a.Tq || (a.causal && k0 + 31 > q0);
      float p[16], ds[16];
      if (edge) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int qb = q0 + 8 * rg + 4 * lh;  // 4 consecutive query rows per register group
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + qb);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(Ds + qb);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = rg * 4 + j, q = qb + j;
            const bool ok = key < kvalid && q < a.Tq && (!a.causal || key <= q);
            const float pv = ok ? __builtin_amdgcn_exp2f(s[r] * sc2 - l4[j]) : 0.f;
            p[r] = pv;
            ds[r] = pv * (dp[r] - d4[j]);
            csl += ds[r];
          }
        }
      } else {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int qb = q0 + 8 * rg + 4 * lh;
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + qb);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(Ds + qb);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = rg * 4 + j;
            const float pv = __builtin_amdgcn_exp2f(s[r] * sc2 - l4[j]);
            p[r] = pv;
            ds[r] = pv * (dp[r] - d4[j]);
            csl += ds[r];
          }
        }
      }
      const bf16x8_t pb0 = pack8(p), pb1 = pack8(p + 8), sb0 = pack8(ds), sb1 = pack8(ds + 8);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Gt, q0, dt * 32, lane), pb0, dv[dt], 0, 0, 0);
        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Gt, q0 + 16, dt * 32, lane), pb1, dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Qt, q0, dt * 32, lane), sb0, dk[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Qt, q0 + 16, dt * 32, lane), sb1, dk[dt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] *= a.scale; dk[1][r] *= a.scale; }
    store_acc_T(a.dK + b * a.dk_sb + hoff, a.dk_st, key, a.Tk, a.hd, dk, lh);
    store_acc_T(a.dV + b * a.dv_sb + hoff, a.dv_st, key, a.Tk, a.hd, dv, lh);
    csl += __shfl_xor(csl, 32, 64);
    if (lh == 0) Cs[key] = csl * a.scale;
  }

  // token sums of dV (= token sums of dO: the rows of P sum to one) while dO is still in LDS
  float accv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) accv[j] = 0.f;
  if (a.colsum_part) {
    const int c = tid & 7, nrl = (int)blockDim.x >> 3;
    for (int r = tid >> 3; r < tqp; r += nrl) {
      const u32x4 gv = *reinterpret_cast<const u32x4*>(Gt + swz(r, c * 8));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        accv[2 * j] += __uint_as_float(gv[j] << 16);
        accv[2 * j + 1] += __uint_as_float(gv[j] & 0xffff0000u);
      }
    }
  }
  __syncthreads();  // every wave is done with Q and dO: the two tiles become K and V
  stage_rows(Kt, Kp, a.k_st, 0, a.Tk, tkp, a.hd, tid, blockDim.x);
  stage_rows(Vt, Vp, a.v_st, 0, a.Tk, tkp, a.hd, tid, blockDim.x);
  __syncthreads();

  // ---------------- pass B: this wave owns query tiles; dQ ----------------
  for (int q0 = wave * 32; q0 < tqp; q0 += nw * 32) {
    bf16x8_t qf[4], gf[4];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      qf[kc] = frag_rows_g(Qp, a.q_st, q0, a.Tq, kc, lane, a.hd);
      gf[kc] = frag_rows_g(Gp, a.do_st, q0, a.Tq, kc, lane, a.hd);
    }
    const int q = q0 + li;
    const float lq = Ls[q], dq_ = Ds[q];
    f32x16 dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
    for (int k0 = 0; k0 < tkp; k0 += 32) {
      if (a.causal && k0 > q0 + 31) break;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Kt, k0, kc, lane), qf[kc], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Vt, k0, kc, lane), gf[kc], dp, 0, 0, 0);
      }
      const bool edge = k0 + 32 > kvalid || q0 + 32 > a.Tq || (a.causal && k0 + 31 > q0);
      float ds[16];
      if (edge) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + accrow(r, lh);
          const bool ok = key < kvalid && q < a.Tq && (!a.causal || key <= q);
          ds[r] = ok ? __builtin_amdgcn_exp2f(s[r] * sc2 - lq) * (dp[r] - dq_) : 0.f;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) ds[r] = __builtin_amdgcn_exp2f(s[r] * sc2 - lq) * (dp[r] - dq_);
      }
      const bf16x8_t sb0 = pack8(ds), sb1 = pack8(ds + 8);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Kt, k0, dt * 32, lane), sb0, dq[dt], 0, 0, 0);
        dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(Kt, k0 + 16, dt * 32, lane), sb1, dq[dt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] *= a.scale; dq[1][r] *= a.scale; }
    store_acc_T(a.dQ + b * a.dq_sb + hoff, a.dq_st, q, a.Tq, a.hd, dq, lh);
  }

  // ---------------- token sums of dQ, dK, dV (in_proj bias gradient of this sample and head) ----------------
  //   sum_q dQ[q][d]   = sum_key K[key][d] cs[key]   (K is in LDS now, cs from the dK/dV pass)
  //   sum_key dV[key][d] = sum_q dO[q][d]            (accumulated above)
  //   sum_key dK[key][d] = sum_q Q[q][d] rowsum_key(dS[q][:]) = 0 exactly: the rows of dS sum to zero (softmax is
  //                        shift invariant; what the reference accumulates there is rounding noise ~1e-8)
  if (a.colsum_part) {
    const int c = tid & 7, nrl = (int)blockDim.x >> 3;
    float accq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) accq[j] = 0.f;
    for (int r = tid >> 3; r < tkp; r += nrl) {
      const u32x4 kv = *reinterpret_cast<const u32x4*>(Kt + swz(r, c * 8));
      const float w = Cs[r];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        accq[2 * j] += w * __uint_as_float(kv[j] << 16);
        accq[2 * j + 1] += w * __uint_as_float(kv[j] & 0xffff0000u);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float vq = accq[j], vv = accv[j];
      vq += __shfl_xor(vq, 8, 64); vv += __shfl_xor(vv, 8, 64);
      vq += __shfl_xor(vq, 16, 64); vv += __shfl_xor(vv, 16, 64);
      vq += __shfl_xor(vq, 32, 64); vv += __shfl_xor(vv, 32, 64);
      if (lane < 8) {
        Red[(wave * 3 + 0) * 64 + c * 8 + j] = vq;
        Red[(wave * 3 + 2) * 64 + c * 8 + j] = vv;
      }
    }
    __syncthreads();
    for (int t = tid; t < 192; t += (int)blockDim.x) {
      const int m = t >> 6, dd = t & 63;
      float v = 0.f;
      if (m != 1)
        for (int w = 0; w < nw; ++w) v += Red[(w * 3 + m) * 64 + dd];
      if (dd < a.hd) a.colsum_part[((int64_t)b * 3 + m) * ((int64_t)a.H * a.hd) + hoff + dd] = v;
    }
  }
}

#include "attention_sp.inc"
#include "attention_pf.inc"
#include "attention_spl.inc"
#include "attention_dqw.inc"
#include "attention_stream.inc"
// e4m3 forward (BASELINE configs[4] as first read): forward-only, non-scaled e4m3 MFMA = the bf16 rate, measured SLOWER than the
// bf16 kernel in every round (1350 vs 1395 pairs/s at B = 128, profiles/r04_bench_configs.json).  Round 5: out of the default
// build - configs[4] runs bf16 attention; the kernel stays reachable in `build.sh -DSEGCLIP_EXPERIMENTS` libraries.
#ifdef SEGCLIP_EXPERIMENTS
#include "attention_fp8.inc"
#endif

// ------------------------------- f32 path helpers -------------------------------------------
// in-place row softmax of S (rows = B*H*Tq, Tk cols), causal mask by query index row % Tq
__global__ void softmax_rows_kernel(float* __restrict__ S, int64_t rows, int Tq, int Tk, int causal, const int* klen,
                                    int H) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int q = (int)(row % Tq);
  float* s = S + row * Tk;
  int lim = causal ? (q + 1 < Tk ? q + 1 : Tk) : Tk;
  if (klen) { const int kv = klen[row / ((int64_t)H * Tq)]; lim = kv < lim ? kv : lim; }
  float mx = -INFINITY;
  for (int c = lane; c < lim; c += 64) mx = fmaxf(mx, s[c]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < lim; c += 64) sum += expf(s[c] - mx);
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int c = lane; c < Tk; c += 64) s[c] = c < lim ? expf(s[c] - mx) * inv : 0.f;
}
// dS = P * (dP - sum_k dP*P) * scale, in place on dP
__global__ void softmax_bwd_rows_kernel(const float* __restrict__ P, float* __restrict__ dP, int64_t rows, int Tk,
                                        float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = P + row * Tk;
  float* d = dP + row * Tk;
  float dot = 0.f;
  for (int c = lane; c < Tk; c += 64) dot += p[c] * d[c];
  dot = wave_sum(dot);
  for (int c = lane; c < Tk; c += 64) d[c] = p[c] * (d[c] - dot) * scale;
}

void base_gemm(segclip_gemm_desc& g, const segclip_attn_desc* d) {
  g = segclip_gemm_desc{};
  g.nb1 = d->B; g.nb2 = d->H;
  g.a_dtype = g.b_dtype = g.c_dtype = g.r_dtype = SEGCLIP_F32;
  g.alpha = 1.f;
}

bool bf16_ok(const segclip_attn_desc* d) {
  const int64_t s[] = {d->q_sb, d->q_st, d->k_sb, d->k_st, d->v_sb, d->v_st, d->o_sb, d->o_st};
  for (int64_t v : s) if (v % 8) return false;
  return d->hd % 8 == 0 && d->hd <= 64;
}


// persistent forward (attention_pf.inc): launch instance <NT, CAUSAL> on a grid of as many workgroups as the device holds
template <int NT, bool CAUSAL, int ABL = 0>
int launch_fwd_pf(const FwdArgs& a, int nitems, hipStream_t stream) {
  int dev = 0;
  SEGCLIP_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, "attn_fwd: cannot query the current device");
  static bool attr_set[64] = {};
  static int ncu_dev[64] = {};
  static int per_cu_cache[64][33] = {};                    // by (device, key tile rows / 8)
  const size_t lds = fwd_pf_lds_bytes(NT, a.Tq);
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_pf_kernel<NT, CAUSAL, ABL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    SEGCLIP_REQUIRE(e == hipSuccess, "attn_fwd bf16: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    ncu_dev[dev] = ncu;
    attr_set[dev] = true;
  }
  int& per_cu = per_cu_cache[dev][pf_kv_rows(a.Tq) >> 3];
  if (per_cu == 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_fwd_pf_kernel<NT, CAUSAL, ABL>, (NT + 1) * 64, lds) != hipSuccess || n < 1) n = 1;
    static const int grid_env = [] { const char* e = segclip_tuning_env("SEGCLIP_ATTN_FWD_GRID"); return e ? atoi(e) : 0; }();
    per_cu = grid_env > 0 ? grid_env : n;
  }
  const int cap = ncu_dev[dev] * per_cu;
  const int grid = nitems < cap ? nitems : cap;
  hipLaunchKernelGGL((attn_fwd_pf_kernel<NT, CAUSAL, ABL>), dim3((unsigned)grid), dim3((NT + 1) * 64), lds, stream, a, nitems);
  SEGCLIP_CHECK_LAUNCH("attn_fwd_pf");
  return 0;
}

// streaming backward with a dQ wave (attention_dqw.inc): one workgroup of NT + 1 waves per CU walks its share of the items
template <int NT, bool MULTI>
int launch_bwd_dqw(const BwdArgs& a, int ncu, int dev, hipStream_t stream) {
  static bool attr_set[64] = {};
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dqw_bf16_kernel<NT, MULTI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    SEGCLIP_REQUIRE(e == hipSuccess, "attn_bwd bf16: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    attr_set[dev] = true;
  }
  const int grid = a.nitems < ncu ? a.nitems : ncu;
  hipLaunchKernelGGL((attn_bwd_dqw_bf16_kernel<NT, MULTI>), dim3((unsigned)grid), dim3((NT + 1) * 64), bwd_dqw_lds_bytes<NT>(), stream, a);
  SEGCLIP_CHECK_LAUNCH("attn_bwd_dqw_bf16");
  return 0;
}

// workgroups of the key-chunked dqw launch: the CUs of the current device
int dqw_multi_max_grid() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  return n;
}

}  // namespace

extern "C" size_t segclip_attn_stats_bytes(const segclip_attn_desc* d) {
  if (d->dtype == SEGCLIP_BF16) return (size_t)d->B * d->H * d->Tq * sizeof(float);
  return (size_t)d->B * d->H * d->Tq * d->Tk * sizeof(float);
}
extern "C" size_t segclip_attn_bwd_ws_bytes(const segclip_attn_desc* d) {
  if (d->dtype == SEGCLIP_BF16) {
    if (d->Tq <= TMAX && d->Tk <= TMAX) return 0;
    const size_t stream_ws = (size_t)d->B * d->H * (d->Tq + d->Tk) * sizeof(float);  // streaming kernels: cs[key] and D[q]
    // attention_dqw.inc, key-chunked: 8 KB of fp32 dQ accumulators per query tile and workgroup (one workgroup per CU)
    const size_t dqw_ws = (size_t)dqw_multi_max_grid() * (size_t)(d->Tq / 32 + 1) * 8192;
    return stream_ws > dqw_ws ? stream_ws : dqw_ws;
  }
  return (size_t)d->B * d->H * d->Tq * d->Tk * sizeof(float);
}

extern "C" int segclip_attn_fwd(const segclip_attn_desc* d, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGCLIP_REQUIRE(d->hd <= 64 && d->hd > 0, "attn: head_dim %lld unsupported (<=64)", (long long)d->hd);
  SEGCLIP_REQUIRE(d->stats != nullptr, "attn_fwd: stats buffer required");
  if (d->B == 0 || d->Tq == 0) return 0;
  if (d->dtype == SEGCLIP_BF16) {
    SEGCLIP_REQUIRE(bf16_ok(d), "attn_fwd bf16: head_dim and all strides must be multiples of 8");
    FwdArgs a;
    a.Q = (const bf16_t*)d->Q; a.K = (const bf16_t*)d->K; a.V = (const bf16_t*)d->V; a.O = (bf16_t*)d->O;
    a.lse = (float*)d->stats;
    a.H = (int)d->H; a.Tq = (int)d->Tq; a.Tk = (int)d->Tk; a.hd = (int)d->hd;
    a.q_sb = d->q_sb; a.q_st = d->q_st; a.k_sb = d->k_sb; a.k_st = d->k_st; a.v_sb = d->v_sb; a.v_st = d->v_st;
    a.o_sb = d->o_sb; a.o_st = d->o_st; a.scale = d->scale; a.causal = d->causal;
    a.klen = (const int*)d->klen;
    // LDS-staged output rows: 117 -> 112 us at T = 196 (B = 256, H = 12), 27.8 -> 29.3 us at T = 77: on for the long sequences
    static const int fwd_staged = [] { const char* e = segclip_tuning_env("SEGCLIP_ATTN_FWD_STAGED"); return e ? atoi(e) : -1; }();
    a.staged = fwd_staged >= 0 ? fwd_staged : (d->Tq > 128 ? 1 : 0);
    static const int fwd_lean = [] { const char* e = segclip_tuning_env("SEGCLIP_ATTN_FWD_LEAN"); return e ? atoi(e) : 1; }();
    a.lean = fwd_lean;
    SEGCLIP_REQUIRE(!(d->klen && (d->flags & SEGCLIP_ATTN_FP8)), "attn_fwd: klen is not supported by the fp8 kernel");
    if (smallq::covers(d)) {   // at most 8 queries (the learnable-center cross-attention): one wave per (batch, head), VALU
      const int nitems = (int)(d->B * d->H);
      hipLaunchKernelGGL(smallq::attn_smallq_fwd_kernel, dim3((unsigned)cdiv(nitems, smallq::WPB_FWD)), dim3(smallq::WPB_FWD * 64),
                         smallq::lds_bytes((int)d->Tk, false), stream, a, nitems);
      SEGCLIP_CHECK_LAUNCH("attn_smallq_fwd");
      return 0;
    }
    const int tiles = (int)cdiv(d->Tq, 32);
    // self-attention of 65..96 / 161..200 tokens: the persistent kernel with a loader wave (attention_pf.inc);
    // SEGCLIP_ATTN_FWD_PF=0 falls back to one workgroup per (batch, head)
    static const int use_pf = [] { const char* e = segclip_tuning_env("SEGCLIP_ATTN_FWD_PF"); return e ? atoi(e) : 1; }();
    if (use_pf && d->Tq == d->Tk && !(d->flags & SEGCLIP_ATTN_FP8) && (tiles == 3 || tiles == 6 || tiles == 7) &&
        fwd_pf_lds_bytes(tiles, (int)d->Tq) <= 160 * 1024) {
      const int nitems = (int)(d->B * d->H);
      if (d->causal) {
        if (tiles == 3) return launch_fwd_pf<3, true>(a, nitems, stream);
        if (tiles == 6) return launch_fwd_pf<6, true>(a, nitems, stream);
        return launch_fwd_pf<7, true>(a, nitems, stream);
      }
#ifdef SEGCLIP_EXPERIMENTS
      static const int pf_abl = segclip_ablation_env("SEGCLIP_ATTN_PF_ABL");
      if (tiles == 7 && pf_abl == 4) return launch_fwd_pf<7, false, 4>(a, nitems, stream);
      if (tiles == 7 && pf_abl == 5) return launch_fwd_pf<7, false, 5>(a, nitems, stream);
#endif
      if (tiles == 3) return launch_fwd_pf<3, false>(a, nitems, stream);
      if (tiles == 6) return launch_fwd_pf<6, false>(a, nitems, stream);
      return launch_fwd_pf<7, false>(a, nitems, stream);
    }
    const int nw = tiles < 8 ? tiles : (tiles <= 8 ? 8 : (int)cdiv(tiles, cdiv(tiles, 8)));
    SEGCLIP_REQUIRE(d->B * d->H <= 65535, "attn_fwd: B*H too large");
#ifndef SEGCLIP_EXPERIMENTS
    if (d->flags & SEGCLIP_ATTN_FP8) {
      segclip_set_error("attn_fwd: the e4m3 forward is not part of the default build (slower than bf16 at head_dim 64; "
                        "build.sh -DSEGCLIP_EXPERIMENTS keeps it): configs[4] runs bf16 attention");
      return SEGCLIP_ERR_UNSUPPORTED;
    }
#else
    if (d->flags & SEGCLIP_ATTN_FP8) {
      hipLaunchKernelGGL(attn_fwd_fp8_kernel, dim3((unsigned)cdiv(tiles, nw), (unsigned)(d->B * d->H)), dim3(nw * 64), 0,
                         stream, a);
      SEGCLIP_CHECK_LAUNCH("attn_fwd_fp8");
      return 0;
    }
#endif
    hipLaunchKernelGGL(attn_fwd_bf16_kernel, dim3((unsigned)cdiv(tiles, nw), (unsigned)(d->B * d->H)), dim3(nw * 64), 0,
                       stream, a);
    SEGCLIP_CHECK_LAUNCH("attn_fwd_bf16");
    return 0;
  }
  // f32: S -> stats, softmax in place, O = P V
  float* P = (float*)d->stats;
  segclip_gemm_desc g;
  base_gemm(g, d);
  g.A = d->Q; g.B = d->K; g.C = P; g.M = d->Tq; g.N = d->Tk; g.K = d->hd;
  g.sam = d->q_st; g.sak = 1; g.sbn = d->k_st; g.sbk = 1; g.ldc = d->Tk;
  g.bsA1 = d->q_sb; g.bsA2 = d->hd; g.bsB1 = d->k_sb; g.bsB2 = d->hd;
  g.bsC1 = d->H * d->Tq * d->Tk; g.bsC2 = d->Tq * d->Tk; g.alpha = d->scale;
  int rc = segclip_gemm_f32_launch(&g, stream);
  if (rc) return rc;
  const int64_t rows = d->B * d->H * d->Tq;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, stream, P, rows, (int)d->Tq,
                     (int)d->Tk, d->causal, (const int*)d->klen, (int)d->H);
  SEGCLIP_CHECK_LAUNCH("attn_softmax_rows");
  base_gemm(g, d);
  g.A = P; g.B = d->V; g.C = d->O; g.M = d->Tq; g.N = d->hd; g.K = d->Tk;
  g.sam = d->Tk; g.sak = 1; g.sbn = 1; g.sbk = d->v_st; g.ldc = d->o_st;
  g.bsA1 = d->H * d->Tq * d->Tk; g.bsA2 = d->Tq * d->Tk; g.bsB1 = d->v_sb; g.bsB2 = d->hd;
  g.bsC1 = d->o_sb; g.bsC2 = d->hd;
  return segclip_gemm_f32_launch(&g, stream);
}

extern "C" int segclip_attn_bwd(const segclip_attn_desc* d, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SEGCLIP_REQUIRE(d->hd <= 64 && d->hd > 0, "attn: head_dim %lld unsupported (<=64)", (long long)d->hd);
  if (d->B == 0 || d->Tq == 0) return 0;
  if (d->dtype == SEGCLIP_BF16) {
    SEGCLIP_REQUIRE(bf16_ok(d), "attn_bwd bf16: head_dim and all strides must be multiples of 8");
    const int64_t s[] = {d->dq_sb, d->dq_st, d->dk_sb, d->dk_st, d->dv_sb, d->dv_st, d->do_sb, d->do_st};
    for (int64_t v : s) SEGCLIP_REQUIRE(v % 8 == 0, "attn_bwd bf16: gradient strides must be multiples of 8");
    BwdArgs a;
    a.Q = (const bf16_t*)d->Q; a.K = (const bf16_t*)d->K; a.V = (const bf16_t*)d->V; a.O = (const bf16_t*)d->O;
    a.dO = (const bf16_t*)d->dO; a.lse = (const float*)d->stats;
    a.dQ = (bf16_t*)d->dQ; a.dK = (bf16_t*)d->dK; a.dV = (bf16_t*)d->dV;
    a.H = (int)d->H; a.Tq = (int)d->Tq; a.Tk = (int)d->Tk; a.hd = (int)d->hd;
    a.q_sb = d->q_sb; a.q_st = d->q_st; a.k_sb = d->k_sb; a.k_st = d->k_st; a.v_sb = d->v_sb; a.v_st = d->v_st;
    a.o_sb = d->o_sb; a.o_st = d->o_st; a.do_sb = d->do_sb; a.do_st = d->do_st;
    a.dq_sb = d->dq_sb; a.dq_st = d->dq_st; a.dk_sb = d->dk_sb; a.dk_st = d->dk_st; a.dv_sb = d->dv_sb; a.dv_st = d->dv_st;
    a.scale = d->scale; a.causal = d->causal;
    a.colsum_part = (float*)d->colsum_part;
    a.klen = (const int*)d->klen;
    static const int abl_env = segclip_ablation_env("SEGCLIP_ATTN_ABL");
    a.abl = abl_env;
    a.nitems = (int)(d->B * d->H);
    a.ws = (float*)d->ws;
    if (smallq::covers(d)) {
      hipLaunchKernelGGL(smallq::attn_smallq_bwd_kernel, dim3((unsigned)cdiv(a.nitems, smallq::WPB_BWD)), dim3(smallq::WPB_BWD * 64),
                         smallq::lds_bytes((int)d->Tk, true), stream, a, a.nitems);
      SEGCLIP_CHECK_LAUNCH("attn_smallq_bwd");
      return 0;
    }
    if (d->Tq > TMAX || d->Tk > TMAX) {
      SEGCLIP_REQUIRE(d->klen == nullptr, "attn_bwd bf16: klen needs sequences of at most %d tokens", TMAX);
      // long sequences: two streaming launches (dK,dV | dQ), 8 owned tiles per workgroup
      SEGCLIP_REQUIRE(d->ws != nullptr, "attn_bwd bf16: workspace required for sequences longer than %d", TMAX);
      // unmasked self-attention, head_dim 64: the dQ-wave kernel over chunks of 224 keys (attention_dqw.inc, MULTI).
      // SEGCLIP_ATTN_BWD_DQW_LONG=0 (tuning) keeps the two streaming launches
      static const int use_dqw_long = [] { const char* e = segclip_tuning_env("SEGCLIP_ATTN_BWD_DQW_LONG"); return e ? atoi(e) : 1; }();
      // (its token sums ride on a padded key in the last chunk and on two padded rows of the last query tile)
      if (use_dqw_long && d->Tq == d->Tk && !d->causal && d->hd == 64 && d->Tq % 32 <= 30 && d->Tq % 224 != 0) {
        int dev = 0;
        SEGCLIP_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, "attn_bwd: cannot query the current device");
        return launch_bwd_dqw<7, true>(a, dqw_multi_max_grid(), dev, stream);
      }
      const size_t lds = bwd_stream_lds_bytes();
      static bool attr_set = false;
      if (!attr_set) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_stream_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_stream_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        SEGCLIP_REQUIRE(e1 == hipSuccess && e2 == hipSuccess, "attn_bwd bf16: cannot raise the dynamic LDS limit");
        attr_set = true;
      }
      const int ktiles = (int)cdiv(d->Tk, 32), qtiles = (int)cdiv(d->Tq, 32);
      const int nwk = ktiles < 8 ? ktiles : 8, nwq = qtiles < 8 ? qtiles : 8;
      hipLaunchKernelGGL(attn_bwd_dkv_stream_kernel, dim3((unsigned)(d->B * d->H), (unsigned)cdiv(ktiles, nwk)),
                         dim3(nwk * 64), lds, stream, a, (float*)d->ws);
      SEGCLIP_CHECK_LAUNCH("attn_bwd_dkv_stream");
      hipLaunchKernelGGL(attn_bwd_dq_stream_kernel, dim3((unsigned)(d->B * d->H), (unsigned)cdiv(qtiles, nwq)),
                         dim3(nwq * 64), lds, stream, a, (const float*)d->ws);
      SEGCLIP_CHECK_LAUNCH("attn_bwd_dq_stream");
      return 0;
    }
    const int tiles = (int)cdiv(d->Tq > d->Tk ? d->Tq : d->Tk, 32);
    // self-attention: the single-pass kernel (attention_sp.inc), one wave per key tile.  SEGCLIP_ATTN_BWD_SP=0 falls back
    // to the two-pass kernel (benchmarking); cross-attention (Tq != Tk) always takes the two-pass kernel.
    static const int use_sp = [] { const char* e = segclip_tuning_env("SEGCLIP_ATTN_BWD_SP"); return e ? atoi(e) : 1; }();
    if (use_sp && d->Tq == d->Tk) {
      const size_t lds_sp = bwd_sp_lds_bytes(tiles * 32);
      // per DEVICE (function attributes and the CU count belong to the device the launch goes to, not to the process)
      int dev = 0;
      SEGCLIP_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, "attn_bwd: cannot query the current device");
      static bool sp_attr_set[64] = {};
      static int ncu_dev[64] = {};
      if (!sp_attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_sp_bf16_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_sp_lds_bytes(TMAX));
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_sp_bf16_kernel<true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_sp_lds_bytes(TMAX));
        SEGCLIP_REQUIRE(e == hipSuccess && e2 == hipSuccess, "attn_bwd bf16: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
        sp_attr_set[dev] = true;
      }
      // persistent grid: as many workgroups as the chip holds at once (LDS / registers: 1 per CU at T = 197, more for the
      // short text sequences); SEGCLIP_ATTN_BWD_GRID overrides the number per CU (0 = one workgroup per item)
      const bool masked = d->causal || d->klen;
      if (ncu_dev[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        ncu_dev[dev] = n;
      }
      const int ncu = ncu_dev[dev];
      static const int grid_env = [] { const char* e = segclip_tuning_env("SEGCLIP_ATTN_BWD_GRID"); return e ? atoi(e) : -1; }();
      int per_cu = 0;
      hipError_t eo = masked ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, attn_bwd_sp_bf16_kernel<true>, tiles * 64, lds_sp)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, attn_bwd_sp_bf16_kernel<false>, tiles * 64, lds_sp);
      if (eo != hipSuccess || per_cu < 1) per_cu = 1;
      if (grid_env > 0) per_cu = grid_env;
      int64_t grid = grid_env == 0 ? a.nitems : (int64_t)ncu * per_cu;
      if (grid > a.nitems) grid = a.nitems;
      // vision tower (no mask, head_dim 64, 7 tiles, padded rows in the last tile): the query tiles as a stream with a dQ wave
      // (attention_dqw.inc, round 6); SEGCLIP_ATTN_BWD_DQW=0 keeps the kernels below
      static const int use_dqw = [] { const char* e = segclip_tuning_env("SEGCLIP_ATTN_BWD_DQW"); return e ? atoi(e) : 1; }();
      if (use_dqw && !masked && d->hd == 64 && tiles == 7 && d->Tq % 32 >= 1 && d->Tq % 32 <= 30)
        return launch_bwd_dqw<7, false>(a, ncu, dev, stream);
      // vision tower (no mask, 5-7 tiles): the variant whose memory traffic is issued by a loader wave (attention_spl.inc);
      // SEGCLIP_ATTN_BWD_SPL=0 keeps attention_sp.inc
      static const int use_spl = [] { const char* e = segclip_tuning_env("SEGCLIP_ATTN_BWD_SPL"); return e ? atoi(e) : 1; }();
      if (use_spl && !masked && tiles >= 5 && tiles <= 7 && bwd_spl_lds_bytes((int)d->Tq) <= 160 * 1024) {
        static bool spl_attr_set[64] = {};
        if (!spl_attr_set[dev]) {
          hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_spl_bf16_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
          SEGCLIP_REQUIRE(e3 == hipSuccess, "attn_bwd bf16: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e3));
          spl_attr_set[dev] = true;
        }
        int64_t g2 = grid_env == 0 ? a.nitems : (int64_t)ncu * (grid_env > 0 ? grid_env : 1);
        if (g2 > a.nitems) g2 = a.nitems;
        hipLaunchKernelGGL(attn_bwd_spl_bf16_kernel, dim3((unsigned)g2), dim3((tiles + 1) * 64), bwd_spl_lds_bytes((int)d->Tq), stream, a);
        SEGCLIP_CHECK_LAUNCH("attn_bwd_spl_bf16");
        return 0;
      }
      if (masked)
        hipLaunchKernelGGL(attn_bwd_sp_bf16_kernel<true>, dim3((unsigned)grid), dim3(tiles * 64), lds_sp, stream, a);
      else
        hipLaunchKernelGGL(attn_bwd_sp_bf16_kernel<false>, dim3((unsigned)grid), dim3(tiles * 64), lds_sp, stream, a);
      SEGCLIP_CHECK_LAUNCH("attn_bwd_sp_bf16");
      return 0;
    }
    // 4 waves per workgroup (each wave walks over 1-2 tiles): two such workgroups fit the registers (2 waves per SIMD
    // at ~215 VGPRs) and the LDS of a CU, so their load / MFMA / store phases interleave.  SEGCLIP_ATTN_BWD_WAVES
    // overrides (benchmarking).
    static const int force_waves = [] { const char* e = segclip_tuning_env("SEGCLIP_ATTN_BWD_WAVES"); return e ? atoi(e) : 0; }();
    int nw = tiles < 4 ? tiles : 4;
    if (force_waves >= 1 && force_waves <= 8) nw = tiles < force_waves ? tiles : force_waves;
    const int tp = (int)(cdiv(d->Tq > d->Tk ? d->Tq : d->Tk, 32) * 32);
    const size_t lds = bwd_lds_bytes(tp);
    static bool lds_attr_set = false;
    if (!lds_attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_bf16_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds_bytes(TMAX));
      SEGCLIP_REQUIRE(e == hipSuccess, "attn_bwd bf16: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
      lds_attr_set = true;
    }
    hipLaunchKernelGGL(attn_bwd_bf16_kernel, dim3((unsigned)(d->B * d->H)), dim3(nw * 64), lds, stream, a);
    SEGCLIP_CHECK_LAUNCH("attn_bwd_bf16");
    return 0;
  }
  SEGCLIP_REQUIRE(d->colsum_part == nullptr, "attn_bwd f32: colsum_part is a bf16-path feature");
  SEGCLIP_REQUIRE(d->ws != nullptr, "attn_bwd f32: workspace required");
  const float* P = (const float*)d->stats;
  float* dP = (float*)d->ws;
  const int64_t pz1 = d->H * d->Tq * d->Tk, pz2 = d->Tq * d->Tk;
  segclip_gemm_desc g;
  int rc;
  // dV(key,d) = sum_q P[q][key] dO[q][d]
  base_gemm(g, d);
  g.A = P; g.B = d->dO; g.C = d->dV; g.M = d->Tk; g.N = d->hd; g.K = d->Tq;
  g.sam = 1; g.sak = d->Tk; g.sbn = 1; g.sbk = d->do_st; g.ldc = d->dv_st;
  g.bsA1 = pz1; g.bsA2 = pz2; g.bsB1 = d->do_sb; g.bsB2 = d->hd; g.bsC1 = d->dv_sb; g.bsC2 = d->hd;
  if ((rc = segclip_gemm_f32_launch(&g, stream))) return rc;
  // dP[q][key] = sum_d dO[q][d] V[key][d]
  base_gemm(g, d);
  g.A = d->dO; g.B = d->V; g.C = dP; g.M = d->Tq; g.N = d->Tk; g.K = d->hd;
  g.sam = d->do_st; g.sak = 1; g.sbn = d->v_st; g.sbk = 1; g.ldc = d->Tk;
  g.bsA1 = d->do_sb; g.bsA2 = d->hd; g.bsB1 = d->v_sb; g.bsB2 = d->hd; g.bsC1 = pz1; g.bsC2 = pz2;
  if ((rc = segclip_gemm_f32_launch(&g, stream))) return rc;
  const int64_t rows = d->B * d->H * d->Tq;
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, stream, P, dP, rows,
                     (int)d->Tk, d->scale);
  SEGCLIP_CHECK_LAUNCH("attn_softmax_bwd_rows");
  // dQ[q][d] = sum_key dS[q][key] K[key][d]
  base_gemm(g, d);
  g.A = dP; g.B = d->K; g.C = d->dQ; g.M = d->Tq; g.N = d->hd; g.K = d->Tk;
  g.sam = d->Tk; g.sak = 1; g.sbn = 1; g.sbk = d->k_st; g.ldc = d->dq_st;
  g.bsA1 = pz1; g.bsA2 = pz2; g.bsB1 = d->k_sb; g.bsB2 = d->hd; g.bsC1 = d->dq_sb; g.bsC2 = d->hd;
  if ((rc = segclip_gemm_f32_launch(&g, stream))) return rc;
  // dK[key][d] = sum_q dS[q][key] Q[q][d]
  base_gemm(g, d);
  g.A = dP; g.B = d->Q; g.C = d->dK; g.M = d->Tk; g.N = d->hd; g.K = d->Tq;
  g.sam = 1; g.sak = d->Tk; g.sbn = 1; g.sbk = d->q_st; g.ldc = d->dk_st;
  g.bsA1 = pz1; g.bsA2 = pz2; g.bsB1 = d->q_sb; g.bsB2 = d->hd; g.bsC1 = d->dk_sb; g.bsC2 = d->hd;
  return segclip_gemm_f32_launch(&g, stream);
}
