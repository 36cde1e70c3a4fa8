// Shared by the two bf16 GEMM kernels (register-staged gemm_bf16.hip, LDS-DMA gemm_bf16_dma.hip):
// argument block and the fused epilogue of one wave's 64x64 sub-tile (2x2 MFMA 32x32 accumulators).
#pragma once
#include "common.h"

namespace {

struct Args {
  const void* A; const bf16_t* B; void* C;
  const float* bias; const void* residual; void* aux;
  int64_t M, N, K, lda, ldb, ldc, ldr, ldaux;
  int64_t nb2, bsA1, bsA2, bsB1, bsB2, bsC1, bsC2, bsR1, bsR2;
  int c_dtype, r_dtype, act, mul_dact; float alpha;
  int nbx, nby;
  int splits; int64_t kper; float* slab;  // split-K: raw fp32 partial tiles go to slab[s][z][M][N]
  float* colsum_part;  // optional [M/64][N] fp32 partial column sums of the stored output (LDS epilogue only)
  int vec_epi;  // host-checked: every C / aux / residual / bias access of a full tile may be a 16-byte vector
  int colgroups; // 8-phase kernel: column groups of the tile order (see the kernel); 1 = plain row-major
  int abl;      // 8-phase kernel, timing experiments (SEGCLIP_P8_EPI_ABL): 1 = no epilogue (results garbage)
  int touch;    // 8-phase kernel: pre-touch the epilogue's side tile (SEGCLIP_P8_TOUCH, default on)
  int aux_kind; // 0: aux = pre-activation u (stored by EPI_ACT, differentiated by EPI_DACT); 1: aux = act'(u);
                // 2: aux = act'(u) as one byte per element (staged epilogue only, see EPI_ACT8)
  int slab_staged; // 8-phase kernel: split-K partial tiles through the staged epilogue (SEGCLIP_P8_SLAB_STAGED)
  int xw_epi;   // 8-phase kernel, bf16 outputs: cross-wave row pass (whole 128-byte lines per store; SEGCLIP_EPI_XW)
};


// ---- epilogue -------------------------------------------------------------------------------
// EPI_ACT8 / EPI_DACT8 (bf16 outputs, QuickGELU, staged epilogue of full tiles only): the saved derivative act'(u), which lies
// in [-0.10, 1.10], is kept as ONE BYTE per element, q = rint((act'(u) + 0.125) * 204): absolute error <= 0.0025, i.e. about
// bf16's relative error at the typical magnitude - and half the bytes of the largest side tensor of a block (M x 4D written
// by the c_fc forward, read by the c_proj data gradient: without ANY of that traffic the step is 1.4 ms shorter).
enum { EPI_PLAIN = 0, EPI_ACT = 1, EPI_DACT = 2, EPI_ACT8 = 3, EPI_DACT8 = 4 };
constexpr bool epi_is_act(int m) { return m == EPI_ACT || m == EPI_ACT8; }
constexpr bool epi_is_dact(int m) { return m == EPI_DACT || m == EPI_DACT8; }
constexpr float AUX8_SCALE = 204.0f, AUX8_OFF = 0.125f;

// forward activation epilogue: v (pre-activation) -> act(v); *side = what is kept for the backward: u, or act'(u) when
// aux_kind == 1 (QuickGELU only - segclip_gemm rejects aux_kind 1 with any other activation; keeping the erf-GELU
// derivative out of here also keeps the unrolled epilogues below the pragma-unroll size threshold)
__device__ __forceinline__ float act_with_side(int act, int aux_kind, float v, float* side) {
  if (act == SEGCLIP_ACT_QUICK_GELU) {
    const float sg = sigmoid1702(v);
    *side = aux_kind ? sg * (1.0f + 1.702f * v * (1.0f - sg)) : v;
    return v * sg;
  }
  *side = v;
  return apply_act(act, v);
}
__device__ __forceinline__ float dact_from_side(int act, int aux_kind, float side) {
  return aux_kind ? side : apply_act_grad(act, side);
}

template <typename CT> __device__ __forceinline__ float ldc_(const void* p, int64_t o);
template <> __device__ __forceinline__ float ldc_<bf16_t>(const void* p, int64_t o) { return bf2f(((const bf16_t*)p)[o]); }
template <> __device__ __forceinline__ float ldc_<float>(const void* p, int64_t o) { return ((const float*)p)[o]; }

// FULL: whole 128x128 tile in range (no per-element bounds checks; bf16 output stored as packed column
// pairs after a lane^1 exchange: 4-byte stores instead of 2-byte ones).
// jstride: distance between the two 32-column MFMA tiles of the sub-tile (32: adjacent; 128: the 8-phase kernel, whose
// waves own one 32-column strip in each 128-column half of the tile)
// (the (i, j) MFMA tile is a template parameter: with the four tiles in `#pragma unroll` loops the unrolled size can
// exceed the pragma-unroll threshold, the loops stay rolled and the accumulators are indexed at run time -> scratch)
template <typename CT, int MODE, bool FULL, int i, int j>
__device__ __forceinline__ void epilogue_tile(const Args& g, const f32x16 (&acc)[2][2], int64_t m0, int64_t n0, int wm,
                                              int wn, int lane, int64_t coff, int64_t roff, int jstride) {
  const int li = lane & 31, lk = lane >> 5;
  constexpr bool PAIR = FULL && sizeof(CT) == 2;
    {
      const int64_t n = n0 + wn * 64 + j * jstride + li;
      if (!FULL && n >= g.N) return;
      const float bv = (MODE != EPI_DACT && g.bias) ? g.bias[n] : 0.f;
      float val[16], pre[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        float v = g.alpha * acc[i][j][r];
        pre[r] = 0.f;
        if (FULL || m < g.M) {
          if (MODE == EPI_DACT) {
            v *= dact_from_side(g.act, g.aux_kind, ldc_<CT>(g.aux, coff + m * g.ldaux + n));
          } else {
            v += bv;
            if (MODE == EPI_ACT) v = act_with_side(g.act, g.aux_kind, v, &pre[r]);
            if (g.residual) {
              const int64_t o = roff + m * g.ldr + n;
              v += g.r_dtype == SEGCLIP_BF16 ? bf2f(((const bf16_t*)g.residual)[o]) : ((const float*)g.residual)[o];
            }
          }
        }
        val[r] = v;
      }
      if constexpr (PAIR) {
        const bool odd = lane & 1;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int64_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk + (odd ? 1 : 0);
          const float recv = __shfl_xor(odd ? val[r] : val[r + 1], 1, 64);
          const uint32_t w = odd ? pack2bf(recv, val[r + 1]) : pack2bf(val[r], recv);
          *reinterpret_cast<uint32_t*>((bf16_t*)g.C + coff + m * g.ldc + (n & ~(int64_t)1)) = w;
          if (MODE == EPI_ACT && g.aux) {
            const float rp = __shfl_xor(odd ? pre[r] : pre[r + 1], 1, 64);
            const uint32_t wp = odd ? pack2bf(rp, pre[r + 1]) : pack2bf(pre[r], rp);
            *reinterpret_cast<uint32_t*>((bf16_t*)g.aux + coff + m * g.ldaux + (n & ~(int64_t)1)) = wp;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          if (!FULL && m >= g.M) continue;
          if (sizeof(CT) == 2) ((bf16_t*)g.C)[coff + m * g.ldc + n] = f2bf(val[r]);
          else ((float*)g.C)[coff + m * g.ldc + n] = val[r];
          if (MODE == EPI_ACT && g.aux) {
            if (sizeof(CT) == 2) ((bf16_t*)g.aux)[coff + m * g.ldaux + n] = f2bf(pre[r]);
            else ((float*)g.aux)[coff + m * g.ldaux + n] = pre[r];
          }
        }
      }
    }
}

template <typename CT, int MODE, bool FULL>
__device__ __forceinline__ void epilogue(const Args& g, const f32x16 (&acc)[2][2], int64_t m0, int64_t n0, int wm, int wn,
                                         int lane, int64_t coff, int64_t roff, int jstride = 32) {
  epilogue_tile<CT, MODE, FULL, 0, 0>(g, acc, m0, n0, wm, wn, lane, coff, roff, jstride);
  epilogue_tile<CT, MODE, FULL, 0, 1>(g, acc, m0, n0, wm, wn, lane, coff, roff, jstride);
  epilogue_tile<CT, MODE, FULL, 1, 0>(g, acc, m0, n0, wm, wn, lane, coff, roff, jstride);
  epilogue_tile<CT, MODE, FULL, 1, 1>(g, acc, m0, n0, wm, wn, lane, coff, roff, jstride);
}

template <typename CT, bool FULL>
__device__ __forceinline__ void epilogue_mode(const Args& g, const f32x16 (&acc)[2][2], int64_t m0, int64_t n0, int wm,
                                              int wn, int lane, int64_t coff, int64_t roff, int jstride = 32) {
  if (g.mul_dact) epilogue<CT, EPI_DACT, FULL>(g, acc, m0, n0, wm, wn, lane, coff, roff, jstride);
  else if (g.act != SEGCLIP_ACT_NONE) epilogue<CT, EPI_ACT, FULL>(g, acc, m0, n0, wm, wn, lane, coff, roff, jstride);
  else epilogue<CT, EPI_PLAIN, FULL>(g, acc, m0, n0, wm, wn, lane, coff, roff, jstride);
}


// ---- LDS-staged epilogue of one wave's 64x64 sub-tile (full tiles, 16-byte aligned operands) ---------
// Phase 1: the wave parks its alpha-scaled fp32 accumulators in a private [64][68] fp32 LDS patch.
// Phase 2: each lane re-reads 8 (bf16 out) / 4 (fp32 out) consecutive columns of a row, applies
// bias / activation / act' / residual with 16-byte loads, and writes one 16-byte chunk: every global
// access of the epilogue is a full, coalesced line segment (the per-element form wrote 2-4 bytes/lane).
constexpr int EPI_PITCH = 68;
constexpr int EPI_WAVE_BYTES = 64 * EPI_PITCH * 4;  // 17408

__device__ __forceinline__ void unpack8(const u32x4 v, float* f) {
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[2 * j] = __uint_as_float(v[j] << 16); f[2 * j + 1] = __uint_as_float(v[j] & 0xffff0000u); }
}

// NSPLIT > 0: the sub-tile's local columns 0-31 / 32-63 live at global columns nw + 0..31 / nw + NSPLIT + 0..31.
// The epilogue of a sub-tile is three stages, so that a kernel with several sub-tiles per wave can interleave them
// (gemm_bf16_p8.hip issues the side loads of its second half-tile under the row pass of the first):
//   epi_side_load : ALL residual / saved-activation loads of the 64x64 sub-tile into <= 64 VGPRs (the operand-fragment
//                   registers are free by now): one memory round trip per sub-tile.  With batches of 4 rows (the first
//                   version) an fp32-residual epilogue paid 8 dependent round trips per tile: +26..57 us per launch.
//   epi_park      : accumulators -> the wave's private LDS patch
//   epi_rows      : row pass (LDS -> bias / activation / act' / residual -> 16-byte stores, column sums)
template <typename CT> struct EpiGeom {
  static constexpr int W = sizeof(CT) == 2 ? 8 : 4;   // columns per lane
  static constexpr int LPR = 64 / W;                   // lanes per row
  static constexpr int RPI = 64 / LPR;                 // rows per iteration
  static constexpr int NIT = 64 / RPI;                 // iterations: 8 (bf16 out) / 16 (fp32 out)
};
template <typename CT, int NSPLIT> __device__ __forceinline__ int64_t epi_col(int64_t nw, int cl) {
  constexpr int W = EpiGeom<CT>::W;
  return NSPLIT > 0 ? nw + ((cl * W) >> 5) * NSPLIT + ((cl * W) & 31) : nw + cl * W;
}

// Where a lane of the row pass reads its W values of a row (tl + row * EPI_PITCH) and where they go (global column n).
//   default: the wave's own patch, its own columns - with NSPLIT > 0 (8-phase kernel) a wave's 64 columns are two 32-column
//     strips NSPLIT apart, so a bf16 row segment of a wave is two 64-byte pieces: every store instruction writes half lines
//     and every 128-byte line of the output is written by two different waves at different times;
//   cross-wave (bf16 output, NSPLIT > 0; xw = patch of wave 0, all 8 patches parked and a workgroup barrier passed): wave
//     (wr, wc) takes the 64 CONTIGUOUS columns n0 + 64 wc .. + 63 of its 64 rows: strip 2 wc from the patch of wave
//     (wr, (2 wc) & 3), strip 2 wc + 1 from its neighbour's - whole 128-byte lines per row and store instruction.
struct EpiLane { const float* tl; int64_t n; int rl; };
template <typename CT, int NSPLIT>
__device__ __forceinline__ EpiLane epi_lane(const float* t, int64_t nw, int lane, const float* xw, int wr, int wc, int64_t n0) {
  using G = EpiGeom<CT>;
  const int cl = lane % G::LPR, rl = lane / G::LPR;
  if (sizeof(CT) == 2 && NSPLIT > 0 && xw != nullptr) {
    const int strip = 2 * wc + (cl >> 2);               // 32-column strip of the 256-wide tile (bf16: 4 lanes per strip)
    const int sw = wr * 4 + (strip & 3), half = strip >> 2;
    return EpiLane{xw + sw * (EPI_WAVE_BYTES / 4) + half * 32 + (cl & 3) * G::W, n0 + wc * 64 + cl * G::W, rl};
  }
  return EpiLane{t + cl * G::W, epi_col<CT, NSPLIT>(nw, cl), rl};
}

// Side registers of a sub-tile: NIT 16-byte words per lane, of the OUTPUT element type (the host routes a residual whose
// dtype differs from the output's to the per-element epilogue): u32x4 = 8 bf16 (W == 8) or f32x4 (W == 4).
template <typename CT> struct EpiSideT { typedef u32x4 type; };
template <> struct EpiSideT<float> { typedef f32x4 type; };

template <typename CT, int MODE, int NSPLIT>
__device__ __forceinline__ void epi_side_load(const Args& g, typename EpiSideT<CT>::type (&sr)[EpiGeom<CT>::NIT], int64_t mw,
                                              const EpiLane& L, int64_t coff, int64_t roff) {
  using G = EpiGeom<CT>;
  typedef typename EpiSideT<CT>::type ST;
  const int rl = L.rl;
  const int64_t n = L.n;
#pragma unroll
  for (int it = 0; it < G::NIT; ++it) {
    const int64_t m = mw + it * G::RPI + rl;
    if constexpr (MODE == EPI_DACT8) {
      const u32x2 q = *reinterpret_cast<const u32x2*>((const uint8_t*)g.aux + coff + m * g.ldaux + n);
      sr[it] = ST{};
      sr[it][0] = q[0]; sr[it][1] = q[1];
    } else if (MODE == EPI_DACT) sr[it] = *reinterpret_cast<const ST*>((const CT*)g.aux + coff + m * g.ldaux + n);
    else sr[it] = *reinterpret_cast<const ST*>((const CT*)g.residual + roff + m * g.ldr + n);
  }
}
template <typename CT> __device__ __forceinline__ void epi_side_get(const typename EpiSideT<CT>::type& w, float* f);
template <> __device__ __forceinline__ void epi_side_get<bf16_t>(const u32x4& w, float* f) { unpack8(w, f); }
template <> __device__ __forceinline__ void epi_side_get<float>(const f32x4& w, float* f) { f[0] = w[0]; f[1] = w[1]; f[2] = w[2]; f[3] = w[3]; }

__device__ __forceinline__ void epi_park(const Args& g, const f32x16 (&acc)[2][2], float* t, int lane) {
  const int li = lane & 31, lk = lane >> 5;
  float* tl = t + (4 * lk) * EPI_PITCH + li;     // lane-constant part of the address: the rest are immediates
  if (g.alpha == 1.0f) {                         // (uniform) the usual case: 64 multiplies less per sub-tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) tl[(i * 32 + (r & 3) + 8 * (r >> 2)) * EPI_PITCH + j * 32] = acc[i][j][r];
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) tl[(i * 32 + (r & 3) + 8 * (r >> 2)) * EPI_PITCH + j * 32] = g.alpha * acc[i][j][r];
}

// one row segment: v[W] (LDS values) -> stored output; returns nothing, accumulates csum
template <typename CT, int MODE>
__device__ __forceinline__ void epi_finish_row(const Args& g, float (&v)[EpiGeom<CT>::W], const float (&bias)[EpiGeom<CT>::W],
                                               float (&csum)[EpiGeom<CT>::W], int64_t m, int64_t n, int64_t coff) {
  constexpr int W = EpiGeom<CT>::W;
  if (!epi_is_dact(MODE)) {
#pragma unroll
    for (int c = 0; c < W; ++c) v[c] += bias[c];
  }
  if (epi_is_act(MODE)) {
    float sd[W];
#pragma unroll
    for (int c = 0; c < W; ++c) v[c] = act_with_side(g.act, MODE == EPI_ACT8 ? 1 : g.aux_kind, v[c], &sd[c]);
    if constexpr (MODE == EPI_ACT8) {
      u32x2 q = {0u, 0u};
#pragma unroll
      for (int c = 0; c < 8; ++c)
        q[c >> 2] = __builtin_amdgcn_cvt_pk_u8_f32((sd[c] + AUX8_OFF) * AUX8_SCALE, c & 3, q[c >> 2]);
      __builtin_nontemporal_store(q, reinterpret_cast<u32x2*>((uint8_t*)g.aux + coff + m * g.ldaux + n));
    } else if (g.aux) {
      if (sizeof(CT) == 2) {
        u32x4 p;
#pragma unroll
        for (int c = 0; c < 4; ++c) p[c] = pack2bf(sd[2 * c], sd[2 * c + 1]);
        __builtin_nontemporal_store(p, reinterpret_cast<u32x4*>((bf16_t*)g.aux + coff + m * g.ldaux + n));
      } else {
        __builtin_nontemporal_store(f32x4{sd[0], sd[1], sd[2], sd[3]},
                                    reinterpret_cast<f32x4*>((float*)g.aux + coff + m * g.ldaux + n));
      }
    }
  }
}
template <typename CT>
__device__ __forceinline__ void epi_store_row(const Args& g, const float (&v)[EpiGeom<CT>::W], int64_t m, int64_t n,
                                              int64_t coff) {
  if (sizeof(CT) == 2) {
    u32x4 p;
#pragma unroll
    for (int c = 0; c < 4; ++c) p[c] = pack2bf(v[2 * c], v[2 * c + 1]);
    // non-temporal: the output is not re-read by this kernel and should not displace the W tiles in L2
    // (measured -0.4 ms per training step)
    __builtin_nontemporal_store(p, reinterpret_cast<u32x4*>((bf16_t*)g.C + coff + m * g.ldc + n));
  } else {
    __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>((float*)g.C + coff + m * g.ldc + n));
  }
}
template <typename CT>
__device__ __forceinline__ void epi_colsum(const Args& g, const float (&csum)[EpiGeom<CT>::W], int64_t mw, int64_t n, int rl) {
  using G = EpiGeom<CT>;
  if (g.colsum_part) {  // column sums of this wave's 64 rows (bias gradient of the producing Linear)
#pragma unroll
    for (int c = 0; c < G::W; ++c) {
      float x = csum[c];
#pragma unroll
      for (int o = G::LPR; o < 64; o <<= 1) x += __shfl_xor(x, o, 64);
      if (rl == 0) g.colsum_part[(mw >> 6) * g.N + n + c] = x;
    }
  }
}
template <typename CT, int MODE>
__device__ __forceinline__ void epi_load_bias(const Args& g, float (&bias)[EpiGeom<CT>::W], int64_t n) {
  constexpr int W = EpiGeom<CT>::W;
#pragma unroll
  for (int c = 0; c < W; ++c) bias[c] = 0.f;
  if (!epi_is_dact(MODE) && g.bias) {
#pragma unroll
    for (int c = 0; c < W; c += 4) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(g.bias + n + c);
      bias[c] = b[0]; bias[c + 1] = b[1]; bias[c + 2] = b[2]; bias[c + 3] = b[3];
    }
  }
}

// row pass with side operands (act' multiplier or residual) held in sr[]: fully unrolled
template <typename CT, int MODE, int NSPLIT>
__device__ __forceinline__ void epi_rows_side(const Args& g, const typename EpiSideT<CT>::type (&sr)[EpiGeom<CT>::NIT],
                                              const EpiLane& L, int64_t mw, int64_t coff) {
  using G = EpiGeom<CT>;
  constexpr int W = G::W;
  const int rl = L.rl;
  const int64_t n = L.n;
  const float* t = L.tl;
  float bias[W], csum[W];
  epi_load_bias<CT, MODE>(g, bias, n);
#pragma unroll
  for (int c = 0; c < W; ++c) csum[c] = 0.f;
#pragma unroll
  for (int it = 0; it < G::NIT; ++it) {
    const int row = it * G::RPI + rl;
    const int64_t m = mw + row;
    float v[W];
#pragma unroll
    for (int c = 0; c < W; c += 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(t + row * EPI_PITCH + c);
      v[c] = a[0]; v[c + 1] = a[1]; v[c + 2] = a[2]; v[c + 3] = a[3];
    }
    float sd[W];   // exactly W entries: a wider array makes VectorCombine widen the side-register loads (-> scratch)
    if constexpr (MODE == EPI_DACT8) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t w = sr[it][c >> 2];
        const float qf = (float)((w >> (8 * (c & 3))) & 0xffu);   // v_cvt_f32_ubyteN
        v[c] *= qf * (1.0f / AUX8_SCALE) - AUX8_OFF;
      }
    } else {
      epi_side_get<CT>(sr[it], sd);
    }
    if constexpr (MODE == EPI_DACT8) {
    } else if (MODE == EPI_DACT) {
#pragma unroll
      for (int c = 0; c < W; ++c) v[c] *= dact_from_side(g.act, g.aux_kind, sd[c]);
    } else {
      epi_finish_row<CT, MODE>(g, v, bias, csum, m, n, coff);
#pragma unroll
      for (int c = 0; c < W; ++c) v[c] += sd[c];
    }
#pragma unroll
    for (int c = 0; c < W; ++c) csum[c] += v[c];
    epi_store_row<CT>(g, v, m, n, coff);
  }
  epi_colsum<CT>(g, csum, mw, n, rl);
}

// row pass without residual / act' operands (bias, activation + saved copy, column sums): rows are processed by a
// rolled loop, 4 at a time (fully unrolling it, as the side-operand form above does, spilled registers and cost the
// QuickGELU epilogue +25 %).
template <typename CT, int MODE, int NSPLIT>
__device__ __forceinline__ void epi_rows_plain(const Args& g, const EpiLane& L, int64_t mw, int64_t coff) {
  using G = EpiGeom<CT>;
  constexpr int W = G::W;
  const int rl = L.rl;
  const int64_t n = L.n;
  const float* t = L.tl;
  float bias[W], csum[W];
  epi_load_bias<CT, MODE>(g, bias, n);
#pragma unroll
  for (int c = 0; c < W; ++c) csum[c] = 0.f;
  constexpr int BATCH = 4;
#pragma unroll 1
  for (int it0 = 0; it0 < G::NIT; it0 += BATCH) {
#pragma unroll
    for (int bi = 0; bi < BATCH; ++bi) {
      const int row = (it0 + bi) * G::RPI + rl;
      const int64_t m = mw + row;
      float v[W];
#pragma unroll
      for (int c = 0; c < W; c += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(t + row * EPI_PITCH + c);
        v[c] = a[0]; v[c + 1] = a[1]; v[c + 2] = a[2]; v[c + 3] = a[3];
      }
      epi_finish_row<CT, MODE>(g, v, bias, csum, m, n, coff);
#pragma unroll
      for (int c = 0; c < W; ++c) csum[c] += v[c];
      epi_store_row<CT>(g, v, m, n, coff);
    }
  }
  epi_colsum<CT>(g, csum, mw, n, rl);
}

// barrier between parking and the row pass: wave-private patches need none (a wave's LDS operations execute in order);
// the cross-wave row pass reads its neighbour's patch
__device__ __forceinline__ void epi_sync(bool xw) {
  if (xw) __syncthreads(); else __builtin_amdgcn_wave_barrier();
}

// one sub-tile (gemm_bf16_dma.hip; the 8-phase kernel's sequential fp32 path)
template <typename CT, int MODE, int NSPLIT = 0>
__device__ __forceinline__ void epilogue_lds(const Args& g, const f32x16 (&acc)[2][2], float* t, int64_t mw, int64_t nw,
                                             int lane, int64_t coff, int64_t roff) {
  const EpiLane L = epi_lane<CT, NSPLIT>(t, nw, lane, nullptr, 0, 0, 0);
  if (!epi_is_dact(MODE) && !g.residual) {
    epi_park(g, acc, t, lane);
    __builtin_amdgcn_wave_barrier();
    epi_rows_plain<CT, MODE, NSPLIT>(g, L, mw, coff);
    return;
  }
  typename EpiSideT<CT>::type sr[EpiGeom<CT>::NIT];
  epi_side_load<CT, MODE, NSPLIT>(g, sr, mw, L, coff, roff);
  epi_park(g, acc, t, lane);
  __builtin_amdgcn_wave_barrier();
  epi_rows_side<CT, MODE, NSPLIT>(g, sr, L, mw, coff);
}

// two sub-tiles of one wave (gemm_bf16_p8.hip: rows mw0.. and mw1..): the side loads of the second are issued as soon
// as the first sub-tile's accumulators are parked, and land under the first row pass.  xw != nullptr (bf16 outputs; every
// wave of the workgroup is here): cross-wave row pass, see epi_lane.
template <typename CT, int MODE, int NSPLIT>
__device__ __forceinline__ void epilogue_lds2(const Args& g, const f32x16 (&acc0)[2][2], const f32x16 (&acc1)[2][2],
                                              float* t, int64_t mw0, int64_t mw1, int64_t nw, int lane, int64_t coff,
                                              int64_t roff, const float* xw = nullptr, int wr = 0, int wc = 0, int64_t n0 = 0) {
  // measured (tools/bench_epi.py, M = 50176): act' data gradient 315 -> 294 us (its side loads become whole lines too),
  // c_fc forward with the saved derivative 334 -> 327; the plain bias-only epilogues do not gain in isolation (qkv forward
  // 185 = 185, out_proj data gradient 87.7 -> 92: three more workgroup barriers per tile) but do in the step, where the
  // second stream competes for the L2's write ports: 45.23 -> 44.98 ms.  SEGCLIP_EPI_XW = 0 off / 1 not for the plain
  // epilogues / 2 (default) all bf16 outputs
  if (MODE == EPI_PLAIN && g.xw_epi < 2) xw = nullptr;
  const bool x = sizeof(CT) == 2 && NSPLIT > 0 && xw != nullptr;
  const EpiLane L = epi_lane<CT, NSPLIT>(t, nw, lane, xw, wr, wc, n0);
  if (!epi_is_dact(MODE) && !g.residual) {
    epi_park(g, acc0, t, lane);
    epi_sync(x);
    epi_rows_plain<CT, MODE, NSPLIT>(g, L, mw0, coff);
    epi_sync(x);
    epi_park(g, acc1, t, lane);
    epi_sync(x);
    epi_rows_plain<CT, MODE, NSPLIT>(g, L, mw1, coff);
    return;
  }
#ifndef EPI_F32_OVERLAP
#define EPI_F32_OVERLAP 0
#endif
  if (sizeof(CT) == 4 && !EPI_F32_OVERLAP) {
    // fp32 side operands are 64 VGPRs per sub-tile: with the second sub-tile's accumulators still live there is no room
    // for both, so the sub-tiles run one after the other (their side tile was touched into L2 / MALL at kernel start)
    epilogue_lds<CT, MODE, NSPLIT>(g, acc0, t, mw0, nw, lane, coff, roff);
    __builtin_amdgcn_wave_barrier();
    epilogue_lds<CT, MODE, NSPLIT>(g, acc1, t, mw1, nw, lane, coff, roff);
    return;
  }
  typename EpiSideT<CT>::type s0[EpiGeom<CT>::NIT], s1[EpiGeom<CT>::NIT];
  epi_side_load<CT, MODE, NSPLIT>(g, s0, mw0, L, coff, roff);
  epi_park(g, acc0, t, lane);
  epi_side_load<CT, MODE, NSPLIT>(g, s1, mw1, L, coff, roff);
  epi_sync(x);
  epi_rows_side<CT, MODE, NSPLIT>(g, s0, L, mw0, coff);
  epi_sync(x);
  epi_park(g, acc1, t, lane);
  epi_sync(x);
  epi_rows_side<CT, MODE, NSPLIT>(g, s1, L, mw1, coff);
}

template <typename CT, int NSPLIT = 0>
__device__ __forceinline__ void epilogue_lds_mode(const Args& g, const f32x16 (&acc)[2][2], float* t, int64_t mw,
                                                  int64_t nw, int lane, int64_t coff, int64_t roff) {
  if constexpr (sizeof(CT) == 2) {
    if (g.aux_kind == 2 && g.aux) {
      if (g.mul_dact) epilogue_lds<CT, EPI_DACT8, NSPLIT>(g, acc, t, mw, nw, lane, coff, roff);
      else epilogue_lds<CT, EPI_ACT8, NSPLIT>(g, acc, t, mw, nw, lane, coff, roff);
      return;
    }
  }
  if (g.mul_dact) epilogue_lds<CT, EPI_DACT, NSPLIT>(g, acc, t, mw, nw, lane, coff, roff);
  else if (g.act != SEGCLIP_ACT_NONE) epilogue_lds<CT, EPI_ACT, NSPLIT>(g, acc, t, mw, nw, lane, coff, roff);
  else epilogue_lds<CT, EPI_PLAIN, NSPLIT>(g, acc, t, mw, nw, lane, coff, roff);
}
template <typename CT, int NSPLIT>
__device__ __forceinline__ void epilogue_lds2_mode(const Args& g, const f32x16 (&acc0)[2][2], const f32x16 (&acc1)[2][2],
                                                   float* t, int64_t mw0, int64_t mw1, int64_t nw, int lane,
                                                   int64_t coff, int64_t roff, const float* xw = nullptr, int wr = 0,
                                                   int wc = 0, int64_t n0 = 0) {
  if constexpr (sizeof(CT) == 2) {
    if (g.aux_kind == 2 && g.aux) {
      if (g.mul_dact) epilogue_lds2<CT, EPI_DACT8, NSPLIT>(g, acc0, acc1, t, mw0, mw1, nw, lane, coff, roff, xw, wr, wc, n0);
      else epilogue_lds2<CT, EPI_ACT8, NSPLIT>(g, acc0, acc1, t, mw0, mw1, nw, lane, coff, roff, xw, wr, wc, n0);
      return;
    }
  }
  if (g.mul_dact) epilogue_lds2<CT, EPI_DACT, NSPLIT>(g, acc0, acc1, t, mw0, mw1, nw, lane, coff, roff, xw, wr, wc, n0);
  else if (g.act != SEGCLIP_ACT_NONE) epilogue_lds2<CT, EPI_ACT, NSPLIT>(g, acc0, acc1, t, mw0, mw1, nw, lane, coff, roff, xw, wr, wc, n0);
  else epilogue_lds2<CT, EPI_PLAIN, NSPLIT>(g, acc0, acc1, t, mw0, mw1, nw, lane, coff, roff, xw, wr, wc, n0);
}

}  // namespace
