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
};


// ---- epilogue -------------------------------------------------------------------------------
enum { EPI_PLAIN = 0, EPI_ACT = 1, EPI_DACT = 2 };

template <typename CT> __device__ __forceinline__ float ldc_(const void* p, int64_t o);
template <> __device__ __forceinline__ float ldc_<bf16_t>(const void* p, int64_t o) { return bf2f(((const bf16_t*)p)[o]); }
template <> __device__ __forceinline__ float ldc_<float>(const void* p, int64_t o) { return ((const float*)p)[o]; }

// FULL: whole 128x128 tile in range (no per-element bounds checks; bf16 output stored as packed column
// pairs after a lane^1 exchange: 4-byte stores instead of 2-byte ones).
// jstride: distance between the two 32-column MFMA tiles of the sub-tile (32: adjacent; 128: the 8-phase kernel, whose
// waves own one 32-column strip in each 128-column half of the tile)
template <typename CT, int MODE, bool FULL>
__device__ __forceinline__ void epilogue(const Args& g, const f32x16 (&acc)[2][2], int64_t m0, int64_t n0, int wm, int wn,
                                         int lane, int64_t coff, int64_t roff, int jstride = 32) {
  const int li = lane & 31, lk = lane >> 5;
  constexpr bool PAIR = FULL && sizeof(CT) == 2;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t n = n0 + wn * 64 + j * jstride + li;
      if (!FULL && n >= g.N) continue;
      const float bv = (MODE != EPI_DACT && g.bias) ? g.bias[n] : 0.f;
      float val[16], pre[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        float v = g.alpha * acc[i][j][r];
        pre[r] = 0.f;
        if (FULL || m < g.M) {
          if (MODE == EPI_DACT) {
            v *= apply_act_grad(g.act, ldc_<CT>(g.aux, coff + m * g.ldaux + n));
          } else {
            v += bv;
            if (MODE == EPI_ACT) { pre[r] = v; v = apply_act(g.act, v); }
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
template <typename CT, int MODE, int NSPLIT>
__device__ __forceinline__ void epilogue_lds_plain(const Args& g, const f32x16 (&acc)[2][2], float* t, int64_t mw,
                                                   int64_t nw, int lane, int64_t coff);

template <typename CT, int MODE, int NSPLIT = 0>
__device__ __forceinline__ void epilogue_lds(const Args& g, const f32x16 (&acc)[2][2], float* t, int64_t mw, int64_t nw,
                                             int lane, int64_t coff, int64_t roff) {
  if (MODE != EPI_DACT && !g.residual) {   // nothing to load besides the bias: the rolled, low-register form
    epilogue_lds_plain<CT, MODE, NSPLIT>(g, acc, t, mw, nw, lane, coff);
    return;
  }
  const int li = lane & 31, lk = lane >> 5;
  constexpr int W = sizeof(CT) == 2 ? 8 : 4;      // columns per lane
  constexpr int LPR = 64 / W;                      // lanes per row
  constexpr int RPI = 64 / LPR;                    // rows per iteration
  const int cl = lane % LPR, rl = lane / LPR;
  const int64_t n = NSPLIT > 0 ? nw + ((cl * W) >> 5) * NSPLIT + ((cl * W) & 31) : nw + cl * W;
  float bias[W];
#pragma unroll
  for (int c = 0; c < W; ++c) bias[c] = 0.f;
  if (MODE != EPI_DACT && g.bias) {
#pragma unroll
    for (int c = 0; c < W; c += 4) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(g.bias + n + c);
      bias[c] = b[0]; bias[c + 1] = b[1]; bias[c + 2] = b[2]; bias[c + 3] = b[3];
    }
  }
  // ALL residual / pre-activation loads of the 64x64 sub-tile are issued first (<= 64 VGPRs: the operand-fragment
  // registers are free by now), so their memory latency is paid once and overlaps the accumulator -> LDS pass.  With
  // batches of 4 rows (the first version) an fp32-residual epilogue paid 8 dependent round trips per tile: +26..57 us
  // per launch on the out_proj / c_proj GEMMs.
  constexpr int NIT = 64 / RPI;
  constexpr int RW = W / 4;  // 16-byte words per lane for an fp32 side operand
  f32x4 side_f[NIT][RW];   // fp32 residual
  u32x4 side_h[NIT];       // bf16 residual (W==8) / bf16 aux
  u32x2 side_q[NIT];       // bf16 residual when W==4
  f32x4 aux_f[NIT];        // fp32 aux (W==4)
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int64_t m = mw + it * RPI + rl;
    if (MODE == EPI_DACT) {
      if (sizeof(CT) == 2) side_h[it] = *reinterpret_cast<const u32x4*>((const bf16_t*)g.aux + coff + m * g.ldaux + n);
      else aux_f[it] = *reinterpret_cast<const f32x4*>((const float*)g.aux + coff + m * g.ldaux + n);
    } else if (g.residual) {
      const int64_t o = roff + m * g.ldr + n;
      if (g.r_dtype == SEGCLIP_BF16) {
        if (W == 8) side_h[it] = *reinterpret_cast<const u32x4*>((const bf16_t*)g.residual + o);
        else side_q[it] = *reinterpret_cast<const u32x2*>((const bf16_t*)g.residual + o);
      } else {
#pragma unroll
        for (int c = 0; c < RW; ++c) side_f[it][c] = *reinterpret_cast<const f32x4*>((const float*)g.residual + o + 4 * c);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        t[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * EPI_PITCH + j * 32 + li] = g.alpha * acc[i][j][r];
  __builtin_amdgcn_wave_barrier();
  float csum[W];
#pragma unroll
  for (int c = 0; c < W; ++c) csum[c] = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    {
      const int bi = it;
      const int row = it * RPI + rl;
      const int64_t m = mw + row;
      float v[W];
#pragma unroll
      for (int c = 0; c < W; c += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(t + row * EPI_PITCH + cl * W + c);
        v[c] = a[0]; v[c + 1] = a[1]; v[c + 2] = a[2]; v[c + 3] = a[3];
      }
      if (MODE == EPI_DACT) {
        float u[8];
        if (sizeof(CT) == 2) unpack8(side_h[bi], u);
        else { u[0] = aux_f[bi][0]; u[1] = aux_f[bi][1]; u[2] = aux_f[bi][2]; u[3] = aux_f[bi][3]; }
#pragma unroll
        for (int c = 0; c < W; ++c) v[c] *= apply_act_grad(g.act, u[c]);
      } else {
#pragma unroll
        for (int c = 0; c < W; ++c) v[c] += bias[c];
        if (MODE == EPI_ACT) {
          if (g.aux) {
            if (sizeof(CT) == 2) {
              u32x4 p;
#pragma unroll
              for (int c = 0; c < 4; ++c) p[c] = pack2bf(v[2 * c], v[2 * c + 1]);
              __builtin_nontemporal_store(p, reinterpret_cast<u32x4*>((bf16_t*)g.aux + coff + m * g.ldaux + n));
            } else {
              __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]},
                                          reinterpret_cast<f32x4*>((float*)g.aux + coff + m * g.ldaux + n));
            }
          }
#pragma unroll
          for (int c = 0; c < W; ++c) v[c] = apply_act(g.act, v[c]);
        }
        if (g.residual) {
          if (g.r_dtype == SEGCLIP_BF16) {
            float rr[8];
            if (W == 8) unpack8(side_h[bi], rr);
            else { rr[0] = __uint_as_float(side_q[bi][0] << 16); rr[1] = __uint_as_float(side_q[bi][0] & 0xffff0000u);
                   rr[2] = __uint_as_float(side_q[bi][1] << 16); rr[3] = __uint_as_float(side_q[bi][1] & 0xffff0000u); }
#pragma unroll
            for (int c = 0; c < W; ++c) v[c] += rr[c];
          } else {
#pragma unroll
            for (int c = 0; c < W; ++c) v[c] += side_f[bi][c / 4][c % 4];
          }
        }
      }
#pragma unroll
      for (int c = 0; c < W; ++c) csum[c] += v[c];
      if (sizeof(CT) == 2) {
        u32x4 p;
#pragma unroll
        for (int c = 0; c < 4; ++c) p[c] = pack2bf(v[2 * c], v[2 * c + 1]);
        // non-temporal: the output is not re-read by this kernel and should not displace the W tiles in L2
        // (measured -0.4 ms per training step)
        __builtin_nontemporal_store(p, reinterpret_cast<u32x4*>((bf16_t*)g.C + coff + m * g.ldc + n));
      } else {
        __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]},
                                    reinterpret_cast<f32x4*>((float*)g.C + coff + m * g.ldc + n));
      }
    }
  }
  if (g.colsum_part) {  // column sums of this wave's 64 rows (bias gradient of the producing Linear)
#pragma unroll
    for (int c = 0; c < W; ++c) {
      float x = csum[c];
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) x += __shfl_xor(x, o, 64);
      if (rl == 0) g.colsum_part[(mw >> 6) * g.N + n + c] = x;
    }
  }
}

// epilogue_lds without residual / act' operands (bias, activation + pre-activation copy, column sums): rows are
// processed by a rolled loop, 4 at a time (fully unrolling it, as the side-operand form above does, spilled registers
// and cost the QuickGELU epilogue +25 %).
template <typename CT, int MODE, int NSPLIT>
__device__ __forceinline__ void epilogue_lds_plain(const Args& g, const f32x16 (&acc)[2][2], float* t, int64_t mw,
                                                   int64_t nw, int lane, int64_t coff) {
  const int li = lane & 31, lk = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        t[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * EPI_PITCH + j * 32 + li] = g.alpha * acc[i][j][r];
  __builtin_amdgcn_wave_barrier();
  constexpr int W = sizeof(CT) == 2 ? 8 : 4;
  constexpr int LPR = 64 / W;
  constexpr int RPI = 64 / LPR;
  const int cl = lane % LPR, rl = lane / LPR;
  const int64_t n = NSPLIT > 0 ? nw + ((cl * W) >> 5) * NSPLIT + ((cl * W) & 31) : nw + cl * W;
  float bias[W];
#pragma unroll
  for (int c = 0; c < W; ++c) bias[c] = 0.f;
  if (g.bias) {
#pragma unroll
    for (int c = 0; c < W; c += 4) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(g.bias + n + c);
      bias[c] = b[0]; bias[c + 1] = b[1]; bias[c + 2] = b[2]; bias[c + 3] = b[3];
    }
  }
  float csum[W];
#pragma unroll
  for (int c = 0; c < W; ++c) csum[c] = 0.f;
  constexpr int NIT = 64 / RPI, BATCH = 4;
#pragma unroll 1
  for (int it0 = 0; it0 < NIT; it0 += BATCH) {
#pragma unroll
    for (int bi = 0; bi < BATCH; ++bi) {
      const int row = (it0 + bi) * RPI + rl;
      const int64_t m = mw + row;
      float v[W];
#pragma unroll
      for (int c = 0; c < W; c += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(t + row * EPI_PITCH + cl * W + c);
        v[c] = a[0]; v[c + 1] = a[1]; v[c + 2] = a[2]; v[c + 3] = a[3];
      }
#pragma unroll
      for (int c = 0; c < W; ++c) v[c] += bias[c];
      if (MODE == EPI_ACT) {
        if (g.aux) {
          if (sizeof(CT) == 2) {
            u32x4 p;
#pragma unroll
            for (int c = 0; c < 4; ++c) p[c] = pack2bf(v[2 * c], v[2 * c + 1]);
            __builtin_nontemporal_store(p, reinterpret_cast<u32x4*>((bf16_t*)g.aux + coff + m * g.ldaux + n));
          } else {
            __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]},
                                        reinterpret_cast<f32x4*>((float*)g.aux + coff + m * g.ldaux + n));
          }
        }
#pragma unroll
        for (int c = 0; c < W; ++c) v[c] = apply_act(g.act, v[c]);
      }
#pragma unroll
      for (int c = 0; c < W; ++c) csum[c] += v[c];
      if (sizeof(CT) == 2) {
        u32x4 p;
#pragma unroll
        for (int c = 0; c < 4; ++c) p[c] = pack2bf(v[2 * c], v[2 * c + 1]);
        __builtin_nontemporal_store(p, reinterpret_cast<u32x4*>((bf16_t*)g.C + coff + m * g.ldc + n));
      } else {
        __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]},
                                    reinterpret_cast<f32x4*>((float*)g.C + coff + m * g.ldc + n));
      }
    }
  }
  if (g.colsum_part) {
#pragma unroll
    for (int c = 0; c < W; ++c) {
      float x = csum[c];
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) x += __shfl_xor(x, o, 64);
      if (rl == 0) g.colsum_part[(mw >> 6) * g.N + n + c] = x;
    }
  }
}

template <typename CT, int NSPLIT = 0>
__device__ __forceinline__ void epilogue_lds_mode(const Args& g, const f32x16 (&acc)[2][2], float* t, int64_t mw,
                                                  int64_t nw, int lane, int64_t coff, int64_t roff) {
  if (g.mul_dact) epilogue_lds<CT, EPI_DACT, NSPLIT>(g, acc, t, mw, nw, lane, coff, roff);
  else if (g.act != SEGCLIP_ACT_NONE) epilogue_lds<CT, EPI_ACT, NSPLIT>(g, acc, t, mw, nw, lane, coff, roff);
  else epilogue_lds<CT, EPI_PLAIN, NSPLIT>(g, acc, t, mw, nw, lane, coff, roff);
}

}  // namespace
