// Grouped 64x64 linear layers on channel-last rows (the kernel-1 grouped Conv1d's of the learnable-center stage,
// modules/module_seg_vit.py:266,269,299,302: k_conv / v_conv, groups = heads, 64 channels per group), bf16, HBM-bound.
//
//   out_o(m, g*64 + n) = sum_i sum_k in_i(m, g*64 + k) * W[i][o][g*64 + n][k]        i < NIN, o < NOUT
//
// NIN = 1, NOUT = 2: k_conv and v_conv of the SAME input in one pass (the input is read once);
// NIN = 2, NOUT = 1: their data gradient dn = dk Wk + dv Wv in one pass (the caller hands over the transposed weights).
// As batched GEMMs of M x 64 x 64 these ran on 256 x 128 tiles with one K-tile: 171 us (forward, each) / 275 us (data
// gradient, each) at M = 50176 for 77 MB in + 77 MB out; this kernel streams rows at HBM speed.
//
// One wave owns 32 rows and walks the groups: per group 4 operand fragments of the rows (16-byte loads, a row's 128-byte
// group slice is consumed by the four k chunks), the 64 x 64 weight block straight from L2 (96 KiB per weight set, hot), 8 MFMAs
// per (input, output) pair with SWAPPED operands (result block transposed: lane = row, registers = 4 consecutive columns),
// bf16 packing in-lane, and the 32 x 64 output block through a wave-private 4-KiB LDS patch so that it leaves as whole
// 128-byte lines.  The next group's row fragments are in flight while the current group is multiplied.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int GL_WAVES = 4, GL_ROWS = 32, GL_HD = 64;

struct GLArgs {
  const bf16_t* in[2]; bf16_t* out[2];
  const bf16_t* W[2][2];        // [input][output]: (groups*64, 64) row-major, row = g*64 + n, column = k
  int64_t M; int64_t ld_in[2], ld_out[2];
  int groups;
};

__device__ __forceinline__ bf16x8_t ld_frag(const bf16_t* p) { return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4*>(p)); }

template <int NIN, int NOUT>
__global__ __launch_bounds__(GL_WAVES * 64) void group_linear_kernel(GLArgs a) {
  __shared__ __attribute__((aligned(16))) char patch[GL_WAVES][NOUT][GL_ROWS * 128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const int64_t row0 = ((int64_t)blockIdx.x * GL_WAVES + wave) * GL_ROWS;
  if (row0 >= a.M) return;
  const int64_t row = row0 + li < a.M ? row0 + li : a.M - 1;      // clamped: rows beyond M are computed and not stored
  // blockIdx.y: a contiguous share of the groups (more waves in flight: a wave's walk is a chain of load -> MFMA -> LDS -> store;
  // 6 shares: 101 + 89 -> 81 + 77 us for the two launches of a step, SEGCLIP_GL64_GSPLIT)
  const int gper = (a.groups + (int)gridDim.y - 1) / (int)gridDim.y;
  const int gbeg = (int)blockIdx.y * gper, gend = gbeg + gper < a.groups ? gbeg + gper : a.groups;
  if (gbeg >= gend) return;
  const bf16_t* xin[NIN];
#pragma unroll
  for (int i = 0; i < NIN; ++i) xin[i] = a.in[i] + row * a.ld_in[i] + 8 * lk;
  bf16x8_t xf[NIN][4], xn[NIN][4];
#pragma unroll
  for (int i = 0; i < NIN; ++i)
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) xn[i][kc] = ld_frag(xin[i] + gbeg * GL_HD + kc * 16);
  for (int g = gbeg; g < gend; ++g) {
#pragma unroll
    for (int i = 0; i < NIN; ++i)
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) xf[i][kc] = xn[i][kc];
    if (g + 1 < gend) {
#pragma unroll
      for (int i = 0; i < NIN; ++i)
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) xn[i][kc] = ld_frag(xin[i] + (g + 1) * GL_HD + kc * 16);
    }
    f32x16 acc[NOUT][2];
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[o][nt][r] = 0.f;
#pragma unroll
    for (int i = 0; i < NIN; ++i)
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        const bf16_t* w = a.W[i][o] + (int64_t)(g * GL_HD + li) * GL_HD + 8 * lk;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int kc = 0; kc < 4; ++kc)
            acc[o][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(w + nt * 32 * GL_HD + kc * 16), xf[i][kc], acc[o][nt], 0, 0, 0);
      }
    // transposed result block: lane (li, lk) holds row li, columns nt*32 + 8*gq + 4*lk + 0..3 in registers gq*4 .. +3
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      char* pt = patch[wave][o];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          u32x2 t;
          t[0] = pack2bf(acc[o][nt][gq * 4 + 0], acc[o][nt][gq * 4 + 1]);
          t[1] = pack2bf(acc[o][nt][gq * 4 + 2], acc[o][nt][gq * 4 + 3]);
          *reinterpret_cast<u32x2*>(pt + li * 128 + (((nt * 4 + gq) ^ (li & 7)) << 4) + lk * 8) = t;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      const char* pt = patch[wave][o];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + (lane >> 3), c = lane & 7;
        const u32x4 v = *reinterpret_cast<const u32x4*>(pt + r * 128 + ((c ^ (r & 7)) << 4));
        if (row0 + r < a.M) *reinterpret_cast<u32x4*>(a.out[o] + (row0 + r) * a.ld_out[o] + g * GL_HD + c * 8) = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the patch is rewritten by the next group
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

// out_o = sum_i in_i x W[i][o] per 64-channel group (see the top of the file).  n_in, n_out in {1, 2} with n_in * n_out <= 2;
// w[i * n_out + o]: bf16 (groups*64, 64); every pointer 16-byte aligned, row pitches multiples of 8 elements.
extern "C" int segclip_group_linear64(const void* const* in, const int64_t* ld_in, int n_in, void* const* out,
                                      const int64_t* ld_out, int n_out, const void* const* w, int64_t M, int groups,
                                      void* stream) {
  SEGCLIP_REQUIRE((n_in == 1 && (n_out == 1 || n_out == 2)) || (n_in == 2 && n_out == 1),
                  "group_linear64: unsupported combination n_in=%d n_out=%d", n_in, n_out);
  SEGCLIP_REQUIRE(groups >= 1 && M >= 0, "group_linear64: bad shape");
  if (M == 0) return 0;
  GLArgs a = {};
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  for (int i = 0; i < n_in; ++i) {
    SEGCLIP_REQUIRE(in[i] && al(in[i]) && ld_in[i] % 8 == 0 && ld_in[i] >= (int64_t)groups * GL_HD, "group_linear64: input %d misaligned", i);
    a.in[i] = (const bf16_t*)in[i]; a.ld_in[i] = ld_in[i];
  }
  for (int o = 0; o < n_out; ++o) {
    SEGCLIP_REQUIRE(out[o] && al(out[o]) && ld_out[o] % 8 == 0 && ld_out[o] >= (int64_t)groups * GL_HD, "group_linear64: output %d misaligned", o);
    a.out[o] = (bf16_t*)out[o]; a.ld_out[o] = ld_out[o];
  }
  for (int i = 0; i < n_in; ++i)
    for (int o = 0; o < n_out; ++o) {
      const void* p = w[i * n_out + o];
      SEGCLIP_REQUIRE(p && al(p), "group_linear64: weight (%d, %d) misaligned", i, o);
      a.W[i][o] = (const bf16_t*)p;
    }
  a.M = M; a.groups = groups;
  static const int gsplit_env = [] { const char* e = segclip_tuning_env("SEGCLIP_GL64_GSPLIT"); const int v = e ? atoi(e) : 6; return v < 1 ? 1 : v; }();
  const int gsplit = gsplit_env < groups ? gsplit_env : groups;
  const dim3 grid((unsigned)cdiv(M, GL_WAVES * GL_ROWS), (unsigned)gsplit), block(GL_WAVES * 64);
  if (n_in == 1 && n_out == 1) hipLaunchKernelGGL((group_linear_kernel<1, 1>), grid, block, 0, (hipStream_t)stream, a);
  else if (n_in == 1) hipLaunchKernelGGL((group_linear_kernel<1, 2>), grid, block, 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((group_linear_kernel<2, 1>), grid, block, 0, (hipStream_t)stream, a);
  SEGCLIP_CHECK_LAUNCH("group_linear64");
  return 0;
}
