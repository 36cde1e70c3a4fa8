// Shared device/host helpers for the SegCLIP gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/segclip_hip.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define SEGCLIP_WAVE 64

void segclip_set_error(const char* fmt, ...);

#define SEGCLIP_CHECK_LAUNCH(name)                                              \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      segclip_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return (int)e__;                                                          \
    }                                                                           \
  } while (0)

#define SEGCLIP_REQUIRE(cond, ...)     \
  do {                                 \
    if (!(cond)) {                     \
      segclip_set_error(__VA_ARGS__);  \
      return SEGCLIP_ERR_INVALID;      \
    }                                  \
  } while (0)

// round-to-nearest-even fp32 -> bf16 (NaN preserved as quiet NaN)
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// two fp32 -> packed bf16x2 with the gfx950 hardware converter v_cvt_pk_bf16_f32 (round-to-nearest-even).  Written as a
// vector conversion, NOT as inline asm (rounds 1-4): hipcc selects the same instruction, and - unlike for an asm statement -
// pads its hazards.  gfx950 needs one wait state between a transcendental (v_exp_f32, v_rcp_f32 ...) and a VALU reading its
// result; with the asm form hipcc scheduled `v_exp_f32 v1, v1` directly in front of `v_cvt_pk_bf16_f32 v64, v0, v1` (72
// places in attention.hip alone) and the converter read the stale register: wrong P values on some code placements
// (found round 5 through the causal instance of the persistent attention forward).
typedef __bf16 segclip_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float segclip_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const segclip_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, segclip_bf16x2_t));
}

template <typename T> struct io;
template <> struct io<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct io<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// sigmoid(1.702 x) with the hardware exp2 / rcp (1 ulp): an IEEE division costs ~10 VALU ops per element and
// made the act' GEMM epilogue VALU-bound
__device__ __forceinline__ float sigmoid1702(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * x));  // 1.702 * log2(e)
}
__device__ __forceinline__ float act_quick_gelu(float x) { return x * sigmoid1702(x); }
__device__ __forceinline__ float act_quick_gelu_grad(float x) {
  const float s = sigmoid1702(x);
  return s * (1.0f + 1.702f * x * (1.0f - s));
}
__device__ __forceinline__ float act_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float act_gelu_erf_grad(float x) {
  float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float apply_act(int act, float x) {
  if (act == SEGCLIP_ACT_QUICK_GELU) return act_quick_gelu(x);
  if (act == SEGCLIP_ACT_GELU_ERF) return act_gelu_erf(x);
  return x;
}
__device__ __forceinline__ float apply_act_grad(int act, float x) {
  if (act == SEGCLIP_ACT_QUICK_GELU) return act_quick_gelu_grad(x);
  if (act == SEGCLIP_ACT_GELU_ERF) return act_gelu_erf_grad(x);
  return 1.0f;
}

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Timing ablations that produce GARBAGE results (SEGCLIP_ATTN_ABL, SEGCLIP_P8_EPI_ABL, SEGCLIP_P8_ABL, SEGCLIP_PQ_ABL) exist only in
// libraries built with `build.sh -DSEGCLIP_EXPERIMENTS`; a production build ignores those environment variables.
#include <stdlib.h>
// Kernel-selection / tuning switches of the library (SEGCLIP_GEMM_PQ, SEGCLIP_PQ_HALF, SEGCLIP_ATTN_FWD_PF, ...: A/B tools and
// profile scripts).  They are honoured only when SEGCLIP_TUNING=1 is set as well: a production process cannot have its kernels
// re-routed by a stray variable, and one that tries is told so once per variable.
static inline const char* segclip_tuning_env(const char* name) {
  static const bool on = [] { const char* t = getenv("SEGCLIP_TUNING"); return t != nullptr && atoi(t) != 0; }();
  const char* e = getenv(name);
  if (e != nullptr && !on) {
    fprintf(stderr, "segclip_hip: %s=%s ignored (tuning switches need SEGCLIP_TUNING=1)\n", name, e);
    return nullptr;
  }
  return e;
}
static inline int segclip_ablation_env(const char* name) {
#ifdef SEGCLIP_EXPERIMENTS
  const char* e = segclip_tuning_env(name);
  return e ? atoi(e) : 0;
#else
  (void)name;
  return 0;
#endif
}
