// Training-step tail for gfx950: multi-tensor gradient norm, fused AdaptAdamW step, NaN-skip bookkeeping.
// Replaces the reference's per-tensor Python loop (modules/optimization_adamw.py:111-174, ~300 tensors x 8
// elementwise launches) and the host round-trips of train_epoch (main_task_align.py:323-347) with
// ceil(T/32)+2 launches and no synchronisation.  Pure HBM streaming: 16 B/element read, 12 B/element written
// (+2 B for the bf16 shadow).
#include <math.h>

#include "common.h"

namespace {

constexpr int kTensorsPerLaunch = 32;
constexpr int kThreads = 256;
constexpr int kChunk = 16384;  // elements per workgroup: 256 lanes x float4 x 16

struct NormLaunch {
  const float* g[kTensorsPerLaunch];
  int64_t n[kTensorsPerLaunch];
  int32_t first_block[kTensorsPerLaunch + 1];
  int32_t count;
  int32_t ws_base;  // index of this launch's first partial in ws
};

__device__ __forceinline__ int find_tensor(const int32_t* first_block, int count, int b) {
  int t = 0;
#pragma unroll 1
  for (int i = 1; i < count; ++i) t += (b >= first_block[i]);
  return t;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(kThreads) void sqnorm_partial_kernel(NormLaunch L, float* __restrict__ ws) {
  __shared__ float red[4];
  const int t = find_tensor(L.first_block, L.count, blockIdx.x);
  const int64_t base = (int64_t)(blockIdx.x - L.first_block[t]) * kChunk;
  const int64_t n = L.n[t];
  const float* __restrict__ g = L.g[t] + base;
  const int64_t left = n - base < kChunk ? n - base : kChunk;
  float acc = 0.f;
  const int64_t nv = left >> 2;
  for (int64_t i = threadIdx.x; i < nv; i += kThreads) {
    const f32x4 x = reinterpret_cast<const f32x4*>(g)[i];
    acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
  }
  for (int64_t i = (nv << 2) + threadIdx.x; i < left; i += kThreads) acc += g[i] * g[i];
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) ws[L.ws_base + blockIdx.x] = s;
}

__global__ __launch_bounds__(kThreads) void sqnorm_final_kernel(const float* __restrict__ ws, int n,
                                                                 segclip_train_ctrl* ctrl, float max_norm) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += kThreads) acc += ws[i];
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) {
    ctrl->grad_sqnorm = s;
    float coef = 1.0f;
    if (max_norm > 0.f) coef = fminf(1.0f, max_norm / (sqrtf(s) + 1e-6f));  // clip_grad_norm_: clamp(max=1)
    ctrl->clip_coef = coef;
  }
}

struct StepLaunch {
  segclip_adamw_tensor t[kTensorsPerLaunch];
  int32_t first_block[kTensorsPerLaunch + 1];
  int32_t count;
  int32_t zero_grads;
};

struct GroupTable {
  segclip_adamw_group g[16];
};

// the three schedules of optimization_adamw.py:26-45, evaluated in double like the host code does
__device__ double schedule_value(int kind, double x, double warmup, double lr_start, double lr_end) {
  if (kind == SEGCLIP_SCHED_WARMUP_COSINE) {
    if (x < warmup) return (x * (1.0 - lr_start) / warmup) + lr_start;
    const double nx = (x - warmup) / (1.0 - warmup);
    return lr_end + 0.5 * (1.0 - lr_end) * (1.0 + cos(3.141592653589793 * nx));
  }
  if (kind == SEGCLIP_SCHED_WARMUP_CONSTANT) return x < warmup ? x / warmup : 1.0;
  if (x < warmup) return x / warmup;
  return fmax((x - 1.0) / (warmup - 1.0), 0.0);
}

struct StepScalars {
  float b1, one_m_b1, b2, one_m_b2, sqrt_bc2, eps, decay, step_size, coef;
  int skip;
};

__global__ __launch_bounds__(kThreads) void adamw_step_kernel(StepLaunch L, GroupTable G,
                                                              const segclip_train_ctrl* __restrict__ ctrl,
                                                              const float* __restrict__ loss) {
  __shared__ StepScalars sc;
  const int ti = find_tensor(L.first_block, L.count, blockIdx.x);
  const segclip_adamw_tensor T = L.t[ti];
  if (threadIdx.x == 0) {
    const segclip_adamw_group g = G.g[T.group];
    int skip = 0;
    int step = T.step;
    float coef = 1.0f;
    if (ctrl) {
      step -= ctrl->nan_skips;
      coef = ctrl->clip_coef;
    }
    if (loss && isnan(*loss)) skip = 1;
    if (step < 1) step = 1;
    const double bc1 = 1.0 - pow(g.b1, (double)step);
    const double bc2 = 1.0 - pow(g.b2, (double)step);
    double lr = g.lr;
    if (g.t_total != -1)
      lr = g.lr * schedule_value(g.schedule, (double)step / (double)g.t_total, g.warmup, g.lr_start, g.lr_end);
    sc.b1 = (float)g.b1;
    sc.one_m_b1 = (float)(1.0 - g.b1);
    sc.b2 = (float)g.b2;
    sc.one_m_b2 = (float)(1.0 - g.b2);
    sc.sqrt_bc2 = (float)sqrt(bc2);
    sc.eps = (float)g.eps;
    sc.decay = (float)(1.0 - lr * g.weight_decay);
    sc.step_size = (float)(lr / bc1);
    sc.coef = coef;
    sc.skip = skip;
  }
  __syncthreads();
  const StepScalars s = sc;
  const int64_t base = (int64_t)(blockIdx.x - L.first_block[ti]) * kChunk;
  const int64_t left = T.n - base < kChunk ? T.n - base : kChunk;
  float* __restrict__ p = T.param + base;
  float* __restrict__ g = T.grad + base;
  float* __restrict__ m = T.exp_avg + base;
  float* __restrict__ v = T.exp_avg_sq + base;
  bf16_t* __restrict__ sh = T.shadow_bf16 ? (bf16_t*)T.shadow_bf16 + base : nullptr;
  const int64_t nv = left >> 2;
  if (s.skip) {
    if (L.zero_grads) {
      for (int64_t i = threadIdx.x; i < nv; i += kThreads) reinterpret_cast<f32x4*>(g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int64_t i = (nv << 2) + threadIdx.x; i < left; i += kThreads) g[i] = 0.f;
    }
    if (sh)  // keep the shadow consistent with the (unchanged) parameter even on the very first iteration
      for (int64_t i = threadIdx.x; i < left; i += kThreads) sh[i] = f2bf(p[i]);
    return;
  }
  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    gg *= s.coef;
    mm = mm * s.b1 + s.one_m_b1 * gg;
    vv = vv * s.b2 + (s.one_m_b2 * gg) * gg;
    const float denom = sqrtf(vv) / s.sqrt_bc2 + s.eps;
    pp = pp * s.decay - s.step_size * (mm / denom);
  };
  for (int64_t i = threadIdx.x; i < nv; i += kThreads) {
    const f32x4 p4 = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 g4 = reinterpret_cast<const f32x4*>(g)[i];
    const f32x4 m4 = reinterpret_cast<f32x4*>(m)[i];
    const f32x4 v4 = reinterpret_cast<f32x4*>(v)[i];
    float pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) upd(pp[e], gg[e], mm[e], vv[e]);
    reinterpret_cast<f32x4*>(p)[i] = f32x4{pp[0], pp[1], pp[2], pp[3]};
    reinterpret_cast<f32x4*>(m)[i] = f32x4{mm[0], mm[1], mm[2], mm[3]};
    reinterpret_cast<f32x4*>(v)[i] = f32x4{vv[0], vv[1], vv[2], vv[3]};
    if (L.zero_grads) reinterpret_cast<f32x4*>(g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (sh) reinterpret_cast<u32x2*>(sh)[i] = u32x2{pack2bf(pp[0], pp[1]), pack2bf(pp[2], pp[3])};
  }
  for (int64_t i = (nv << 2) + threadIdx.x; i < left; i += kThreads) {
    float pp = p[i], mm = m[i], vv = v[i];
    upd(pp, g[i], mm, vv);
    p[i] = pp;
    m[i] = mm;
    v[i] = vv;
    if (L.zero_grads) g[i] = 0.f;
    if (sh) sh[i] = f2bf(pp);
  }
}

__global__ void train_finish_kernel(segclip_train_ctrl* ctrl, const float* loss, float* logit_scale, float clamp_max) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (loss) {
    const float l = *loss;
    const bool bad = isnan(l);
    ctrl->nan_skips += bad ? 1 : 0;
    ctrl->loss_sum += bad ? 0.f : l;
    ctrl->last_loss = l;
  }
  ctrl->steps += 1;
  if (logit_scale) *logit_scale = fminf(*logit_scale, clamp_max);
}

struct CastLaunch {
  const float* src[kTensorsPerLaunch];
  bf16_t* dst[kTensorsPerLaunch];
  int64_t n[kTensorsPerLaunch];
  int32_t first_block[kTensorsPerLaunch + 1];
  int32_t count;
};

__global__ __launch_bounds__(kThreads) void multi_cast_bf16_kernel(CastLaunch L) {
  const int t = find_tensor(L.first_block, L.count, blockIdx.x);
  const int64_t base = (int64_t)(blockIdx.x - L.first_block[t]) * kChunk;
  const int64_t left = L.n[t] - base < kChunk ? L.n[t] - base : kChunk;
  const float* __restrict__ src = L.src[t] + base;
  bf16_t* __restrict__ dst = L.dst[t] + base;
  const int64_t nv = left >> 2;
  for (int64_t i = threadIdx.x; i < nv; i += kThreads) {
    const f32x4 x = reinterpret_cast<const f32x4*>(src)[i];
    reinterpret_cast<u32x2*>(dst)[i] = u32x2{pack2bf(x.x, x.y), pack2bf(x.z, x.w)};
  }
  for (int64_t i = (nv << 2) + threadIdx.x; i < left; i += kThreads) dst[i] = f2bf(src[i]);
}

struct AddLaunch {
  float* dst[kTensorsPerLaunch];
  const float* src[kTensorsPerLaunch];
  int64_t n[kTensorsPerLaunch];
  int32_t first_block[kTensorsPerLaunch + 1];
  int32_t count;
};

__global__ __launch_bounds__(kThreads) void multi_add_f32_kernel(AddLaunch L) {
  const int t = find_tensor(L.first_block, L.count, blockIdx.x);
  const int64_t base = (int64_t)(blockIdx.x - L.first_block[t]) * kChunk;
  const int64_t left = L.n[t] - base < kChunk ? L.n[t] - base : kChunk;
  const float* __restrict__ src = L.src[t] + base;
  float* __restrict__ dst = L.dst[t] + base;
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
    const int64_t nv = left >> 2;
    for (int64_t i = threadIdx.x; i < nv; i += kThreads) {
      const f32x4 a = reinterpret_cast<const f32x4*>(dst)[i], b = reinterpret_cast<const f32x4*>(src)[i];
      reinterpret_cast<f32x4*>(dst)[i] = a + b;
    }
    for (int64_t i = (nv << 2) + threadIdx.x; i < left; i += kThreads) dst[i] += src[i];
  } else {
    for (int64_t i = threadIdx.x; i < left; i += kThreads) dst[i] += src[i];
  }
}

int64_t blocks_of(int64_t n) { return cdiv(n, kChunk); }

}  // namespace

extern "C" size_t segclip_grad_sqnorm_ws_bytes(const int64_t* n, int64_t count) {
  int64_t b = 0;
  for (int64_t i = 0; i < count; ++i) b += blocks_of(n[i]);
  return (size_t)(b > 0 ? b : 1) * sizeof(float);
}

extern "C" int segclip_grad_sqnorm(const float* const* grads, const int64_t* n, int64_t count, float* ws,
                                   segclip_train_ctrl* ctrl, float max_norm, void* stream) {
  SEGCLIP_REQUIRE(ctrl && (count == 0 || (grads && n && ws)), "segclip_grad_sqnorm: null argument");
  hipStream_t st = (hipStream_t)stream;
  int64_t total_blocks = 0;
  for (int64_t i0 = 0; i0 < count;) {
    NormLaunch L;
    int c = 0;
    int32_t nb = 0;
    for (; i0 < count && c < kTensorsPerLaunch; ++i0) {
      if (n[i0] <= 0) continue;
      SEGCLIP_REQUIRE(grads[i0], "segclip_grad_sqnorm: null gradient pointer at %lld", (long long)i0);
      L.g[c] = grads[i0];
      L.n[c] = n[i0];
      L.first_block[c] = nb;
      nb += (int32_t)blocks_of(n[i0]);
      ++c;
    }
    if (c == 0) continue;
    for (int j = c; j <= kTensorsPerLaunch; ++j) L.first_block[j] = nb;
    L.count = c;
    L.ws_base = (int32_t)total_blocks;
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nb), dim3(kThreads), 0, st, L, ws);
    total_blocks += nb;
  }
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(kThreads), 0, st, ws, (int)total_blocks, ctrl, max_norm);
  SEGCLIP_CHECK_LAUNCH("segclip_grad_sqnorm");
  return 0;
}

extern "C" int segclip_adamw_step(const segclip_adamw_tensor* tensors, int64_t count, const segclip_adamw_group* groups,
                                  int64_t ngroups, const segclip_train_ctrl* ctrl, const float* loss, int zero_grads,
                                  void* stream) {
  SEGCLIP_REQUIRE(count == 0 || (tensors && groups), "segclip_adamw_step: null argument");
  SEGCLIP_REQUIRE(ngroups >= 0 && ngroups <= 16, "segclip_adamw_step: at most 16 param groups (got %lld)", (long long)ngroups);
  hipStream_t st = (hipStream_t)stream;
  GroupTable G;
  for (int64_t i = 0; i < ngroups; ++i) {
    const segclip_adamw_group& g = groups[i];
    SEGCLIP_REQUIRE(g.schedule >= 0 && g.schedule <= 2, "segclip_adamw_step: unknown schedule %d", g.schedule);
    SEGCLIP_REQUIRE(g.b1 >= 0 && g.b1 < 1 && g.b2 >= 0 && g.b2 < 1 && g.eps >= 0 && g.lr >= 0,
                    "segclip_adamw_step: invalid hyper-parameters in group %lld", (long long)i);
    G.g[i] = g;
  }
  for (int64_t i0 = 0; i0 < count;) {
    StepLaunch L;
    int c = 0;
    int32_t nb = 0;
    for (; i0 < count && c < kTensorsPerLaunch; ++i0) {
      const segclip_adamw_tensor& t = tensors[i0];
      if (t.n <= 0) continue;
      SEGCLIP_REQUIRE(t.param && t.grad && t.exp_avg && t.exp_avg_sq, "segclip_adamw_step: null pointer in tensor %lld",
                      (long long)i0);
      SEGCLIP_REQUIRE(t.group >= 0 && t.group < ngroups, "segclip_adamw_step: tensor %lld: bad group %d", (long long)i0, t.group);
      SEGCLIP_REQUIRE((((uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq) & 15) == 0 &&
                          ((uintptr_t)t.shadow_bf16 & 7) == 0,
                      "segclip_adamw_step: tensor %lld is not 16-byte aligned", (long long)i0);
      L.t[c] = t;
      L.first_block[c] = nb;
      nb += (int32_t)blocks_of(t.n);
      ++c;
    }
    if (c == 0) continue;
    for (int j = c; j <= kTensorsPerLaunch; ++j) L.first_block[j] = nb;
    L.count = c;
    L.zero_grads = zero_grads;
    hipLaunchKernelGGL(adamw_step_kernel, dim3(nb), dim3(kThreads), 0, st, L, G, ctrl, loss);
  }
  SEGCLIP_CHECK_LAUNCH("segclip_adamw_step");
  return 0;
}

extern "C" int segclip_train_step_finish(segclip_train_ctrl* ctrl, const float* loss, float* logit_scale, float clamp_max,
                                         void* stream) {
  SEGCLIP_REQUIRE(ctrl, "segclip_train_step_finish: null ctrl");
  hipLaunchKernelGGL(train_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ctrl, loss, logit_scale, clamp_max);
  SEGCLIP_CHECK_LAUNCH("segclip_train_step_finish");
  return 0;
}

extern "C" int segclip_multi_cast_bf16(const float* const* src, void* const* dst, const int64_t* n, int64_t count,
                                       void* stream) {
  SEGCLIP_REQUIRE(count == 0 || (src && dst && n), "segclip_multi_cast_bf16: null argument");
  hipStream_t st = (hipStream_t)stream;
  for (int64_t i0 = 0; i0 < count;) {
    CastLaunch L;
    int c = 0;
    int32_t nb = 0;
    for (; i0 < count && c < kTensorsPerLaunch; ++i0) {
      if (n[i0] <= 0) continue;
      SEGCLIP_REQUIRE(src[i0] && dst[i0], "segclip_multi_cast_bf16: null pointer in tensor %lld", (long long)i0);
      SEGCLIP_REQUIRE(((uintptr_t)src[i0] & 15) == 0 && ((uintptr_t)dst[i0] & 7) == 0,
                      "segclip_multi_cast_bf16: tensor %lld is not aligned", (long long)i0);
      L.src[c] = src[i0];
      L.dst[c] = (bf16_t*)dst[i0];
      L.n[c] = n[i0];
      L.first_block[c] = nb;
      nb += (int32_t)blocks_of(n[i0]);
      ++c;
    }
    if (c == 0) continue;
    for (int j = c; j <= kTensorsPerLaunch; ++j) L.first_block[j] = nb;
    L.count = c;
    hipLaunchKernelGGL(multi_cast_bf16_kernel, dim3(nb), dim3(kThreads), 0, st, L);
  }
  SEGCLIP_CHECK_LAUNCH("segclip_multi_cast_bf16");
  return 0;
}

extern "C" int segclip_multi_add_f32(float* const* dst, const float* const* src, const int64_t* n, int64_t count, void* stream) {
  SEGCLIP_REQUIRE(count == 0 || (src && dst && n), "segclip_multi_add_f32: null argument");
  hipStream_t st = (hipStream_t)stream;
  for (int64_t i0 = 0; i0 < count;) {
    AddLaunch L;
    int c = 0;
    int32_t nb = 0;
    for (; i0 < count && c < kTensorsPerLaunch; ++i0) {
      if (n[i0] <= 0) continue;
      SEGCLIP_REQUIRE(src[i0] && dst[i0], "segclip_multi_add_f32: null pointer in tensor %lld", (long long)i0);
      SEGCLIP_REQUIRE(((uintptr_t)src[i0] & 3) == 0 && ((uintptr_t)dst[i0] & 3) == 0,
                      "segclip_multi_add_f32: tensor %lld is not aligned", (long long)i0);
      L.dst[c] = dst[i0];
      L.src[c] = src[i0];
      L.n[c] = n[i0];
      L.first_block[c] = nb;
      nb += (int32_t)blocks_of(n[i0]);
      ++c;
    }
    if (c == 0) continue;
    for (int j = c; j <= kTensorsPerLaunch; ++j) L.first_block[j] = nb;
    L.count = c;
    hipLaunchKernelGGL(multi_add_f32_kernel, dim3(nb), dim3(kThreads), 0, st, L);
  }
  SEGCLIP_CHECK_LAUNCH("segclip_multi_add_f32");
  return 0;
}
