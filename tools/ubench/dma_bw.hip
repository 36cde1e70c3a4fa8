// micro-benchmark: L2 -> LDS global_load_lds throughput per CU for different access shapes / depths
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
template <int N> __device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if constexpr (N == 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
}
// each wave issues 4 x 1KiB DMA per "slice"; DEPTH slices in flight; ROWB = contiguous bytes per row piece
template <int ROWB, int DEPTH, bool TOLDS>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, long row_stride, long region, int iters, float* out) {
  __shared__ __attribute__((aligned(16))) char smem[131072];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int LPR = ROWB / 16;            // lanes per row piece
  constexpr int RPI = 64 / LPR;             // rows per instruction
  const long base = ((long)blockIdx.x * 7919 % 64) * 65536;  // spread CUs over the region
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long row = (long)((it * 32 + wave * 4 + i) * RPI + lane / LPR);
      long off = (base + row * row_stride + (long)(it & 1) * 0 + (lane % LPR) * 16) % region;
      if (TOLDS) __builtin_amdgcn_global_load_lds((gbl_void*)(src + off), (lds_void*)(smem + ((it % DEPTH) * 32 + wave * 4 + i) * 1024), 16, 0, 0);
      else { float4 v = *(const float4*)(src + off); acc += v.x; asm volatile("" :: "v"(v.y), "v"(v.z), "v"(v.w)); }
    }
    if (TOLDS) wait_vm<(DEPTH - 1) * 4>();
  }
  wait_vm<0>();
  if (out && acc == 123.f) out[0] = acc + smem[lane];
}
template <int ROWB, int DEPTH, bool TOLDS>
void run(const char* name, const char* d, long stride, long region) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, blocks = 256;
  k<ROWB, DEPTH, TOLDS><<<blocks, 512>>>(d, stride, region, 50, nullptr);
  hipEventRecord(e0);
  k<ROWB, DEPTH, TOLDS><<<blocks, 512>>>(d, stride, region, iters, nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double bytes = (double)blocks * iters * 32 * 1024;
  printf("%-44s %8.1f GB/s per CU  (%6.2f TB/s chip)\n", name, bytes / (ms * 1e-3) / blocks / 1e9, bytes / (ms * 1e-3) / 1e12);
}
int main() {
  const long region = 64l << 20;  // 64 MiB: MALL resident; per-CU windows overlap -> L2 hits
  char* d; hipMalloc(&d, region + (4 << 20)); hipMemset(d, 1, region + (4 << 20));
  run<1024, 4, true>("lds-dma contiguous 1KiB, depth 4", d, 1024, region);
  run<128, 4, true>("lds-dma 8 rows x 128B (stride 1536), depth 4", d, 1536, region);
  run<64, 4, true>("lds-dma 16 rows x 64B (stride 1536), depth 4", d, 1536, region);
  run<64, 2, true>("lds-dma 16 rows x 64B (stride 1536), depth 2", d, 1536, region);
  run<128, 2, true>("lds-dma 8 rows x 128B (stride 1536), depth 2", d, 1536, region);
  run<128, 4, false>("vgpr load 8 rows x 128B (stride 1536)", d, 1536, region);
  run<1024, 4, false>("vgpr load contiguous 1KiB", d, 1024, region);
  const long small = 2l << 20;  // 2 MiB: L2 resident everywhere
  run<128, 4, true>("L2-hot: lds-dma 8 rows x 128B, depth 4", d, 1536, small);
  run<64, 4, true>("L2-hot: lds-dma 16 rows x 64B, depth 4", d, 1536, small);
  run<1024, 4, true>("L2-hot: lds-dma contiguous, depth 4", d, 1024, small);
  return 0;
}
