// Probe of ds_read_b64_tr_b8 (gfx950): which LDS byte lands in byte j of lane l when lane l supplies address addr[l]?
// Also checks the fp8 MFMA (32x32x16, e4m3) operand layout against a scalar reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) i32x2 lds_i32x2;

__global__ void probe(unsigned* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = mode == 0 ? (i & 0xff) : (i >> 8);
  __syncthreads();
  const i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(lds + lane * 8));
  out[lane * 2] = v[0];
  out[lane * 2 + 1] = v[1];
}

// D = A B^T style check: A[i][k], B[j][k] as fp8 e4m3 bytes; lane l holds A row (l&31), k = 8*(l>>5)..+7 (same for B)
__global__ void mfma_probe(const unsigned char* A, const unsigned char* B, float* D) {
  const int lane = threadIdx.x;
  const long a = *reinterpret_cast<const long*>(A + (lane & 31) * 16 + 8 * (lane >> 5));
  const long b = *reinterpret_cast<const long*>(B + (lane & 31) * 16 + 8 * (lane >> 5));
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = c[r];   // D[i][j], j = lane&31
}
__global__ void cvt_probe(const float* x, unsigned* out) {
  const int i = threadIdx.x;
  out[i] = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false);
}

static float e4m3_to_float(unsigned char v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f;
  if (e == 0) f = m / 8.0f * (1.0f / 64.0f);
  else if (e == 15 && m == 7) f = NAN;
  else f = (1.0f + m / 8.0f) * powf(2.0f, e - 7);
  return s ? -f : f;
}

int main() {
  unsigned *d, h[2][128];
  hipMalloc(&d, 128 * 4);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h[mode], d, 128 * 4, hipMemcpyDeviceToHost);
  }
  printf("ds_read_b64_tr_b8: lane l supplies address 8*l; source ADDRESS of each output byte j=0..7\n");
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 8; ++j) {
      const unsigned lo = (h[0][l * 2 + j / 4] >> (8 * (j % 4))) & 0xff, hi = (h[1][l * 2 + j / 4] >> (8 * (j % 4))) & 0xff;
      const unsigned addr = hi * 256 + lo;
      printf(" %3u(L%u.b%u)", addr, addr / 8, addr % 8);
    }
    printf("\n");
  }
  // fp8 MFMA layout + conversion check
  unsigned char hA[32 * 16], hB[32 * 16];
  float fx[64];
  for (int i = 0; i < 32 * 16; ++i) { hA[i] = (unsigned char)((i * 37 + 11) % 120); hB[i] = (unsigned char)((i * 53 + 7) % 120 | ((i % 3 == 0) ? 0x80 : 0)); }
  unsigned char *dA, *dB; float* dD; float hD[1024];
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 4096);
  hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      float ref = 0;
      for (int k = 0; k < 16; ++k) ref += e4m3_to_float(hA[i * 16 + k]) * e4m3_to_float(hB[j * 16 + k]);
      double e = fabs(hD[i * 32 + j] - ref) / (fabs(ref) + 1e-6);
      if (e > worst) worst = e;
    }
  printf("fp8 mfma 32x32x16 (D[i][j] = sum_k A[i][k] B[j][k], lane l: row l&31, k bytes 8*(l>>5)..): worst rel err %.3g\n", worst);
  for (int i = 0; i < 64; ++i) fx[i] = (i - 20) * 0.37f * (i % 5 == 0 ? 100.f : 1.f);
  float* dx; unsigned* dc; unsigned hc[32];
  hipMalloc(&dx, 256); hipMalloc(&dc, 128);
  hipMemcpy(dx, fx, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(32), 0, 0, dx, dc);
  hipMemcpy(hc, dc, 128, hipMemcpyDeviceToHost);
  printf("cvt_pk_fp8_f32 (x -> e4m3 -> float):");
  for (int i = 0; i < 12; ++i) printf(" %g->%g", fx[i], e4m3_to_float((hc[i / 2] >> (8 * (i % 2))) & 0xff));
  printf("\n   big:");
  for (int i = 0; i < 64; i += 5) printf(" %g->%g", fx[i], e4m3_to_float((hc[i / 2] >> (8 * (i % 2))) & 0xff));
  printf("\n");
  return 0;
}
