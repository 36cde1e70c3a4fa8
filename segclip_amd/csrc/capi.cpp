// C-ABI glue: version / error string / GEMM dispatch.  Compiled with hipcc like the kernels.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

int segclip_gemm_f32_launch(const segclip_gemm_desc* d, hipStream_t stream);
int segclip_gemm_bf16_launch(const segclip_gemm_desc* d, hipStream_t stream);
size_t segclip_gemm_bf16_ws_bytes(const segclip_gemm_desc* d);
int segclip_gemm_bf16_splits(const segclip_gemm_desc* d);

static thread_local char g_err[512] = "";

void segclip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int segclip_version(void) { return SEGCLIP_ABI_VERSION; }
extern "C" const char* segclip_last_error_string(void) { return g_err; }

extern "C" size_t segclip_gemm_ws_bytes(const segclip_gemm_desc* d) {
  if (d->b_dtype == SEGCLIP_BF16) return segclip_gemm_bf16_ws_bytes(d);
  return 0;
}

extern "C" int segclip_gemm_splits(const segclip_gemm_desc* d) {
  if (d->b_dtype == SEGCLIP_BF16 && !(d->a_dtype == SEGCLIP_F32 && d->b_dtype == SEGCLIP_F32)) return segclip_gemm_bf16_splits(d);
  return 1;
}

extern "C" int segclip_gemm(const segclip_gemm_desc* d, void* stream) {
  SEGCLIP_REQUIRE(d != nullptr, "gemm: null descriptor");
  SEGCLIP_REQUIRE(d->M >= 0 && d->N >= 0 && d->K >= 0, "gemm: negative size");
  if (d->M == 0 || d->N == 0) return 0;
  SEGCLIP_REQUIRE(d->A && d->B && d->C, "gemm: null operand");
  SEGCLIP_REQUIRE(!d->mul_dact || d->aux, "gemm: mul_dact needs aux");
  SEGCLIP_REQUIRE(d->aux_kind == 0 || ((d->aux_kind == 1 || d->aux_kind == 2) && d->act == SEGCLIP_ACT_QUICK_GELU) ||
                      (d->aux_kind == 2 && d->act == SEGCLIP_ACT_GELU_ERF),
                  "gemm: aux_kind 1 (aux = act'(pre-activation) in the output type) is implemented for QuickGELU only, aux_kind 2 for QuickGELU and erf-GELU");
  SEGCLIP_REQUIRE(d->aux_kind != 2 || (d->a_dtype == SEGCLIP_BF16 && d->b_dtype == SEGCLIP_BF16 && d->c_dtype == SEGCLIP_BF16),
                  "gemm: aux_kind 2 (one byte per element) needs bf16 operands and output");
  if (d->res_row_mod > 0 && !(d->a_dtype == SEGCLIP_BF16 && d->b_dtype == SEGCLIP_BF16)) {
    segclip_set_error("gemm: res_row_mod is a bf16-operand feature");
    return SEGCLIP_ERR_UNSUPPORTED;
  }
  if (d->a_dtype == SEGCLIP_F32 && d->b_dtype == SEGCLIP_F32) {
    SEGCLIP_REQUIRE(d->c_dtype == SEGCLIP_F32 && (!d->residual || d->r_dtype == SEGCLIP_F32),
                    "gemm f32: output / residual must be f32");
    return segclip_gemm_f32_launch(d, (hipStream_t)stream);
  }
  if (d->b_dtype == SEGCLIP_BF16) return segclip_gemm_bf16_launch(d, (hipStream_t)stream);
  segclip_set_error("gemm: unsupported dtype combination a=%d b=%d", d->a_dtype, d->b_dtype);
  return SEGCLIP_ERR_UNSUPPORTED;
}
