// api.hip — library-level entry points of libwfae.so (version, errors, workspace sizing).
#include "common.h"
#include <atomic>
#include <stdlib.h>

namespace wfae {

// process-wide arithmetic mode of the MFMA GEMM family, the counterpart of torch.set_float32_matmul_precision
// (which the reference sets once at start-up, experiments/ae_v2/train.py:270): 0 = fp32 MFMA, 1 = bf16 operands
static std::atomic<int> g_matmul_precision{0};
int matmul_precision() { return g_matmul_precision.load(std::memory_order_relaxed); }

// fp32 GEMMs on the bf16 matrix pipe with exact three-plane operands (splitgemm.hip; in-register split in gemm.hip):
// on by default, WFAE_SPLIT_GEMM=0 or wfae_set_split_gemm(0) keeps every GEMM on v_mfma_f32_32x32x2_f32
static std::atomic<int> g_split_gemm{-1};
bool split_gemm_enabled() {
  int v = g_split_gemm.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("WFAE_SPLIT_GEMM");
    v = (e && e[0] == '0') ? 0 : 1;
    g_split_gemm.store(v, std::memory_order_relaxed);
  }
  return v == 1 && matmul_precision() == WFAE_PRECISION_FP32;
}

char* err_buf() {
  static thread_local char buf[512] = "ok";
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace wfae

extern "C" {

int wfae_version(void) { return 103; }  // 0.1.3 (see the ABI history in include/wfae.h)

int wfae_set_matmul_precision(int mode) {
  WFAE_REQUIRE(mode == WFAE_PRECISION_FP32 || mode == WFAE_PRECISION_BF16, WFAE_ERR_BAD_SHAPE,
               "set_matmul_precision: unknown mode %d", mode);
  wfae::g_matmul_precision.store(mode, std::memory_order_relaxed);
  return WFAE_OK;
}

int wfae_get_matmul_precision(void) { return wfae::matmul_precision(); }

int wfae_set_split_gemm(int on) {
  wfae::g_split_gemm.store(on ? 1 : 0, std::memory_order_relaxed);
  return WFAE_OK;
}

int wfae_get_split_gemm(void) { return wfae::split_gemm_enabled() ? 1 : 0; }

const char* wfae_last_error_string(void) { return wfae::err_buf(); }

size_t wfae_workspace_bytes(int64_t max_weight_elems) {
  // split-K slabs: up to 8 partial copies of the largest weight gradient;
  // BN / loss partials and SSIM maps fit comfortably in the 64 MiB floor.
  size_t w = (size_t)(max_weight_elems > 0 ? max_weight_elems : 0) * sizeof(float) * 8;
  const size_t floor_bytes = (size_t)64 << 20;
  return w > floor_bytes ? w : floor_bytes;
}

}  // extern "C"
