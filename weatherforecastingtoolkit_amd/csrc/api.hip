// api.hip — library-level entry points of libwfae.so (version, errors, workspace sizing).
#include "common.h"

namespace wfae {

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

int wfae_version(void) { return 100; }  // 0.1.0

const char* wfae_last_error_string(void) { return wfae::err_buf(); }

size_t wfae_workspace_bytes(int64_t max_weight_elems) {
  // split-K slabs: up to 8 partial copies of the largest weight gradient;
  // BN / loss partials and SSIM maps fit comfortably in the 64 MiB floor.
  size_t w = (size_t)(max_weight_elems > 0 ? max_weight_elems : 0) * sizeof(float) * 8;
  const size_t floor_bytes = (size_t)64 << 20;
  return w > floor_bytes ? w : floor_bytes;
}

}  // extern "C"
