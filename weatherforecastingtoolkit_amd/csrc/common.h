// common.h — shared helpers for the libwfae.so HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/wfae.h"

namespace wfae {

// thread-local last-error text (wfae_last_error_string)
char* err_buf();
int fail(int code, const char* fmt, ...);
int matmul_precision();  // api.hip: WFAE_PRECISION_*

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(WFAE_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return WFAE_OK;
}

#define WFAE_REQUIRE(cond, code, ...) \
  do {                                \
    if (!(cond)) return ::wfae::fail(code, __VA_ARGS__); \
  } while (0)

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum of a double for blocks of up to 1024 threads; result valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double* sm /* >= 16 doubles */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += sm[i];
  }
  return r;
}

// exact (erf) GELU and its derivative — nn.GELU(approximate='none')
__device__ __forceinline__ float gelu_f(float u) {
  return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f));
}
// d/du gelu(u) = Phi(u) + u phi(u).  Phi through Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 in erf, i.e. 7.5e-8 in
// Phi — below fp32 resolution of the O(1) derivative): its exp(-x^2), x = u/sqrt(2), IS the exp(-u^2/2) of phi(u), so
// the whole derivative costs one v_exp, one v_rcp and a 5-term Horner chain instead of erff + expf (the backward
// BatchNorm kernels were VALU-bound on those two calls).
__device__ __forceinline__ float gelu_grad_f(float u) {
  const float ax = fabsf(u) * 0.70710678118654752440f;
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
  const float e = __expf(-0.5f * u * u);
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float half_erfc = 0.5f * poly * t * e;            // 0.5 * erfc(|x|)
  const float cdf = u >= 0.f ? 1.0f - half_erfc : half_erfc;
  return fmaf(u * 0.39894228040143267794f, e, cdf);
}
__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + expf(-v)); }

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Conv2d(C, 1, 3, padding=1): the decoder's full-resolution output convolution (c1conv.hip)
int c1conv3_fwd(const float* x, const float* w, const float* bias, float* y, int NB, int C, int H, int W, hipStream_t st);
int c1_wgrad_mfma(int flip, const float* big, const float* small, float* dw, int NB, int C, int H, int W, int accumulate,
                  void* ws, size_t ws_bytes, hipStream_t st);   // dconv.hip: weight gradients with one channel count = 1
int c1conv3_wgrad(const float* dy, const float* x, float* dw, int NB, int C, int H, int W, int accumulate, void* ws,
                  size_t ws_bytes, hipStream_t st);

// Winograd transforms of the 4x4 stride-2 convolutions (wino.hip); variant 0 = F(2x2,2x2) (N = 3, M = 2),
// variant 1 = F(4x4,2x2) (N = 5, M = 4); T = NB*Hlo*Wlo/M^2 tiles, N*N transform positions xi
int wino_in(int variant, const float* hi, float* V, int NB, int Chi, int Hlo, int Wlo, hipStream_t st);      // -> V[xi][4Chi][T]
int wino_in_t(int variant, const float* dV, float* hi, int NB, int Chi, int Hlo, int Wlo, hipStream_t st);   // adjoint
int wino_out(int variant, const float* M, float* lo, int NB, int Clo, int Hlo, int Wlo, hipStream_t st);     // M[xi][Clo][T] ->
int wino_out_t(int variant, const float* lo, float* Mt, int NB, int Clo, int Hlo, int Wlo, hipStream_t st);  // adjoint
int wino_weights(int variant, const float* w, float* U, int Clo, int Chi, hipStream_t st);                   // -> U[xi][Clo][4Chi]
int wino_weights_t(int variant, const float* dU, float* dw, int Clo, int Chi, int beta, hipStream_t st);     // G^T dU G

// out = (beta ? out : 0) + sum over `splits` partial slabs of MN floats (+ bias_n[i % N]); fixed order.
int slab_reduce(const float* slab, float* out, const float* bias_n, long MN, int N, int splits, int beta,
                hipStream_t st, int transpose_m = 0);

// pixel-parallel VALU weight gradient of the grouped 3x3 conv (dconv.hip); WFAE_ERR_UNSUPPORTED if the shape is not covered
int gconv3_wgrad_valu(const float* dy, const float* x, float* dw, int NB, int C, int H, int W, int groups,
                      int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
int gconv3_wgrad_mfma(const float* dy, const float* x, float* dw, int NB, int C, int H, int W, int groups,
                      int accumulate, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace wfae
