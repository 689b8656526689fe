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
bool split_gemm_enabled();  // api.hip: fp32 GEMMs on the bf16 pipe with split operands (only at WFAE_PRECISION_FP32)

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

// GELU(approximate='none') = u * Phi(u) and its derivative Phi(u) + u * phi(u).
// Phi through a rational-exponential form of the upper tail,  1 - Phi(|u|) = t Q(t) exp(-u^2 / 2),  t = 1 / (1 + p |u|),
// Q of degree 5 (a degree-6 fit in t, p = 0.275, Lawson-weighted least squares on [0, 6.2]: |fit error| <= 5e-9, ten times
// below fp32 resolution; the classic Abramowitz & Stegun 7.1.26 has 7.5e-8).  One v_rcp, one v_exp, six FMAs: ~15
// vector instructions against ~38 for 0.5 u (1 + erff(u / sqrt 2)) through the device library — erff was what made
// rebuilding the activation inside GEMM operand loaders cost more than the HBM pass it saves (tools/kbench.py --only
// fuse).  Accuracy against fp64 (tools/gelu_accuracy.py, 4.4 M points in [-9, 9]): relative error <= 4.1e-7 for u > 0 —
// torch's own fp32 CPU GELU: 3.7e-7 — and <= 4e-6 for u in [-3, 0), where 1 + erf cancels and torch's fp32 result is
// off by up to 2.4e-4; the tail form has no cancellation.  The exp(-u^2/2) is shared with phi(u) in the derivative.
__device__ __forceinline__ float gelu_tail_f(float u, float* e_out) {   // 1 - Phi(|u|)
  const float au = fabsf(u);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.275f, au, 1.0f));   // v_rcp_f32 (1 ulp); __frcp_rn expands to the 12-instruction IEEE division
  const float e = __expf(-0.5f * u * u);
  float q = -0.1148583822191914f;
  q = fmaf(q, t, 0.45199892303145267f);
  q = fmaf(q, t, -0.3309937454509166f);
  q = fmaf(q, t, 0.33393257052088204f);
  q = fmaf(q, t, 0.04211136686506703f);
  q = fmaf(q, t, 0.11780927197708843f);
  *e_out = e;
  return q * t * e;
}
__device__ __forceinline__ float gelu_f(float u) {
  float e;
  const float h = gelu_tail_f(u, &e);
  return u * (u >= 0.f ? 1.0f - h : h);
}
__device__ __forceinline__ float gelu_grad_f(float u) {
  float e;
  const float h = gelu_tail_f(u, &e);
  const float cdf = u >= 0.f ? 1.0f - h : h;
  return fmaf(u * 0.39894228040143267794f, e, cdf);
}
__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + expf(-v)); }

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// x == h + m + l exactly, each a bf16 (round to nearest even; both residuals are exact fp32 differences): the operand form
// of the split GEMMs (splitgemm.hip)
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  const __bf16 bh = (__bf16)x;
  const float r1 = x - (float)bh;
  const __bf16 bm = (__bf16)r1;
  const float r2 = r1 - (float)bm;
  const __bf16 bl = (__bf16)r2;
  h = __builtin_bit_cast(unsigned short, bh);
  m = __builtin_bit_cast(unsigned short, bm);
  l = __builtin_bit_cast(unsigned short, bl);
}

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
// wino_out that also leaves per-channel fp64 partial sums of lo in part[(split * Clo + l) * 2 + {0,1}]
int wino_out_stats(int variant, const float* M, float* lo, int NB, int Clo, int Hlo, int Wlo, double* part, hipStream_t st);
int wino_out_stat_splits(int variant, int NB, int Hlo, int Wlo);   // also the split count of wino_in_t_stats
int wino_in_t_stats(int variant, const float* dV, float* hi, int NB, int Chi, int Hlo, int Wlo, double* part, hipStream_t st);
int wino_weights(int variant, const float* w, float* U, int Clo, int Chi, hipStream_t st);                   // -> U[xi][Clo][4Chi]
// bf16-plane forms of the GEMM operands (splitgemm.hip) and the GEMM itself
int wino_in_split(int variant, const float* hi, unsigned short* V3, int planes, int NB, int Chi, int Hlo, int Wlo, hipStream_t st);
int wino_out_t_split(int variant, const float* lo, unsigned short* Mt3, int planes, int NB, int Clo, int Hlo, int Wlo, hipStream_t st);
int wino_weights_split(int variant, const float* w, unsigned short* U3, unsigned short* Ut3, int planes, int Clo, int Chi,
                       hipStream_t st);
int split_gemm(int kind, int planes, const unsigned short* A, const unsigned short* B, float* C, int M, int N, int K,
               long a_plane, long b_plane, long a_y, long b_y, long c_y, int batches, int k_per_split, int splits,
               long c_split, hipStream_t st, const char* what);
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
