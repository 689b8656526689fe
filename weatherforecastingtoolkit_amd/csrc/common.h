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
// Phi(u) from the tail: 1 - h for u >= 0, h below.  The choice is made on u's SIGN BIT with a bit-field insert, not with
// v_cmp + v_cndmask through VCC: same values (at u = -0 both branches are 0.5), and no lane mask travels between two vector
// instructions.  With the compare form, csrc/c1rb.hip's prologue (two packed evaluations per dword, the low element's compare two
// wait states in front of its select as hipcc pads it) gave results that differed between identical launches on gfx950: the
// LOW bf16 of a dword took the wrong branch in most lanes of one instruction, about one launch in ten
// (tools/debug_c1rb_repeat.py; a full s_waitcnt or s_nops around the MFMAs changed nothing).
__device__ __forceinline__ float gelu_cdf_select(float u, float h) {
  const int neg = __builtin_bit_cast(int, u) >> 31;   // all ones below zero
  const float a = 1.0f - h;
  float r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(neg), "v"(h), "v"(a));   // (neg & h) | (~neg & a); written out, the compiler turns it back into v_cmp + v_cndmask
  return r;
}
__device__ __forceinline__ float gelu_f(float u) {
  float e;
  const float h = gelu_tail_f(u, &e);
  return u * gelu_cdf_select(u, h);
}
__device__ __forceinline__ float gelu_grad_f(float u) {
  float e;
  const float h = gelu_tail_f(u, &e);
  const float cdf = gelu_cdf_select(u, h);
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

// ---- activation storage type: fp32, or bf16 bit patterns (torch.bfloat16) in the bf16-storage mode of 'medium' precision
// (BASELINE config 5: the reference's bf16 regime).  All arithmetic stays fp32: a kernel templated on the element type T
// reads W = 16 / sizeof(T) elements per 16-byte access (4 floats or 8 bf16 — every lane keeps moving whole 16-byte pieces,
// the width this chip's memory pipeline is built for), widens them to fp32 registers and rounds results to nearest even
// (v_cvt_pk_bf16_f32) on the way out.
typedef unsigned short bf16_t;
typedef float wfae_vf4 __attribute__((ext_vector_type(4)));
typedef unsigned wfae_vu4 __attribute__((ext_vector_type(4)));
typedef float wfae_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wfae_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {   // a -> low half, b -> high half
  const wfae_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, wfae_bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
template <typename T> struct ElemW { static constexpr int W = 16 / (int)sizeof(T); };
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return __builtin_bit_cast(float, (unsigned)*p << 16); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = (bf16_t)(pack_bf16(v, 0.f) & 0xffffu); }
// one 16-byte access: 4 floats / 8 bf16 -> v[0..W); NT: nontemporal
template <bool NT = false>
__device__ __forceinline__ void ldv(const float* p, float (&v)[4]) {
  const wfae_vf4 r = NT ? __builtin_nontemporal_load(reinterpret_cast<const wfae_vf4*>(p)) : *reinterpret_cast<const wfae_vf4*>(p);
  v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
}
template <bool NT = false>
__device__ __forceinline__ void ldv(const bf16_t* p, float (&v)[8]) {
  const wfae_vu4 r = NT ? __builtin_nontemporal_load(reinterpret_cast<const wfae_vu4*>(p)) : *reinterpret_cast<const wfae_vu4*>(p);
  v[0] = bf16_lo(r.x); v[1] = bf16_hi(r.x); v[2] = bf16_lo(r.y); v[3] = bf16_hi(r.y);
  v[4] = bf16_lo(r.z); v[5] = bf16_hi(r.z); v[6] = bf16_lo(r.w); v[7] = bf16_hi(r.w);
}
template <bool NT = false>
__device__ __forceinline__ void stv(float* p, const float (&v)[4]) {
  const wfae_vf4 r = {v[0], v[1], v[2], v[3]};
  if (NT) __builtin_nontemporal_store(r, reinterpret_cast<wfae_vf4*>(p));
  else *reinterpret_cast<wfae_vf4*>(p) = r;
}
template <bool NT = false>
__device__ __forceinline__ void stv(bf16_t* p, const float (&v)[8]) {
  const wfae_vu4 r = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
  if (NT) __builtin_nontemporal_store(r, reinterpret_cast<wfae_vu4*>(p));
  else *reinterpret_cast<wfae_vu4*>(p) = r;
}
// half-width access: 4 elements (16 bytes of fp32, 8 bytes of bf16) — loaders whose thread map is built on quads
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
  const uint2 r = *reinterpret_cast<const uint2*>(p);
  return make_float4(bf16_lo(r.x), bf16_hi(r.x), bf16_lo(r.y), bf16_hi(r.y));
}
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float2 ld2(const bf16_t* p) {
  const unsigned r = *reinterpret_cast<const unsigned*>(p);
  return make_float2(bf16_lo(r), bf16_hi(r));
}
__device__ __forceinline__ void st2(float* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }
__device__ __forceinline__ void st2(bf16_t* p, float2 v) { *reinterpret_cast<unsigned*>(p) = pack_bf16(v.x, v.y); }
// value a reader of a tensor of element type T sees after v was stored
__device__ __forceinline__ float rounded_as(const float*, float v) { return v; }
__device__ __forceinline__ float rounded_as(const bf16_t*, float v) { return bf16_lo(pack_bf16(v, 0.f)); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
}

// Conv2d(C, 1, 3, padding=1): the decoder's full-resolution output convolution (c1conv.hip)
int c1_wgrad_mfma(int flip, const float* big, const float* small, float* dw, int NB, int C, int H, int W, int accumulate,
                  void* ws, size_t ws_bytes, hipStream_t st);   // dconv.hip: weight gradients with one channel count = 1
int c1_wgrad_mfma(int flip, const unsigned short* big, const float* small, float* dw, int NB, int C, int H, int W, int accumulate,
                  void* ws, size_t ws_bytes, hipStream_t st);   // `big` stored as bf16
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
// bf16-stored tensors (hi / lo as bf16 bit patterns): the operand transforms read them, the result transforms write them
int wino_in_split(int variant, const unsigned short* hi, unsigned short* V3, int planes, int NB, int Chi, int Hlo, int Wlo, hipStream_t st);
int wino_out_t_split(int variant, const unsigned short* lo, unsigned short* Mt3, int planes, int NB, int Clo, int Hlo, int Wlo, hipStream_t st);
int wino_in_t(int variant, const float* dV, unsigned short* hi, int NB, int Chi, int Hlo, int Wlo, hipStream_t st);
int wino_in_t_stats(int variant, const float* dV, unsigned short* hi, int NB, int Chi, int Hlo, int Wlo, double* part, hipStream_t st);
int wino_out(int variant, const float* M, unsigned short* lo, int NB, int Clo, int Hlo, int Wlo, hipStream_t st);
int wino_out_stats(int variant, const float* M, unsigned short* lo, int NB, int Clo, int Hlo, int Wlo, double* part, hipStream_t st);
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
int gconv3_wgrad_mfma(const unsigned short* dy, const unsigned short* x, float* dw, int NB, int C, int H, int W, int groups,
                      int accumulate, void* ws, size_t ws_bytes, hipStream_t st);   // bf16-stored activations

}  // namespace wfae
