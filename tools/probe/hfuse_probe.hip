// Can ONE launch keep the matrix pipe and HBM busy at the same time?  A grid whose blocks take one of two roles by index:
// role M = the LDS -> MFMA loop of the GEMM core (64x64 wave tiles, 4 waves, fp32 32x32x2 MFMA, no global traffic),
// role S = a streaming pass (float4 read of two tensors, some VALU, float4 write: the shape of bn_act_bwd_dx).
// Timed: M blocks alone, S blocks alone, the interleaved grid (every second block S), and the two launches back to
// back.  If the interleaved launch takes ~max(M, S) instead of ~M + S, weight-gradient GEMMs could hide behind the
// BatchNorm backward passes in a role-interleaved launch (two streams do not achieve that: tools/overlap_probe.py).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float vf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void role_mfma(float* out, const float* in, int iters, int bid) {
  constexpr int TM = 2, TN = 2, BM = 128, BN = 128, BK = 16;
  __shared__ __attribute__((aligned(16))) float As[2][BK][BM + 8], Bs[2][BK][BN + 8];
  for (int i = threadIdx.x; i < 2 * BK * (BM + 8); i += 256) (&As[0][0][0])[i] = in[i & 1023];
  for (int i = threadIdx.x; i < 2 * BK * (BN + 8); i += 256) (&Bs[0][0][0])[i] = in[(i * 7) & 1023];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
  const int wm0 = (wave >> 1) * TM * 32, wn0 = (wave & 1) * TN * 32;
  typedef float vA __attribute__((ext_vector_type(TM)));
  typedef float vB __attribute__((ext_vector_type(TN)));
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const vA a = *reinterpret_cast<const vA*>(&As[buf][kk + lh][wm0 + l31 * TM]);
      const vB b = *reinterpret_cast<const vB*>(&Bs[buf][kk + lh][wn0 + l31 * TN]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[(long)bid * 256 + threadIdx.x] = s;
}

// block `sb` of `nsb` streams its contiguous share of n4 float4 elements: y = x * a + d * b (+ a little VALU)
__device__ __forceinline__ void role_stream(const vf4* __restrict__ x, const vf4* __restrict__ d, vf4* __restrict__ y,
                                            long n4, int sb, int nsb) {
  const long per = (n4 + nsb - 1) / nsb;
  const long beg = (long)sb * per, end = beg + per < n4 ? beg + per : n4;
  for (long i = beg + threadIdx.x; i < end; i += 256) {
    const vf4 xv = __builtin_nontemporal_load(x + i), dv = __builtin_nontemporal_load(d + i);
    vf4 o;
    for (int k = 0; k < 4; ++k) {
      const float u = xv[k] * 1.25f + 0.5f;
      o[k] = dv[k] * (u * u * 0.1f + u) - 0.25f * xv[k];
    }
    __builtin_nontemporal_store(o, y + i);
  }
}

// mode 0: every block role M;  1: every block role S;  2: block b -> role (b & 1) ? S : M
__global__ __launch_bounds__(256, 4) void fused(float* out, const float* in, int iters, const vf4* x, const vf4* d, vf4* y,
                                               long n4, int mode, int nm, int ns) {
  const int b = blockIdx.x;
  if (mode == 0) role_mfma(out, in, iters, b);
  else if (mode == 1) role_stream(x, d, y, n4, b, ns);
  else if (b & 1) { if ((b >> 1) < ns) role_stream(x, d, y, n4, b >> 1, ns); }
  else if ((b >> 1) < nm) role_mfma(out, in, iters, b >> 1);
}

int main() {
  float *in, *out;
  const long n4 = (long)604 * 1000 * 1000 / 4;   // 604 M floats per tensor: the 128 ch @ 384x384, B = 32 tensor
  vf4 *x, *d, *y;
  hipMalloc(&in, 1024 * 4); hipMalloc(&out, (size_t)8192 * 256 * 4);
  hipMalloc(&x, n4 * 16); hipMalloc(&d, n4 * 16); hipMalloc(&y, n4 * 16);
  hipMemset(x, 0, n4 * 16); hipMemset(d, 0, n4 * 16);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](auto fn) { float best = 1e9; for (int r = 0; r < 4; ++r) { hipEventRecord(e0); fn(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; } return best; };
  for (int nm : {2048, 4096}) {
    for (int iters : {300, 600}) {
      const int ns = nm;   // as many streaming blocks as MFMA blocks
      const float tm = timeit([&] { hipLaunchKernelGGL(fused, dim3(nm), dim3(256), 0, 0, out, in, iters, x, d, y, n4, 0, nm, ns); });
      const float ts = timeit([&] { hipLaunchKernelGGL(fused, dim3(ns), dim3(256), 0, 0, out, in, iters, x, d, y, n4, 1, nm, ns); });
      const float tf = timeit([&] { hipLaunchKernelGGL(fused, dim3(2 * nm), dim3(256), 0, 0, out, in, iters, x, d, y, n4, 2, nm, ns); });
      const float tb = timeit([&] { hipLaunchKernelGGL(fused, dim3(nm), dim3(256), 0, 0, out, in, iters, x, d, y, n4, 0, nm, ns);
                                    hipLaunchKernelGGL(fused, dim3(ns), dim3(256), 0, 0, out, in, iters, x, d, y, n4, 1, nm, ns); });
      const double fl = (double)nm * 4 * iters * 8 * 4 * (32.0 * 32 * 2 * 2), by = 3.0 * n4 * 16;
      printf("blocks %d+%d iters %d: MFMA alone %.2f ms (%.0f TF)  stream alone %.2f ms (%.2f TB/s)  back to back %.2f ms  "
             "role-interleaved launch %.2f ms  (max %.2f, sum %.2f)\n", nm, ns, iters, tm, fl / tm / 1e9, ts, by / ts / 1e9, tb, tf,
             tm > ts ? tm : ts, tm + ts);
    }
  }
  return 0;
}
