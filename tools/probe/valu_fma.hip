// VALU fp32 FMA issue-rate probe (gfx950): v_fma_f32 with a VGPR / SGPR multiplier, v_pk_fma_f32 with an SGPR pair, at 1 - 8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_fma valu_fma.hip ; run: ./valu_fma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int NACC = 16, INNER = 64, OUTER = 256;

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ w, float* __restrict__ y, int g) {
  float acc[NACC];
  f2 acc2[NACC / 2];
  for (int i = 0; i < NACC; ++i) acc[i] = (float)i;
  for (int i = 0; i < NACC / 2; ++i) acc2[i] = {(float)i, 1.f};
  float a = (float)threadIdx.x * 1e-3f, b = a + 1.f;
  const float* wc = w + g * 64;
  for (int o = 0; o < OUTER; ++o) {
    float sw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sw[j] = wc[(o & 7) * 8 + j];   // uniform -> s_load
    float vw = y[threadIdx.x & 7] + (float)o;                    // a VGPR multiplier
#pragma unroll
    for (int r = 0; r < INNER / 8; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (MODE == 0) {
#pragma unroll
          for (int i = 0; i < NACC; ++i) acc[i] = fmaf((i & 1) ? a : b, vw, acc[i]);
        } else if constexpr (MODE == 1) {
#pragma unroll
          for (int i = 0; i < NACC; ++i) acc[i] = fmaf((i & 1) ? a : b, sw[j], acc[i]);
        } else {
          const f2 w2 = {sw[j], sw[(j + 1) & 7]};
#pragma unroll
          for (int i = 0; i < NACC / 2; ++i) acc2[i] = __builtin_elementwise_fma((f2){(i & 1) ? a : b, (i & 1) ? a : b}, w2, acc2[i]);
        }
      }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  for (int i = 0; i < NACC / 2; ++i) s += acc2[i].x + acc2[i].y;
  if (s == 123.456f) y[threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, const float* w, float* y, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, w, y, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, w, y, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 10;
  const double fma = (double)blocks * 256 * OUTER * INNER * NACC;
  printf("%-28s waves/SIMD %d  %.3f ms  %.1f TFLOP/s\n", name, waves_per_simd, ms, 2 * fma / ms * 1e-9);
}

int main() {
  float *w, *y;
  hipMalloc(&w, 4096); hipMalloc(&y, 4096);
  hipMemset(w, 0, 4096); hipMemset(y, 0, 4096);
  for (int wps : {1, 2, 4, 8}) {
    run<0>("v_fma_f32 v,v,v", w, y, wps);
    run<1>("v_fma_f32 v,s,v", w, y, wps);
    run<2>("v_pk_fma_f32 v,s[2],v", w, y, wps);
  }
  return 0;
}
