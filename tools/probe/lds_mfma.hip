// Ceiling of the GEMM inner structure: operands re-read from LDS (ds_read_b32, As[k][m] / Bs[k][n] images)
// for every 32x32x2 fp32 MFMA, 64x64 or 128x64 wave tiles, no global traffic.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int TM, int TN, int BAR>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
  constexpr int BM = TM * 64, BN = TN * 64, BK = 16;
  __shared__ float As[2][BK][BM + 4], Bs[2][BK][BN + 4];
  for (int i = threadIdx.x; i < 2 * BK * (BM + 4); i += 256) (&As[0][0][0])[i] = in[i & 1023];
  for (int i = threadIdx.x; i < 2 * BK * (BN + 4); i += 256) (&Bs[0][0][0])[i] = in[(i * 7) & 1023];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
  const int wm0 = (wave >> 1) * TM * 32, wn0 = (wave & 1) * TN * 32;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[buf][kk + lh][wm0 + i * 32 + l31];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[buf][kk + lh][wn0 + j * 32 + l31];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (BAR) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// same loop with PERMUTED LDS images: the TM (TN) values a lane needs for one k are adjacent, so one
// ds_read_b128 / ds_read_b64 fetches them all
template <int TM, int TN>
__global__ __launch_bounds__(256) void kp(float* out, const float* in, int iters) {
  constexpr int BM = TM * 64, BN = TN * 64, BK = 16;
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
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int TM, int TN>
void runp(const char* name, int blocks, float* out, float* in) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kp<TM, TN>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double fl = (double)blocks * 4 * iters * 8 * TM * TN * (32.0 * 32 * 2 * 2);
  printf("%s blocks %d: %.2f ms  %.1f TFLOP/s\n", name, blocks, best, fl / best / 1e9);
}
template <int TM, int TN, int BAR>
void run(const char* name, int blocks, float* out, float* in) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<TM, TN, BAR>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double fl = (double)blocks * 4 * iters * 8 * TM * TN * (32.0 * 32 * 2 * 2);
  printf("%s blocks %d: %.2f ms  %.1f TFLOP/s\n", name, blocks, best, fl / best / 1e9);
}
int main() {
  float *in, *out;
  hipMalloc(&in, 1024 * 4); hipMalloc(&out, (size_t)4096 * 256 * 4);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<2, 2, 0>("wave 64x64, no barrier, 4 blk/CU", 1024, out, in);
  run<2, 2, 1>("wave 64x64, barrier/stage, 4 blk/CU", 1024, out, in);
  run<4, 2, 0>("wave 128x64, no barrier, 2 blk/CU", 512, out, in);
  run<4, 2, 1>("wave 128x64, barrier/stage, 2 blk/CU", 512, out, in);
  run<4, 2, 1>("wave 128x64, barrier/stage, 4 blk/CU-worth", 1024, out, in);
  runp<2, 2>("PERMUTED wave 64x64 (b64+b64), barrier, 4 blk/CU", 1024, out, in);
  runp<4, 2>("PERMUTED wave 128x64 (b128+b64), barrier, 2 blk/CU", 512, out, in);
  runp<4, 4>("PERMUTED wave 128x128 (b128+b128), barrier, 1 blk/CU", 256, out, in);
  return 0;
}
