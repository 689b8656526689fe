// forecast.hip — kernels of the Path-B latent linear forecaster (reference
// experiments/v1_experiments/pretrained_ae_linear_sevir/train.py:67,73-83): difference the latent sequence
// against the last input frame and lay it out as the (pixels x features) matrices of the per-latent-pixel
// nn.Linear, the MSE loss, and the inverse layout (+ last frame) for decoding predictions.
#include "common.h"

using namespace wfae;

namespace {

// v (B,T,C,HW) -> X (B*HW, Tin*C) [t*C+c] = v[b,t,c,p] - v[b,Tin-1,c,p];  Y (B*HW, Tout*C) likewise for t >= Tin.
// One block per (b, pixel tile): reads are coalesced along p, writes are staged through LDS so that each row of
// X / Y (Tin*C resp. Tout*C consecutive floats) leaves as one contiguous run.
__global__ __launch_bounds__(256) void diff_pack_kernel(const float* __restrict__ v, float* __restrict__ X,
                                                        float* __restrict__ Y, int T, int Tin, int C, int HW, int PS) {
  extern __shared__ float sm[];            // [PT = 1 << PS pixels][T*C + 1]
  const int F = T * C, FS = F + 1, PT = 1 << PS;
  const int b = blockIdx.y, p0 = blockIdx.x * PT;
  const float* __restrict__ src = v + (long)b * F * HW;
  for (int i = threadIdx.x; i < F * PT; i += blockDim.x) {
    const int f = i >> PS, pl = i & (PT - 1);
    const int p = p0 + pl;
    sm[pl * FS + f] = p < HW ? src[(long)f * HW + p] : 0.f;
  }
  __syncthreads();
  const int Fin = Tin * C, Fout = (T - Tin) * C;
  for (int i = threadIdx.x; i < F * PT; i += blockDim.x) {
    const int pl = i / F, f = i - pl * F;
    const int p = p0 + pl;
    if (p >= HW) continue;
    const int c = f % C;
    const float d = sm[pl * FS + f] - sm[pl * FS + (Tin - 1) * C + c];
    const long row = (long)b * HW + p;
    if (f < Fin) X[row * Fin + f] = d;
    else Y[row * Fout + (f - Fin)] = d;
  }
}

// pred (B*HW, Tout*C) + last input frame of v (B,T,C,HW) -> out (B,Tout,C,HW)
__global__ __launch_bounds__(256) void unpack_add_kernel(const float* __restrict__ pred, const float* __restrict__ v,
                                                         float* __restrict__ out, int T, int Tin, int C, int HW, int PS) {
  extern __shared__ float sm[];            // [PT = 1 << PS pixels][Tout*C + 1]
  const int Fout = (T - Tin) * C, FS = Fout + 1, PT = 1 << PS;
  const int b = blockIdx.y, p0 = blockIdx.x * PT;
  for (int i = threadIdx.x; i < Fout * PT; i += blockDim.x) {
    const int pl = i / Fout, f = i - pl * Fout;
    const int p = p0 + pl;
    sm[pl * FS + f] = p < HW ? pred[((long)b * HW + p) * Fout + f] : 0.f;
  }
  __syncthreads();
  const float* __restrict__ last = v + ((long)b * T + (Tin - 1)) * C * HW;
  float* __restrict__ dst = out + (long)b * Fout * HW;
  for (int i = threadIdx.x; i < Fout * PT; i += blockDim.x) {
    const int f = i >> PS, pl = i & (PT - 1);
    const int p = p0 + pl;
    if (p < HW) dst[(long)f * HW + p] = sm[pl * FS + f] + last[(long)(f % C) * HW + p];
  }
}

__global__ __launch_bounds__(256) void mse_part_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       double* __restrict__ part, long n) {
  __shared__ double sm[16];
  const long stride = (long)gridDim.x * blockDim.x;
  double s = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = a[i] - b[i];
    s += (double)d * d;
  }
  const double r = block_sum(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

__global__ void mse_finalize_kernel(const double* __restrict__ part, int n, double scale, float* __restrict__ out) {
  __shared__ double sm[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
  const double r = block_sum(s, sm);
  if (threadIdx.x == 0) out[0] = (float)(r * scale);
}

__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ g, float w, float* __restrict__ da, long n) {
  const float gv = g[0] * w;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) da[i] = gv * (a[i] - b[i]);
}

}  // namespace

extern "C" {

int wfae_latent_diff_pack(const float* v, float* X, float* Y, int B, int T, int Tin, int C, int HW,
                          wfae_stream_t stream) {
  WFAE_REQUIRE(v && X && Y, WFAE_ERR_NULL_POINTER, "latent_diff_pack: null pointer");
  WFAE_REQUIRE(B > 0 && B <= 65535 && T > 1 && Tin >= 1 && Tin < T && C > 0 && HW > 0, WFAE_ERR_BAD_SHAPE,
               "latent_diff_pack: bad shape");
  int PS = 6;  // pixels per block = 1 << PS, as many as fit a 64 KB LDS tile
  while (PS > 0 && ((size_t)(1 << PS) * (T * C + 1) * sizeof(float)) > 64 * 1024) --PS;
  const size_t lds = (size_t)(1 << PS) * (T * C + 1) * sizeof(float);
  WFAE_REQUIRE(lds <= 64 * 1024, WFAE_ERR_UNSUPPORTED, "latent_diff_pack: T*C = %d too large for the LDS tile", T * C);
  hipLaunchKernelGGL(diff_pack_kernel, dim3(cdiv(HW, 1 << PS), B), dim3(256), lds, (hipStream_t)stream, v, X, Y, T, Tin, C,
                     HW, PS);
  return check_launch("latent_diff_pack");
}

int wfae_latent_unpack_add(const float* pred, const float* v, float* out, int B, int T, int Tin, int C, int HW,
                           wfae_stream_t stream) {
  WFAE_REQUIRE(pred && v && out, WFAE_ERR_NULL_POINTER, "latent_unpack_add: null pointer");
  WFAE_REQUIRE(B > 0 && B <= 65535 && T > 1 && Tin >= 1 && Tin < T && C > 0 && HW > 0, WFAE_ERR_BAD_SHAPE,
               "latent_unpack_add: bad shape");
  const int Fout = (T - Tin) * C;
  int PS = 6;
  while (PS > 0 && ((size_t)(1 << PS) * (Fout + 1) * sizeof(float)) > 64 * 1024) --PS;
  const size_t lds = (size_t)(1 << PS) * (Fout + 1) * sizeof(float);
  WFAE_REQUIRE(lds <= 64 * 1024, WFAE_ERR_UNSUPPORTED, "latent_unpack_add: Tout*C too large for the LDS tile");
  hipLaunchKernelGGL(unpack_add_kernel, dim3(cdiv(HW, 1 << PS), B), dim3(256), lds, (hipStream_t)stream, pred, v, out, T, Tin,
                     C, HW, PS);
  return check_launch("latent_unpack_add");
}

int wfae_mse_fwd(const float* pred, const float* target, float* loss, int64_t n, void* ws, size_t ws_bytes,
                 wfae_stream_t stream) {
  WFAE_REQUIRE(pred && target && loss, WFAE_ERR_NULL_POINTER, "mse_fwd: null pointer");
  WFAE_REQUIRE(n > 0, WFAE_ERR_BAD_SHAPE, "mse_fwd: bad size");
  int blocks = cdiv(n, 256 * 16);
  if (blocks > 1024) blocks = 1024;
  WFAE_REQUIRE(ws && ws_bytes >= (size_t)blocks * sizeof(double), WFAE_ERR_WORKSPACE, "mse_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mse_part_kernel, dim3(blocks), dim3(256), 0, st, pred, target, (double*)ws, (long)n);
  int rc = check_launch("mse_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, blocks, 1.0 / (double)n, loss);
  return check_launch("mse_finalize");
}

int wfae_mse_bwd(const float* pred, const float* target, const float* gloss, float* dpred, int64_t n,
                 wfae_stream_t stream) {
  WFAE_REQUIRE(pred && target && gloss && dpred, WFAE_ERR_NULL_POINTER, "mse_bwd: null pointer");
  WFAE_REQUIRE(n > 0, WFAE_ERR_BAD_SHAPE, "mse_bwd: bad size");
  int blocks = cdiv(n, 256 * 4);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, target, gloss,
                     (float)(2.0 / (double)n), dpred, (long)n);
  return check_launch("mse_bwd");
}

}  // extern "C"
