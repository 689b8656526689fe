// ssim.hip — SSIM (11-tap Gaussian sigma 1.5, valid window) forward/backward and
// PSNR, the metric/loss kernels of experiments/ae_v2/train.py:57-63 and
// pipeline/metrics.py:71-93.  Separable Gaussian evaluated from an LDS tile;
// per-block sums are combined with wavefront shuffles in fp64.
// Single-channel images (NB,1,H,W): the reference triples the channel before
// calling ssim, which does not change the value.
#include "common.h"

using namespace wfae;

namespace {

constexpr int WIN = 11, HALF = 5;
constexpr int TS = 16;            // output tile
constexpr int TI = TS + WIN - 1;  // 26 input rows/cols
constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;

struct Gauss { float g[WIN]; };

inline Gauss make_gauss() {
  Gauss k;
  double s = 0.0, v[WIN];
  for (int i = 0; i < WIN; ++i) { const double c = i - HALF; v[i] = exp(-(c * c) / (2.0 * 1.5 * 1.5)); s += v[i]; }
  // normalise in fp32 like the python packages do (g / g.sum() on float tensors)
  float vf[WIN], sf = 0.f;
  for (int i = 0; i < WIN; ++i) { vf[i] = (float)v[i]; sf += vf[i]; }
  for (int i = 0; i < WIN; ++i) k.g[i] = vf[i] / sf;
  (void)s;
  return k;
}

__device__ __forceinline__ float clamp01f(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

// Computes, for one 16x16 tile of the valid map, the five filtered moments and
// either the block's SSIM sum (MODE 0) or the three gradient maps a,b,c (MODE 1).
template <int MODE>
__global__ __launch_bounds__(256) void ssim_tile_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        Gauss k, int H, int W, int clamp, double* __restrict__ part,
                                                        float* __restrict__ maps) {
  __shared__ float xs[TI][TI + 1], ys[TI][TI + 1];
  __shared__ float hx[TI][TS], hy[TI][TS], hxx[TI][TS], hyy[TI][TS], hxy[TI][TS];
  __shared__ double sm[16];
  const int Ho = H - WIN + 1, Wo = W - WIN + 1;
  const int tiles_x = (Wo + TS - 1) / TS;
  const int n = blockIdx.y;
  const int oy0 = (blockIdx.x / tiles_x) * TS, ox0 = (blockIdx.x % tiles_x) * TS;
  const float* xp = x + (long)n * H * W;
  const float* yp = y + (long)n * H * W;
  const int t = threadIdx.x;
  for (int idx = t; idx < TI * TI; idx += 256) {
    const int r = idx / TI, c = idx - r * TI;
    const int iy = oy0 + r, ix = ox0 + c;
    float xv = 0.f, yv = 0.f;
    if (iy < H && ix < W) { xv = xp[(long)iy * W + ix]; yv = yp[(long)iy * W + ix]; }
    if (clamp) { xv = clamp01f(xv); yv = clamp01f(yv); }
    xs[r][c] = xv; ys[r][c] = yv;
  }
  __syncthreads();
  // horizontal pass: TI rows x TS cols
  for (int idx = t; idx < TI * TS; idx += 256) {
    const int r = idx / TS, c = idx - r * TS;
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
    for (int j = 0; j < WIN; ++j) {
      const float g = k.g[j], xv = xs[r][c + j], yv = ys[r][c + j];
      sx = fmaf(g, xv, sx); sy = fmaf(g, yv, sy);
      sxx = fmaf(g, xv * xv, sxx); syy = fmaf(g, yv * yv, syy); sxy = fmaf(g, xv * yv, sxy);
    }
    hx[r][c] = sx; hy[r][c] = sy; hxx[r][c] = sxx; hyy[r][c] = syy; hxy[r][c] = sxy;
  }
  __syncthreads();
  const int ty = t >> 4, tx = t & 15;
  float mx = 0.f, my = 0.f, exx = 0.f, eyy = 0.f, exy = 0.f;
#pragma unroll
  for (int j = 0; j < WIN; ++j) {
    const float g = k.g[j];
    mx = fmaf(g, hx[ty + j][tx], mx); my = fmaf(g, hy[ty + j][tx], my);
    exx = fmaf(g, hxx[ty + j][tx], exx); eyy = fmaf(g, hyy[ty + j][tx], eyy); exy = fmaf(g, hxy[ty + j][tx], exy);
  }
  const int oy = oy0 + ty, ox = ox0 + tx;
  const bool valid = oy < Ho && ox < Wo;
  const float sxx = exx - mx * mx, syy = eyy - my * my, sxy = exy - mx * my;
  const float A1 = 2.f * mx * my + C1, A2 = 2.f * sxy + C2;
  const float B1 = mx * mx + my * my + C1, B2 = sxx + syy + C2;
  if (MODE == 0) {
    const double v = valid ? (double)((A1 * A2) / (B1 * B2)) : 0.0;
    const double r = block_sum(v, sm);
    if (t == 0) part[(long)n * gridDim.x + blockIdx.x] = r;
  } else if (valid) {
    const float inv = 1.f / (B1 * B2);
    // partial derivatives of s = A1 A2 / (B1 B2) w.r.t. mu_y, E[yy], E[xy]
    const float dmu = ((2.f * mx * A2 - 2.f * mx * A1) * inv) - (A1 * A2 * inv * inv) * (2.f * my * B2 - 2.f * my * B1);
    const float deyy = -(A1 * A2) * inv / B2;
    const float dexy = 2.f * A1 * inv;
    const long o = ((long)n * Ho + oy) * Wo + ox;
    const long plane = (long)gridDim.y * Ho * Wo;
    maps[o] = dmu; maps[plane + o] = deyy; maps[2 * plane + o] = dexy;
  }
}

// dy[q] = g/N * ( (G^T a)(q) + 2 y(q) (G^T b)(q) + x(q) (G^T c)(q) )
__global__ __launch_bounds__(256) void ssim_bwd_gather_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ maps, Gauss k,
                                                              const float* __restrict__ gout, float invN,
                                                              float* __restrict__ dy, int NB, int H, int W) {
  __shared__ float ms[3][TI][TI + 1];
  const int Ho = H - WIN + 1, Wo = W - WIN + 1;
  const int tiles_x = (W + TS - 1) / TS;
  const int n = blockIdx.y;
  const int qy0 = (blockIdx.x / tiles_x) * TS, qx0 = (blockIdx.x % tiles_x) * TS;
  const long plane = (long)NB * Ho * Wo;
  const int t = threadIdx.x;
  // map position p contributes to q = p + d, d in [0,10]: need p in [q-10, q]
  for (int idx = t; idx < 3 * TI * TI; idx += 256) {
    const int m = idx / (TI * TI);
    const int rr = idx - m * TI * TI;
    const int r = rr / TI, c = rr - r * TI;
    const int py = qy0 - (WIN - 1) + r, px = qx0 - (WIN - 1) + c;
    float v = 0.f;
    if (py >= 0 && py < Ho && px >= 0 && px < Wo) v = maps[m * plane + ((long)n * Ho + py) * Wo + px];
    ms[m][r][c] = v;
  }
  __syncthreads();
  const int ty = t >> 4, tx = t & 15;
  const int qy = qy0 + ty, qx = qx0 + tx;
  if (qy >= H || qx >= W) return;
  float sa = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
  for (int dyy = 0; dyy < WIN; ++dyy) {
    float ra = 0.f, rb = 0.f, rc = 0.f;
#pragma unroll
    for (int dxx = 0; dxx < WIN; ++dxx) {
      // p = q - d  -> tile index (q - d) - (q0 - 10) = t + 10 - d
      const float g = k.g[dxx];
      ra = fmaf(g, ms[0][ty + WIN - 1 - dyy][tx + WIN - 1 - dxx], ra);
      rb = fmaf(g, ms[1][ty + WIN - 1 - dyy][tx + WIN - 1 - dxx], rb);
      rc = fmaf(g, ms[2][ty + WIN - 1 - dyy][tx + WIN - 1 - dxx], rc);
    }
    sa = fmaf(k.g[dyy], ra, sa); sb = fmaf(k.g[dyy], rb, sb); sc = fmaf(k.g[dyy], rc, sc);
  }
  const long q = ((long)n * H + qy) * W + qx;
  dy[q] = gout[0] * invN * (sa + 2.f * y[q] * sb + x[q] * sc);
}

__global__ void ssim_finalize_kernel(const double* __restrict__ part, long parts, double invN, float* out) {
  __shared__ double sm[16];
  double s = 0.0;
  for (long i = threadIdx.x; i < parts; i += blockDim.x) s += part[i];
  const double r = block_sum(s, sm);
  if (threadIdx.x == 0) out[0] = (float)(r * invN);
}

// per image: sum (p-t)^2, min t, max t
__global__ __launch_bounds__(256) void psnr_part_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                        int HW, int clamp, double* __restrict__ psum,
                                                        float* __restrict__ pmin, float* __restrict__ pmax) {
  __shared__ double sm[16];
  __shared__ float smin[4], smax[4];
  const int n = blockIdx.y;
  double s = 0.0;
  float lo = INFINITY, hi = -INFINITY;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
    float p = pred[(long)n * HW + i], g = tgt[(long)n * HW + i];
    if (clamp) { p = clamp01f(p); g = clamp01f(g); }
    const float d = p - g;
    s += (double)d * d;
    lo = fminf(lo, g); hi = fmaxf(hi, g);
  }
  lo = wave_min(lo); hi = wave_max(hi);
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
  const double r = block_sum(s, sm);
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) { lo = fminf(lo, smin[w]); hi = fmaxf(hi, smax[w]); }
    lo = fminf(lo, smin[0]); hi = fmaxf(hi, smax[0]);
    const long o = (long)n * gridDim.x + blockIdx.x;
    psum[o] = r; pmin[o] = lo; pmax[o] = hi;
  }
}

__global__ void psnr_finalize_kernel(const double* __restrict__ psum, const float* __restrict__ pmin,
                                     const float* __restrict__ pmax, int NB, int chunks, int HW, float* out) {
  __shared__ double sm[16];
  double tot = 0.0;
  for (int n = threadIdx.x; n < NB; n += blockDim.x) {
    double s = 0.0;
    float lo = INFINITY, hi = -INFINITY;
    for (int c = 0; c < chunks; ++c) {
      s += psum[(long)n * chunks + c];
      lo = fminf(lo, pmin[(long)n * chunks + c]);
      hi = fmaxf(hi, pmax[(long)n * chunks + c]);
    }
    const double mse = s / (double)HW;
    const double range = (double)hi - (double)lo;
    tot += 10.0 * log10(range * range / mse);
  }
  const double r = block_sum(tot, sm);
  if (threadIdx.x == 0) out[0] = (float)(r / (double)NB);
}

}  // namespace

extern "C" {

int wfae_ssim_fwd(const float* x, const float* y, float* out, int NB, int H, int W, int clamp01, void* ws,
                  size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(x && y && out, WFAE_ERR_NULL_POINTER, "ssim_fwd: null pointer");
  WFAE_REQUIRE(NB > 0 && NB <= 65535 && H >= WIN && W >= WIN, WFAE_ERR_BAD_SHAPE, "ssim_fwd: bad shape");
  const int Ho = H - WIN + 1, Wo = W - WIN + 1;
  const int tiles = cdiv(Wo, TS) * cdiv(Ho, TS);
  WFAE_REQUIRE(ws && ws_bytes >= (size_t)NB * tiles * sizeof(double), WFAE_ERR_WORKSPACE, "ssim_fwd: workspace");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((ssim_tile_kernel<0>), dim3(tiles, NB), dim3(256), 0, st, x, y, make_gauss(), H, W, clamp01,
                     (double*)ws, (float*)nullptr);
  int rc = check_launch("ssim_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(ssim_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, (long)NB * tiles,
                     1.0 / ((double)NB * Ho * Wo), out);
  return check_launch("ssim_finalize");
}

int wfae_ssim_bwd(const float* x, const float* y, const float* gout, float* dy, int NB, int H, int W, void* ws,
                  size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(x && y && gout && dy, WFAE_ERR_NULL_POINTER, "ssim_bwd: null pointer");
  WFAE_REQUIRE(NB > 0 && NB <= 65535 && H >= WIN && W >= WIN, WFAE_ERR_BAD_SHAPE, "ssim_bwd: bad shape");
  const int Ho = H - WIN + 1, Wo = W - WIN + 1;
  const size_t need = (size_t)3 * NB * Ho * Wo * sizeof(float);
  WFAE_REQUIRE(ws && ws_bytes >= need, WFAE_ERR_WORKSPACE, "ssim_bwd: workspace %zu < %zu", ws_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  const Gauss k = make_gauss();
  hipLaunchKernelGGL((ssim_tile_kernel<1>), dim3(cdiv(Wo, TS) * cdiv(Ho, TS), NB), dim3(256), 0, st, x, y, k, H, W,
                     0, (double*)nullptr, (float*)ws);
  int rc = check_launch("ssim_bwd_maps");
  if (rc) return rc;
  hipLaunchKernelGGL(ssim_bwd_gather_kernel, dim3(cdiv(W, TS) * cdiv(H, TS), NB), dim3(256), 0, st, x, y,
                     (const float*)ws, k, gout, (float)(1.0 / ((double)NB * Ho * Wo)), dy, NB, H, W);
  return check_launch("ssim_bwd_gather");
}

int wfae_psnr(const float* pred, const float* target, float* out, int NB, int HW, int clamp01, void* ws,
              size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(pred && target && out, WFAE_ERR_NULL_POINTER, "psnr: null pointer");
  WFAE_REQUIRE(NB > 0 && NB <= 65535 && HW > 0, WFAE_ERR_BAD_SHAPE, "psnr: bad shape");
  int chunks = cdiv(HW, 256 * 16);
  if (chunks > 64) chunks = 64;
  const size_t n = (size_t)NB * chunks;
  WFAE_REQUIRE(ws && ws_bytes >= n * (sizeof(double) + 2 * sizeof(float)), WFAE_ERR_WORKSPACE, "psnr: workspace");
  double* psum = (double*)ws;
  float* pmin = (float*)(psum + n);
  float* pmax = pmin + n;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(psnr_part_kernel, dim3(chunks, NB), dim3(256), 0, st, pred, target, HW, clamp01, psum, pmin, pmax);
  int rc = check_launch("psnr_part");
  if (rc) return rc;
  hipLaunchKernelGGL(psnr_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)psum, (const float*)pmin,
                     (const float*)pmax, NB, chunks, HW, out);
  return check_launch("psnr_finalize");
}

}  // extern "C"
