// c1conv.hip — the decoder's output convolution Conv2d(C, 1, 3, padding=1) (ae_64x8x8_lin.py:84) at full
// resolution: one output channel, C = 128 input channels, 384 x 384 pixels (fallback weight gradient for shapes
// c1_wgrad_mfma declines; the forward runs on dconv_fwd_kernel, which measured 1.3 ms against 2.4 ms for a dedicated
// kernel of this design).  The kernel streams the C-channel
// tensor exactly once (2.4 GB at B = 32): a lane owns one pixel column, walks the rows of its tile with the three
// input rows of the stencil in registers; left / right neighbours are two more (overlapping, cache-resident) loads:
// branch-free, no LDS traffic in the loop.
#include "common.h"

using namespace wfae;

namespace {

constexpr int C1_TH = 16;   // rows per tile
constexpr int C1_TW = 64;   // columns per tile = lanes

// value of row `yy` of plane `src` at this lane's column and its two neighbours: three overlapping coalesced loads
// from clamped addresses (the neighbours hit the lines the centre load brought in), zeros outside the image
__device__ __forceinline__ void load_row3(const float* __restrict__ src, int yy, int H, int W, int col, int lane,
                                          float& l, float& c, float& r) {
  (void)lane;
  const bool rok = yy >= 0 && yy < H;
  const float* __restrict__ row = src + (long)(rok ? yy : 0) * W;
  const int cc = col < W ? col : W - 1;
  const float vl = row[cc >= 1 ? cc - 1 : 0], vc = row[cc], vr = row[cc + 1 < W ? cc + 1 : W - 1];
  c = (rok && col < W) ? vc : 0.f;
  l = (rok && col >= 1 && col - 1 < W) ? vl : 0.f;
  r = (rok && col + 1 < W) ? vr : 0.f;
}

// dw[c][tap] partials: part[tile][c][9] = sum over the tile's pixels of dy[n,0,oy,ox] x[n,c,oy+ky-1,ox+kx-1];
// grid (tiles, C / 8, NB): a block covers 8 channels, each wave two of them
__global__ __launch_bounds__(256) void c1conv3_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            float* __restrict__ part, int C, int H, int W, int tiles_x,
                                                            int tiles_per_img) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.z, c0 = blockIdx.y * 8 + wave * 2;
  const int oy0 = (blockIdx.x / tiles_x) * C1_TH, col = (blockIdx.x % tiles_x) * C1_TW + lane;
  float acc[2][9];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[j][k] = 0.f;
  const float* __restrict__ dyp = dy + (long)n * H * W;
  float d[C1_TH];
#pragma unroll
  for (int r = 0; r < C1_TH; ++r) d[r] = (oy0 + r < H && col < W) ? dyp[(long)(oy0 + r) * W + col] : 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (c0 + j >= C) continue;
    const float* __restrict__ src = x + ((long)n * C + c0 + j) * H * W;
#pragma unroll
    for (int i = 0; i < C1_TH + 2; ++i) {
      float l, m, r;
      load_row3(src, oy0 - 1 + i, H, W, col, lane, l, m, r);
      if (i < C1_TH) {          // stencil row ky = 0 of output row i
        acc[j][0] = fmaf(d[i], l, acc[j][0]); acc[j][1] = fmaf(d[i], m, acc[j][1]); acc[j][2] = fmaf(d[i], r, acc[j][2]);
      }
      if (i >= 1 && i - 1 < C1_TH) {
        acc[j][3] = fmaf(d[i - 1], l, acc[j][3]); acc[j][4] = fmaf(d[i - 1], m, acc[j][4]); acc[j][5] = fmaf(d[i - 1], r, acc[j][5]);
      }
      if (i >= 2) {
        acc[j][6] = fmaf(d[i - 2], l, acc[j][6]); acc[j][7] = fmaf(d[i - 2], m, acc[j][7]); acc[j][8] = fmaf(d[i - 2], r, acc[j][8]);
      }
    }
  }
  const long tile = (long)n * tiles_per_img + blockIdx.x;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float v = wave_sum(acc[j][k]);
      if (lane == 0 && c0 + j < C) part[(tile * C + c0 + j) * 9 + k] = v;
    }
}

}  // namespace

namespace wfae {


// returns WFAE_ERR_WORKSPACE if the partials do not fit (the caller falls back to the generic kernel)
int c1conv3_wgrad(const float* dy, const float* x, float* dw, int NB, int C, int H, int W, int accumulate, void* ws,
                  size_t ws_bytes, hipStream_t st) {
  const int tiles_x = cdiv(W, C1_TW), tiles_y = cdiv(H, C1_TH);
  const long parts = (long)NB * tiles_x * tiles_y;
  const size_t need = (size_t)parts * C * 9 * sizeof(float);
  if (!ws || need > ws_bytes) return WFAE_ERR_WORKSPACE;
  hipLaunchKernelGGL(c1conv3_wgrad_kernel, dim3(tiles_x * tiles_y, cdiv(C, 8), NB), dim3(256), 0, st, dy, x, (float*)ws, C, H,
                     W, tiles_x, tiles_x * tiles_y);
  int rc = check_launch("c1conv3_wgrad");
  if (rc) return rc;
  return slab_reduce((const float*)ws, dw, nullptr, (long)C * 9, 1, (int)parts, accumulate, st);
}

}  // namespace wfae
