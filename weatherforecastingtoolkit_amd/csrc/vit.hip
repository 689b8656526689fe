// vit.hip — pieces of the Path-B token autoencoder AE_ViT_2048 (reference pipeline/models/ae_vit.py:84-162)
// that the conv path does not have: 16x16 patch (un)folding around the patch-embedding GEMMs, the broadcast
// add of the positional tokens, and the single-query cross attention of GlobalCrossEncode (:4-42).
// The transformer blocks reuse layernorm / mha / linear kernels (transformer.hip, gemm.hip).
#include "common.h"

using namespace wfae;

namespace {

// x (B, C, Hp*P, Wp*P) <-> rows (B*Hp*Wp, C*P*P): row = (b, py, px), column = (c, ky, kx)   (Conv2d(k=P, s=P) im2col,
// which for non-overlapping patches is a pure permutation).  dir 0: image -> rows, dir 1: rows -> image (+ bias[c]).
__global__ __launch_bounds__(256) void patch_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                    const float* __restrict__ bias, int C, int Hp, int Wp, int P,
                                                    int dir, long total) {
  const long stride = (long)gridDim.x * blockDim.x;
  const int W = Wp * P, H = Hp * P;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    // i indexes the image (b, c, y, x): consecutive threads -> consecutive x
    const int x = (int)(i % W);
    long r = i / W;
    const int y = (int)(r % H);
    r /= H;
    const int c = (int)(r % C);
    const long b = r / C;
    const int py = y / P, ky = y - py * P, px = x / P, kx = x - px * P;
    const long row = (b * Hp + py) * Wp + px;
    const long j = row * ((long)C * P * P) + ((long)c * P + ky) * P + kx;
    if (dir == 0) dst[j] = src[i];
    else dst[i] = src[j] + (bias ? bias[c] : 0.f);
  }
}

// out[o][i] = x[o][i] + p[i]
// dst[r][dst_off + c] = src[(r / row_div) * src_ld + src_off + c]  (c < cols); with zero_fill the other columns of each
// dst row are zeroed.  One strided row copy serves the layout-only steps of AE_ViT_2048 (pipeline/models/ae_vit.py):
// a (1,F) parameter expanded to B rows (src_ld = 0), a column slice and its zero-padded gradient, a row repeated l times.
__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows,
                                                        int cols, long src_ld, int src_off, int row_div, int dst_ld,
                                                        int dst_off, int zero_fill, long total) {
  const long stride = (long)gridDim.x * blockDim.x;
  const int width = zero_fill ? dst_ld : cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long r = i / width;
    const int c = (int)(i - r * width);
    if (zero_fill) {
      const int cs = c - dst_off;
      dst[r * dst_ld + c] = (cs >= 0 && cs < cols) ? src[(r / row_div) * src_ld + src_off + cs] : 0.f;
    } else {
      dst[r * dst_ld + dst_off + c] = src[(r / row_div) * src_ld + src_off + c];
    }
  }
}

__global__ __launch_bounds__(256) void add_bcast_kernel(const float* __restrict__ x, const float* __restrict__ p,
                                                        float* __restrict__ out, long inner, long total) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) out[i] = x[i] + p[i % inner];
}

// Single-query attention per (b, head): q (B, H*D), kv rows (b, l) with columns [k | v] each H*D wide,
// out (B, H*D), probs (B, H, L).  One block of 256 threads per (b, h); L <= 64, D <= 256.
__global__ __launch_bounds__(256) void sq_attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                          float* __restrict__ out, float* __restrict__ probs, int L,
                                                          int H, int D, float scale) {
  __shared__ float sc[64];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int E = H * D, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const float* qr = q + (long)b * E + h * D;
  for (int l = wv; l < L; l += 4) {
    const float* kr = kv + ((long)b * L + l) * 2 * E + h * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s = fmaf(qr[d], kr[d], s);
    s = wave_sum(s);
    if (lane == 0) sc[l] = s * scale;
  }
  __syncthreads();
  if (wv == 0) {
    const float v = lane < L ? sc[lane] : -INFINITY;
    const float mx = wave_max(v);
    const float e = lane < L ? expf(v - mx) : 0.f;
    const float den = wave_sum(e);
    if (lane < L) {
      sc[lane] = e / den;
      probs[((long)b * H + h) * L + lane] = e / den;
    }
  }
  __syncthreads();
  for (int d = t; d < D; d += 256) {
    float o = 0.f;
    for (int l = 0; l < L; ++l) o = fmaf(sc[l], kv[((long)b * L + l) * 2 * E + E + h * D + d], o);
    out[(long)b * E + h * D + d] = o;
  }
}

__global__ __launch_bounds__(256) void sq_attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                          const float* __restrict__ probs, const float* __restrict__ dout,
                                                          float* __restrict__ dq, float* __restrict__ dkv, int L, int H,
                                                          int D, float scale) {
  __shared__ float ds[64];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int E = H * D, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const float* pr = probs + ((long)b * H + h) * L;
  const float* dor = dout + (long)b * E + h * D;
  // dp_l = dout . v_l
  for (int l = wv; l < L; l += 4) {
    const float* vr = kv + ((long)b * L + l) * 2 * E + E + h * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s = fmaf(dor[d], vr[d], s);
    s = wave_sum(s);
    if (lane == 0) ds[l] = s;
  }
  __syncthreads();
  if (wv == 0) {
    const float p = lane < L ? pr[lane] : 0.f;
    const float dp = lane < L ? ds[lane] : 0.f;
    const float dot = wave_sum(p * dp);
    if (lane < L) ds[lane] = p * (dp - dot);   // dS
  }
  __syncthreads();
  const float* qr = q + (long)b * E + h * D;
  for (int d = t; d < D; d += 256) {
    float a = 0.f;
    const float qd = qr[d], dod = dor[d];
    for (int l = 0; l < L; ++l) {
      const long base = ((long)b * L + l) * 2 * E + h * D + d;
      a = fmaf(ds[l], kv[base], a);
      dkv[base] = ds[l] * qd * scale;       // dk
      dkv[base + E] = pr[l] * dod;          // dv
    }
    dq[(long)b * E + h * D + d] = a * scale;
  }
}

// out[a][b] = sum_m x[a][m][b]
__global__ __launch_bounds__(256) void sum_mid_kernel(const float* __restrict__ x, float* __restrict__ out, int M,
                                                      long Bn, long total) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long a = i / Bn, b = i - a * Bn;
    const float* __restrict__ src = x + a * M * Bn + b;
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += src[(long)m * Bn];
    out[i] = s;
  }
}

inline int grid1(long n) {
  long b = (n + 255) / 256;
  return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int wfae_patchify(const float* img, float* rows, int B, int C, int Hp, int Wp, int P, wfae_stream_t stream) {
  WFAE_REQUIRE(img && rows, WFAE_ERR_NULL_POINTER, "patchify: null pointer");
  WFAE_REQUIRE(B > 0 && C > 0 && Hp > 0 && Wp > 0 && P > 0, WFAE_ERR_BAD_SHAPE, "patchify: bad shape");
  const long total = (long)B * C * Hp * P * Wp * P;
  hipLaunchKernelGGL(patch_kernel, dim3(grid1(total)), dim3(256), 0, (hipStream_t)stream, img, rows, (const float*)nullptr, C,
                     Hp, Wp, P, 0, total);
  return check_launch("patchify");
}

int wfae_unpatchify(const float* rows, const float* bias, float* img, int B, int C, int Hp, int Wp, int P,
                    wfae_stream_t stream) {
  WFAE_REQUIRE(rows && img, WFAE_ERR_NULL_POINTER, "unpatchify: null pointer");
  WFAE_REQUIRE(B > 0 && C > 0 && Hp > 0 && Wp > 0 && P > 0, WFAE_ERR_BAD_SHAPE, "unpatchify: bad shape");
  const long total = (long)B * C * Hp * P * Wp * P;
  hipLaunchKernelGGL(patch_kernel, dim3(grid1(total)), dim3(256), 0, (hipStream_t)stream, rows, img, bias, C, Hp, Wp, P, 1, total);
  return check_launch("unpatchify");
}

int wfae_copy_rows(const float* src, float* dst, int64_t rows, int cols, int64_t src_ld, int src_off, int row_div,
                   int dst_ld, int dst_off, int zero_fill, wfae_stream_t stream) {
  WFAE_REQUIRE(src && dst, WFAE_ERR_NULL_POINTER, "copy_rows: null pointer");
  WFAE_REQUIRE(rows > 0 && cols > 0 && row_div > 0 && src_ld >= 0 && src_off >= 0 && dst_off >= 0 && dst_off + cols <= dst_ld,
               WFAE_ERR_BAD_SHAPE, "copy_rows: bad shape");
  const long total = (long)rows * (zero_fill ? dst_ld : cols);
  hipLaunchKernelGGL(copy_rows_kernel, dim3(grid1(total)), dim3(256), 0, (hipStream_t)stream, src, dst, (long)rows, cols,
                     (long)src_ld, src_off, row_div, dst_ld, dst_off, zero_fill, total);
  return check_launch("copy_rows");
}

int wfae_add_bcast(const float* x, const float* p, float* out, int64_t outer, int64_t inner, wfae_stream_t stream) {
  WFAE_REQUIRE(x && p && out, WFAE_ERR_NULL_POINTER, "add_bcast: null pointer");
  WFAE_REQUIRE(outer > 0 && inner > 0, WFAE_ERR_BAD_SHAPE, "add_bcast: bad shape");
  hipLaunchKernelGGL(add_bcast_kernel, dim3(grid1(outer * inner)), dim3(256), 0, (hipStream_t)stream, x, p, out, (long)inner,
                     (long)(outer * inner));
  return check_launch("add_bcast");
}

int wfae_sq_attn_fwd(const float* q, const float* kv, float* out, float* probs, int B, int L, int H, int D,
                     wfae_stream_t stream) {
  WFAE_REQUIRE(q && kv && out && probs, WFAE_ERR_NULL_POINTER, "sq_attn_fwd: null pointer");
  WFAE_REQUIRE(B > 0 && L > 0 && L <= 64 && H > 0 && D > 0, WFAE_ERR_BAD_SHAPE, "sq_attn_fwd: needs 1 <= L <= 64");
  hipLaunchKernelGGL(sq_attn_fwd_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, q, kv, out, probs, L, H, D,
                     1.0f / sqrtf((float)D));
  return check_launch("sq_attn_fwd");
}

int wfae_sq_attn_bwd(const float* q, const float* kv, const float* probs, const float* dout, float* dq, float* dkv, int B,
                     int L, int H, int D, wfae_stream_t stream) {
  WFAE_REQUIRE(q && kv && probs && dout && dq && dkv, WFAE_ERR_NULL_POINTER, "sq_attn_bwd: null pointer");
  WFAE_REQUIRE(B > 0 && L > 0 && L <= 64 && H > 0 && D > 0, WFAE_ERR_BAD_SHAPE, "sq_attn_bwd: needs 1 <= L <= 64");
  hipLaunchKernelGGL(sq_attn_bwd_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, q, kv, probs, dout, dq, dkv, L, H, D,
                     1.0f / sqrtf((float)D));
  return check_launch("sq_attn_bwd");
}

int wfae_sum_mid(const float* x, float* out, int64_t A, int M, int64_t Bn, wfae_stream_t stream) {
  WFAE_REQUIRE(x && out, WFAE_ERR_NULL_POINTER, "sum_mid: null pointer");
  WFAE_REQUIRE(A > 0 && M > 0 && Bn > 0, WFAE_ERR_BAD_SHAPE, "sum_mid: bad shape");
  hipLaunchKernelGGL(sum_mid_kernel, dim3(grid1(A * Bn)), dim3(256), 0, (hipStream_t)stream, x, out, M, (long)Bn,
                     (long)(A * Bn));
  return check_launch("sum_mid");
}

}  // extern "C"
