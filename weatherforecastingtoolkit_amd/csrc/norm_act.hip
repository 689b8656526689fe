// norm_act.hip — bandwidth-bound kernels of the hot path: BatchNorm2d statistics
// (wavefront-shuffle reductions, fp64 accumulation), fused BN-apply + exact GELU,
// their backward, sigmoid + L1 loss, element-wise helpers and channel reductions.
// All kernels use 16-byte loads when HW % 4 == 0 (always true for the model).
#include <stdlib.h>
#include "common.h"

using namespace wfae;

namespace {

constexpr int RT = 256;  // reduction block size
constexpr float kLeaky = 0.2f;  // nn.LeakyReLU(0.2) of the PatchGAN discriminator (losses/model.py:125-141)

// ACT: 0 identity, 1 exact GELU, 2 LeakyReLU(0.2)
template <int ACT>
__device__ __forceinline__ float act_f(float u) {
  if (ACT == 1) return gelu_f(u);
  if (ACT == 2) return u > 0.f ? u : kLeaky * u;
  return u;
}
template <int ACT>
__device__ __forceinline__ float act_grad_f(float u) {
  if (ACT == 1) return gelu_grad_f(u);
  if (ACT == 2) return u > 0.f ? 1.f : kLeaky;
  return 1.f;
}

// ------------------------------------------------------------- reductions
// Generic per-channel reduction over x[outer][C][inner]:
// block (c, s) reduces the s-th slice of the (outer*inner) index space.
struct RedGeom {
  int outer, C, inner, splits;
  long per_split;  // elements per split (multiple of 8: whole vector accesses of fp32 and bf16 tensors)
};

inline RedGeom red_geom(int outer, int C, int inner, long max_parts_per_c, long target_blocks = 2048) {
  RedGeom g;
  g.outer = outer; g.C = C; g.inner = inner;
  const long total = (long)outer * inner;
  long splits = (target_blocks + C - 1) / C;
  if (splits > max_parts_per_c) splits = max_parts_per_c;
  const long min_chunk = 4096;
  if (splits > (total + min_chunk - 1) / min_chunk) splits = (total + min_chunk - 1) / min_chunk;
  if (splits < 1) splits = 1;
  long per = (total + splits - 1) / splits;
  per = (per + 7) / 8 * 8;   // whole 16-byte accesses of either element type
  g.per_split = per;
  g.splits = (int)((total + per - 1) / per);
  return g;
}

// mode 0: (sum x, sum x^2)          stats
// mode 1: (sum x, 0)                plain sum
template <int MODE, typename T = float>
__global__ __launch_bounds__(RT) void chan_reduce_kernel(const T* __restrict__ x, double* __restrict__ part,
                                                         RedGeom g, int vec) {
  constexpr int W = ElemW<T>::W;
  __shared__ double sm[16];
  const int c = blockIdx.x, s = blockIdx.y;
  const long total = (long)g.outer * g.inner;
  const long beg = (long)s * g.per_split;
  long end = beg + g.per_split;
  if (end > total) end = total;
  double s1 = 0.0, s2 = 0.0;
  if (vec) {
    // (outer, inner) cursor advanced incrementally: no 64-bit division in the loop; the values of a 16-byte access
    // are combined four at a time in fp32 before they enter the fp64 accumulators (the kernels were VALU-, not HBM-bound)
    long i = beg + (long)threadIdx.x * W;
    long o = i / g.inner;
    long in = i - o * g.inner;
    for (; i < end; i += RT * W) {
      float v[W];
      ldv(x + (o * g.C + c) * g.inner + in, v);
#pragma unroll
      for (int q = 0; q < W; q += 4) {
        s1 += (double)((v[q] + v[q + 1]) + (v[q + 2] + v[q + 3]));
        if (MODE == 0) s2 += (double)(fmaf(v[q], v[q], v[q + 1] * v[q + 1]) + fmaf(v[q + 2], v[q + 2], v[q + 3] * v[q + 3]));
      }
      in += RT * W;
      while (in >= g.inner) { in -= g.inner; ++o; }
    }
  } else {
    for (long i = beg + threadIdx.x; i < end; i += RT) {
      const long o = i / g.inner;
      const long in = i - o * g.inner;
      const float v = ld1(x + (o * g.C + c) * g.inner + in);
      s1 += v;
      if (MODE == 0) s2 += (double)v * v;
    }
  }
  const double r1 = block_sum(s1, sm);
  double r2 = 0.0;
  if (MODE == 0) r2 = block_sum(s2, sm);
  if (threadIdx.x == 0) {
    part[((long)c * g.splits + s) * 2 + 0] = r1;
    part[((long)c * g.splits + s) * 2 + 1] = r2;
  }
}

// partial rows of a producer epilogue (wfae_conv1x1_fwd_stats): sum[P][C], sq[P][C] -> fp64 partials in the layout
// chan_reduce_kernel writes, part[(c * splits + s) * 2 + {0,1}]; block (c / 64, s) adds the rows p = s, s + splits, ...
__global__ __launch_bounds__(256) void stat_rows_reduce_kernel(const double* __restrict__ sum, const double* __restrict__ sq,
                                                               double* __restrict__ part, int P, int C, int splits) {
  __shared__ double sm[2][4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, s = blockIdx.y;
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (long p = s + (long)rg * splits; p < P; p += 4l * splits) {
      s1 += sum[p * C + c];
      s2 += sq[p * C + c];
    }
  sm[0][rg][cl] = s1;
  sm[1][rg][cl] = s2;
  __syncthreads();
  if (rg == 0 && c < C) {
    part[((long)c * splits + s) * 2 + 0] = (sm[0][0][cl] + sm[0][1][cl]) + (sm[0][2][cl] + sm[0][3][cl]);
    part[((long)c * splits + s) * 2 + 1] = (sm[1][0][cl] + sm[1][1][cl]) + (sm[1][2][cl] + sm[1][3][cl]);
  }
}

// column sums of a row-major matrix x[rows][C] (bias gradient of nn.Linear: inner == 1 in wfae_reduce_sum), first
// stage: block (column block, row slice) = 64 columns x 16 row groups over `rps` rows, coalesced reads along the
// columns, fp64 partials in the layout sum_finalize_kernel reads.  (chan_reduce_kernel walks a column with stride C:
// 22 us per call on 2048 x 512 problems.)
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ x, int rows, int C, int rps,
                                                      double* __restrict__ part, int splits) {
  __shared__ double sm[16][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, s = blockIdx.y;
  const long r1 = min((long)rows, (long)(s + 1) * rps);
  double a = 0.0;
  if (c < C)
    for (long r = (long)s * rps + rg; r < r1; r += 16) a += (double)x[r * C + c];
  sm[rg][cl] = a;
  __syncthreads();
  if (rg == 0 && c < C) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += sm[r][cl];
    part[((long)c * splits + s) * 2] = t;
  }
}

__global__ void bn_finalize_kernel(const double* __restrict__ part, int splits, long count, int C,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float* __restrict__ save_mean,
                                   float* __restrict__ save_invstd, float* __restrict__ scale,
                                   float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int s = 0; s < splits; ++s) {
    s1 += part[((long)c * splits + s) * 2 + 0];
    s2 += part[((long)c * splits + s) * 2 + 1];
  }
  const double mean = s1 / (double)count;
  double var = s2 / (double)count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float meanf = (float)mean;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  save_mean[c] = meanf;
  save_invstd[c] = invstd;
  const float a = gamma[c] * invstd;
  scale[c] = a;
  shift[c] = beta[c] - meanf * a;
  if (running_mean) {
    const double unb = count > 1 ? var * ((double)count / (double)(count - 1)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

__global__ void bn_fold_eval_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                    float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                    float* __restrict__ scale, float* __restrict__ shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(rv[c] + eps);
  save_mean[c] = rm[c];
  save_invstd[c] = invstd;
  const float a = gamma[c] * invstd;
  scale[c] = a;
  shift[c] = beta[c] - rm[c] * a;
}

__global__ void sum_finalize_kernel(const double* __restrict__ part, int splits, int C, float* __restrict__ out,
                                    int beta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0;
  for (int s = 0; s < splits; ++s) s1 += part[((long)c * splits + s) * 2];
  float v = (float)s1;
  if (beta) v += out[c];
  out[c] = v;
}

// ------------------------------------------------------- BN apply (+GELU)
// grid.x = NB*C planes, grid.y = chunks of the HW plane
// STATS: the block also reduces the sum / sum of squares of the values it WRITES (fp64 accumulators fed with fp32 sums of
// four, exactly like chan_reduce_kernel) and leaves them in part[(split * C + c) * 2 + {0,1}], split = n * gridDim.y +
// blockIdx.y — the BatchNorm that reads y next (the first Bottleneck after a Down/Up unit) needs no statistics pass
template <int ACT, bool STATS = false, typename T = float>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ x,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, T* __restrict__ y,
                                                         int C, int HW, int vec, int nt, double* __restrict__ part = nullptr) {
  constexpr int W = ElemW<T>::W;
  __shared__ double sm[16];
  const int plane = blockIdx.x;
  const int c = plane % C;
  const float a = scale[c], b = shift[c];
  const T* xp = x + (long)plane * HW;
  T* yp = y + (long)plane * HW;
  double s1 = 0.0, s2 = 0.0;
  if (vec) {
    const int nv = HW / W;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < nv; i += gridDim.y * blockDim.x) {
      float v[W], o[W];
      ldv(xp + (long)i * W, v);
#pragma unroll
      for (int q = 0; q < W; ++q) o[q] = act_f<ACT>(fmaf(v[q], a, b));
      if (nt) stv<true>(yp + (long)i * W, o);
      else stv(yp + (long)i * W, o);
      if constexpr (STATS) {
        // the sums are those of the values a reader of y will see: rounded to the storage type first
        if constexpr (sizeof(T) == 2) {
#pragma unroll
          for (int q = 0; q < W; q += 2) {
            const unsigned pk = pack_bf16(o[q], o[q + 1]);
            o[q] = bf16_lo(pk);
            o[q + 1] = bf16_hi(pk);
          }
        }
#pragma unroll
        for (int q = 0; q < W; q += 4) {
          s1 += (double)((o[q] + o[q + 1]) + (o[q + 2] + o[q + 3]));
          s2 += (double)(fmaf(o[q], o[q], o[q + 1] * o[q + 1]) + fmaf(o[q + 2], o[q + 2], o[q + 3] * o[q + 3]));
        }
      }
    }
  } else {
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < HW; i += gridDim.y * blockDim.x) {
      float o = act_f<ACT>(fmaf(ld1(xp + i), a, b));
      st1(yp + i, o);
      if constexpr (STATS) {
        if constexpr (sizeof(T) == 2) o = ld1(yp + i);
        s1 += o;
        s2 += (double)o * o;
      }
    }
  }
  if constexpr (STATS) {
    const double r1 = block_sum(s1, sm);
    const double r2 = block_sum(s2, sm);
    if (threadIdx.x == 0) {
      const long split = (long)(plane / C) * gridDim.y + blockIdx.y;
      part[(split * C + c) * 2 + 0] = r1;
      part[(split * C + c) * 2 + 1] = r2;
    }
  }
}

// finalize for producer-side partial sums part[(split * C + c) * 2 + {0,1}]: one 64-lane block per channel adds the
// splits in a fixed order (lane-strided, then a wave reduction), then the arithmetic of bn_finalize_kernel
__global__ __launch_bounds__(64) void bn_finalize_parts_kernel(const double* __restrict__ part, int splits, long ss, long cs,
                                                               long count, int C,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float eps, float momentum, float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, float* __restrict__ save_mean,
                                                               float* __restrict__ save_invstd, float* __restrict__ scale,
                                                               float* __restrict__ shift) {
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int s = threadIdx.x; s < splits; s += 64) {
    s1 += part[((long)s * ss + (long)c * cs) * 2 + 0];
    s2 += part[((long)s * ss + (long)c * cs) * 2 + 1];
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (threadIdx.x != 0) return;
  const double mean = s1 / (double)count;
  double var = s2 / (double)count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float meanf = (float)mean;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  save_mean[c] = meanf;
  save_invstd[c] = invstd;
  const float a = gamma[c] * invstd;
  scale[c] = a;
  shift[c] = beta[c] - meanf * a;
  if (running_mean) {
    const double unb = count > 1 ? var * ((double)count / (double)(count - 1)) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

// backward pass 1: per channel  sum dU  and  sum dU * xhat,  dU = dy * act'(u)
template <int ACT, typename T = float, int U = 2>
__global__ __launch_bounds__(RT) void bn_act_bwd_reduce_kernel(const T* __restrict__ dy,
                                                               const T* __restrict__ x,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               double* __restrict__ part, RedGeom g, int vec) {
  constexpr int W = ElemW<T>::W;
  __shared__ double sm[16];
  const int c = blockIdx.x, s = blockIdx.y;
  const long total = (long)g.outer * g.inner;
  const long beg = (long)s * g.per_split;
  long end = beg + g.per_split;
  if (end > total) end = total;
  const float a = scale[c], b = shift[c], mu = mean[c], is = invstd[c];
  double s1 = 0.0, s2 = 0.0;
  auto one = [&](float xv, float dv) {
    const float du = dv * act_grad_f<ACT>(fmaf(xv, a, b));
    s1 += du;
    s2 += (double)du * ((xv - mu) * is);
  };
  if (vec) {
    long i = beg + (long)threadIdx.x * W;
    long o = i / g.inner;
    long in = i - o * g.inner;
    // two chunks per iteration, all four loads issued before the arithmetic (the fp64 accumulation chain kept the compiler
    // from overlapping iterations: 5.2 TB/s against 6.1 of the dx pass); sums are added in the same order as one chunk
    // per iteration would, a missing second chunk re-reads the first and is skipped
    auto quad = [&](const float* xv, const float* dv) {
      const float d0 = dv[0] * act_grad_f<ACT>(fmaf(xv[0], a, b)), d1 = dv[1] * act_grad_f<ACT>(fmaf(xv[1], a, b));
      const float d2 = dv[2] * act_grad_f<ACT>(fmaf(xv[2], a, b)), d3 = dv[3] * act_grad_f<ACT>(fmaf(xv[3], a, b));
      s1 += (double)((d0 + d1) + (d2 + d3));
      const float h0 = (xv[0] - mu) * is, h1 = (xv[1] - mu) * is, h2 = (xv[2] - mu) * is, h3 = (xv[3] - mu) * is;
      s2 += (double)(fmaf(d0, h0, d1 * h1) + fmaf(d2, h2, d3 * h3));
    };
    for (; i < end; i += U * RT * W) {
      // U chunks per iteration, all 2 U loads issued before the arithmetic; sums are added in the same order as one chunk per
      // iteration would, a missing chunk re-reads the first and is skipped
      long off[U];
      bool on[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        on[u] = i + (long)u * RT * W < end;
        off[u] = on[u] ? (o * g.C + c) * g.inner + in : off[0];
        in += RT * W;
        while (in >= g.inner) { in -= g.inner; ++o; }
      }
      float xv[U][W], dv[U][W];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ldv(x + off[u], xv[u]);
        ldv(dy + off[u], dv[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (on[u]) {
#pragma unroll
          for (int q = 0; q < W; q += 4) quad(xv[u] + q, dv[u] + q);
        }
      }
    }
  } else {
    for (long i = beg + threadIdx.x; i < end; i += RT) {
      const long o = i / g.inner;
      const long off = (o * g.C + c) * g.inner + (i - o * g.inner);
      one(ld1(x + off), ld1(dy + off));
    }
  }
  const double r1 = block_sum(s1, sm);
  const double r2 = block_sum(s2, sm);
  if (threadIdx.x == 0) {
    part[((long)c * g.splits + s) * 2 + 0] = r1;
    part[((long)c * g.splits + s) * 2 + 1] = r2;
  }
}

// finalize: dbeta = sum dU, dgamma = sum dU*xhat; also leaves the two fp32 sums in coef[2c..]
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ part, int splits, int C,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ coef, int beta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int s = 0; s < splits; ++s) {
    s1 += part[((long)c * splits + s) * 2 + 0];
    s2 += part[((long)c * splits + s) * 2 + 1];
  }
  coef[2 * c + 0] = (float)s1;
  coef[2 * c + 1] = (float)s2;
  if (dbeta) dbeta[c] = (beta ? dbeta[c] : 0.f) + (float)s1;
  if (dgamma) dgamma[c] = (beta ? dgamma[c] : 0.f) + (float)s2;
}

// backward pass 2: dx = gamma*invstd*(dU - sum_dU/n - xhat*sum_dUxhat/n) (+res)
template <int ACT, int NTMODE, typename T = float>
__global__ __launch_bounds__(256) void bn_act_bwd_dx_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ coef,
                                                            const T* __restrict__ res, T* __restrict__ dx,
                                                            int C, int HW, float inv_count, int training, int vec) {
  constexpr int W = ElemW<T>::W;
  const int plane = blockIdx.x;
  const int c = plane % C;
  const float a = scale[c], b = shift[c], mu = mean[c], is = invstd[c];
  const float gi = gamma[c] * is;
  const float k1 = training ? coef[2 * c + 0] * inv_count : 0.f;
  const float k2 = training ? coef[2 * c + 1] * inv_count : 0.f;
  const long base = (long)plane * HW;
  auto one = [&](float xv, float dv, float rv) {
    const float du = dv * act_grad_f<ACT>(fmaf(xv, a, b));
    const float xh = (xv - mu) * is;
    return gi * (du - k1 - xh * k2) + rv;
  };
  if (vec) {
    const int nv = HW / W;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < nv; i += gridDim.y * blockDim.x) {
      const long off = base + (long)i * W;
      float xv[W], dv[W], rv[W], o[W];
#pragma unroll
      for (int q = 0; q < W; ++q) rv[q] = 0.f;
      // streamed once: keep them out of the way of the GEMM operands in L2
      ldv<NTMODE != 0>(x + off, xv);
      ldv<NTMODE != 0>(dy + off, dv);
      if (res) ldv<NTMODE != 0>(res + off, rv);
#pragma unroll
      for (int q = 0; q < W; ++q) o[q] = one(xv[q], dv[q], rv[q]);
      stv<NTMODE == 2>(dx + off, o);
    }
  } else {
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < HW; i += gridDim.y * blockDim.x)
      st1(dx + base + i, one(ld1(x + base + i), ld1(dy + base + i), res ? ld1(res + base + i) : 0.f));
  }
}

// ------------------------------------------------------------ element-wise
enum EwOp { EW_GELU = 0, EW_GELU_BWD = 1, EW_SIGMOID = 2, EW_SIGMOID_BWD = 3, EW_ADD = 4, EW_LRELU = 5, EW_LRELU_BWD = 6 };

template <int OP>
__device__ __forceinline__ float ew_apply(float a, float b) {
  if (OP == EW_GELU) return gelu_f(a);
  if (OP == EW_GELU_BWD) return a * gelu_grad_f(b);           // a = dy, b = x
  if (OP == EW_SIGMOID) return sigmoid_f(a);
  if (OP == EW_SIGMOID_BWD) return a * b * (1.f - b);         // a = dy, b = y
  if (OP == EW_LRELU) return a > 0.f ? a : kLeaky * a;
  if (OP == EW_LRELU_BWD) return b > 0.f ? a : kLeaky * a;    // a = dy, b = x
  return a + b;
}

template <int OP>
__global__ __launch_bounds__(256) void ew_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                 float* __restrict__ out, long n, int vec) {
  const long stride = (long)gridDim.x * blockDim.x;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    const long n4 = n >> 2;
    for (long i = i0; i < n4; i += stride) {
      const float4 av = reinterpret_cast<const float4*>(a)[i];
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b) bv = reinterpret_cast<const float4*>(b)[i];
      float4 o;
      o.x = ew_apply<OP>(av.x, bv.x); o.y = ew_apply<OP>(av.y, bv.y);
      o.z = ew_apply<OP>(av.z, bv.z); o.w = ew_apply<OP>(av.w, bv.w);
      reinterpret_cast<float4*>(out)[i] = o;
    }
    for (long i = (n4 << 2) + i0; i < n; i += stride) out[i] = ew_apply<OP>(a[i], b ? b[i] : 0.f);
  } else {
    for (long i = i0; i < n; i += stride) out[i] = ew_apply<OP>(a[i], b ? b[i] : 0.f);
  }
}

template <int OP>
int launch_ew(const float* a, const float* b, float* out, int64_t n, hipStream_t st, const char* what) {
  WFAE_REQUIRE(a && out, WFAE_ERR_NULL_POINTER, "%s: null pointer", what);
  WFAE_REQUIRE(n >= 0, WFAE_ERR_BAD_SHAPE, "%s: bad size", what);
  if (n == 0) return WFAE_OK;
  const int vec = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) |
                    reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((ew_kernel<OP>), dim3((unsigned)blocks), dim3(256), 0, st, a, b, out, (long)n, vec);
  return check_launch(what);
}

// ---------------------------------------------------------- sigmoid + L1
// mode 0: recon = sigmoid(h), partial sum |recon - x| ; mode 1: h already is recon (plain L1)
template <int MODE>
__global__ __launch_bounds__(RT) void l1_fwd_kernel(const float* __restrict__ h, const float* __restrict__ x,
                                                    float* __restrict__ recon, double* __restrict__ part, long n,
                                                    int vec) {
  __shared__ double sm[16];
  const long stride = (long)gridDim.x * blockDim.x;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  double s = 0.0;
  auto one = [&](float hv, float xv) {
    const float r = MODE == 0 ? sigmoid_f(hv) : hv;
    s += (double)fabsf(r - xv);
    return r;
  };
  if (vec) {
    const long n4 = n >> 2;
    for (long i = i0; i < n4; i += stride) {
      const float4 hv = reinterpret_cast<const float4*>(h)[i];
      const float4 xv = reinterpret_cast<const float4*>(x)[i];
      float4 r;
      r.x = one(hv.x, xv.x); r.y = one(hv.y, xv.y); r.z = one(hv.z, xv.z); r.w = one(hv.w, xv.w);
      if (MODE == 0) reinterpret_cast<float4*>(recon)[i] = r;
    }
    for (long i = (n4 << 2) + i0; i < n; i += stride) {
      const float r = one(h[i], x[i]);
      if (MODE == 0) recon[i] = r;
    }
  } else {
    for (long i = i0; i < n; i += stride) {
      const float r = one(h[i], x[i]);
      if (MODE == 0) recon[i] = r;
    }
  }
  const double r = block_sum(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

__global__ void scalar_finalize_kernel(const double* __restrict__ part, int parts, double mul, float* out_f,
                                       double* out_d) {
  __shared__ double sm[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < parts; i += blockDim.x) s += part[i];
  const double r = block_sum(s, sm);
  if (threadIdx.x == 0) {
    if (out_f) out_f[0] = (float)(r * mul);
    if (out_d) out_d[0] = r * mul;
  }
}

// mode 0: dh = g*w/n * sign(r-x) * r(1-r) ; mode 1: drecon = g*w/n * sign(r-x)
template <int MODE>
__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ recon, const float* __restrict__ x,
                                                     const float* __restrict__ gloss, float wn,
                                                     float* __restrict__ dh, long n) {
  const float g = gloss[0] * wn;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float r = recon[i];
    const float d = r - x[i];
    float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    if (MODE == 0) sg *= r * (1.f - r);
    dh[i] = g * sg;
  }
}

__global__ __launch_bounds__(RT) void sumsq_kernel(const float* __restrict__ x, double* __restrict__ part, long n) {
  __shared__ double sm[16];
  const long stride = (long)gridDim.x * blockDim.x;
  double s = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += (double)x[i] * x[i];
  const double r = block_sum(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1,
                                                    float bc2_sqrt, float gscale) {
  const long stride = (long)gridDim.x * blockDim.x;
  const float step = lr / bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gv = g[i] * gscale;
    float pv = p[i];
    pv *= 1.f - lr * wd;
    const float mv = m[i] + (gv - m[i]) * (1.f - b1);       // lerp form used by torch (_single_tensor_adamw)
    const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = pv - step * (mv / denom);
  }
}

__global__ void vil_u8_to_f32_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int H, int W, int T,
                                     float scale, long total) {
  // dst[n][t][h][w] = scale * src[n][h][w][t]
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int w = (int)(i % W);
    long r = i / W;
    const int h = (int)(r % H);
    r /= H;
    const int t = (int)(r % T);
    const long n = r / T;
    dst[i] = scale * ((float)src[((n * H + h) * W + w) * T + t] + 0.f);
  }
}

// zero padding of every (n,c) plane by `pad` pixels (mode 0) / cropping the interior back (mode 1)
template <int MODE>
__global__ __launch_bounds__(256) void pad2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W,
                                                    int pad, long total) {
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    if (MODE == 0) {  // i indexes the padded tensor
      const int xw = (int)(i % Wp);
      const long r = i / Wp;
      const int yh = (int)(r % Hp);
      const long pl = r / Hp;
      const int sy = yh - pad, sx = xw - pad;
      dst[i] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? src[(pl * H + sy) * W + sx] : 0.f;
    } else {  // i indexes the cropped tensor
      const int xw = (int)(i % W);
      const long r = i / W;
      const int yh = (int)(r % H);
      const long pl = r / H;
      dst[i] = src[(pl * Hp + yh + pad) * Wp + xw + pad];
    }
  }
}

// mode 0: sum x ; mode 1: sum relu(1 + sign*x)   (hinge terms, contperceptual.py:19-23)
template <int MODE>
__global__ __launch_bounds__(RT) void mean_part_kernel(const float* __restrict__ x, double* __restrict__ part, long n,
                                                       float sign) {
  __shared__ double sm[16];
  const long stride = (long)gridDim.x * blockDim.x;
  double s = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    s += MODE == 0 ? (double)x[i] : (double)fmaxf(1.f + sign * x[i], 0.f);
  const double r = block_sum(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

// mode 0: dx = g*w ; mode 1: dx = g*w*sign*[1 + sign*x > 0]
template <int MODE>
__global__ __launch_bounds__(256) void mean_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, float w,
                                                       float sign, float* __restrict__ dx, long n) {
  const float gv = g[0] * w;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dx[i] = MODE == 0 ? gv : ((1.f + sign * x[i] > 0.f) ? gv * sign : 0.f);
}

__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ x, const float* __restrict__ sdev, float s,
                                                    float* __restrict__ y, long n) {
  const float f = sdev ? sdev[0] * s : s;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = x[i] * f;
}

// torch.nn.utils.clip_grad_norm_ coefficient from partial sums of squares:
// out[0] = min(1, max_norm / (sqrt(sum) + 1e-6)), out[1] = sqrt(sum)
__global__ void clip_coef_kernel(const double* __restrict__ parts, int n, float max_norm, float pre_scale,
                                 float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += parts[i];
    const float total = (float)sqrt(s) * pre_scale;
    const float coef = max_norm / (total + 1e-6f);
    out[0] = coef < 1.f ? coef : 1.f;
    out[1] = total;
  }
}

// Loss.calculate_adaptive_weight (experiments/ae_v2_2/train.py:46-52):
// clamp(disc_weight * ||rec_grad|| / (||disc_grad|| + 1e-4), 0, 1e4)
__global__ void adaptive_weight_kernel(const double* __restrict__ ss_rec, const double* __restrict__ ss_disc,
                                       float disc_weight, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float nr = (float)sqrt(ss_rec[0]), nd = (float)sqrt(ss_disc[0]);
    float w = disc_weight * (nr / (nd + 1e-4f));
    w = fminf(fmaxf(w, 0.f), 1e4f);
    out[0] = w;
  }
}

// fp32 <-> bf16 storage conversion (the boundaries of the bf16-storage mode: latent projections, tests)
template <bool TO_BF16>
__global__ __launch_bounds__(256) void convert_kernel(const void* __restrict__ src, void* __restrict__ dst, long n) {
  const long stride = (long)gridDim.x * blockDim.x * 8;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      float v[8];
      if (TO_BF16) {
        float a[4], b[4];
        ldv((const float*)src + i, a);
        ldv((const float*)src + i + 4, b);
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] = a[q]; v[4 + q] = b[q]; }
        stv((bf16_t*)dst + i, v);
      } else {
        ldv((const bf16_t*)src + i, v);
        const float a[4] = {v[0], v[1], v[2], v[3]}, b[4] = {v[4], v[5], v[6], v[7]};
        stv((float*)dst + i, a);
        stv((float*)dst + i + 4, b);
      }
    } else {
      for (long j = i; j < n; ++j) {
        if (TO_BF16) st1((bf16_t*)dst + j, ((const float*)src)[j]);
        else ((float*)dst)[j] = ld1((const bf16_t*)src + j);
      }
    }
  }
}

inline int grid_1d(long n, int per_thread = 4) {
  long b = (n / per_thread + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

template <typename T>
int bn_stats_train_impl(const T* x, int NB, int C, int HW, const float* gamma, const float* beta, float eps, float momentum,
                        float* running_mean, float* running_var, float* save_mean, float* save_invstd, float* scale,
                        float* shift, void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(x && gamma && beta && save_mean && save_invstd && scale && shift, WFAE_ERR_NULL_POINTER,
               "bn_stats_train: null pointer");
  WFAE_REQUIRE(NB > 0 && C > 0 && HW > 0 && C <= 65535 * 32, WFAE_ERR_BAD_SHAPE, "bn_stats_train: bad shape");
  const long maxp = (long)(ws_bytes / (sizeof(double) * 2 * (size_t)C));
  WFAE_REQUIRE(ws && maxp >= 1, WFAE_ERR_WORKSPACE, "bn_stats_train: workspace too small");
  RedGeom g = red_geom(NB, C, HW, maxp < 65535 ? maxp : 65535);
  const int vec = (HW % ElemW<T>::W == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((chan_reduce_kernel<0, T>), dim3(C, g.splits), dim3(RT), 0, st, x, (double*)ws, g, vec);
  int rc = check_launch("bn_stats");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, (const double*)ws, g.splits,
                     (long)NB * HW, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean,
                     save_invstd, scale, shift);
  return check_launch("bn_finalize");
}

template <typename T>
int bn_act_fwd_impl(const T* x, const float* scale, const float* shift, T* y, int NB, int C, int HW, int act,
                    double* part, int64_t part_capacity, int* splits_out, wfae_stream_t stream) {
  WFAE_REQUIRE(x && scale && shift && y, WFAE_ERR_NULL_POINTER, "bn_act_fwd: null pointer");
  WFAE_REQUIRE(NB > 0 && C > 0 && HW > 0, WFAE_ERR_BAD_SHAPE, "bn_act_fwd: bad shape");
  constexpr int W = ElemW<T>::W;
  const int vec = (HW % W == 0) && (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0);
  int gy = cdiv(vec ? HW / W : HW, 256 * 4);
  if (gy < 1) gy = 1;
  if (gy > 1024) gy = 1024;
  dim3 grid((unsigned)((long)NB * C), gy);
  hipStream_t st = (hipStream_t)stream;
  const int nt_fwd = 0;   // nontemporal store of y: measured neutral in round 2
  if (splits_out) {
    const int64_t splits = (int64_t)NB * gy;
    WFAE_REQUIRE(part && part_capacity >= splits * C * 2, WFAE_ERR_WORKSPACE, "bn_act_fwd_stats: part holds %lld doubles, needs %lld",
                 (long long)part_capacity, (long long)(splits * C * 2));
    WFAE_REQUIRE(splits < (1ll << 31), WFAE_ERR_BAD_SHAPE, "bn_act_fwd_stats: too many partial rows");
    *splits_out = (int)splits;
    if (act == 1)
      hipLaunchKernelGGL((bn_act_fwd_kernel<1, true, T>), grid, dim3(256), 0, st, x, scale, shift, y, C, HW, vec, nt_fwd, part);
    else if (act == 2)
      hipLaunchKernelGGL((bn_act_fwd_kernel<2, true, T>), grid, dim3(256), 0, st, x, scale, shift, y, C, HW, vec, nt_fwd, part);
    else
      hipLaunchKernelGGL((bn_act_fwd_kernel<0, true, T>), grid, dim3(256), 0, st, x, scale, shift, y, C, HW, vec, nt_fwd, part);
    return check_launch("bn_act_fwd_stats");
  }
  if (act == 1)
    hipLaunchKernelGGL((bn_act_fwd_kernel<1, false, T>), grid, dim3(256), 0, st, x, scale, shift, y, C, HW, vec, nt_fwd, (double*)nullptr);
  else if (act == 2)
    hipLaunchKernelGGL((bn_act_fwd_kernel<2, false, T>), grid, dim3(256), 0, st, x, scale, shift, y, C, HW, vec, nt_fwd, (double*)nullptr);
  else
    hipLaunchKernelGGL((bn_act_fwd_kernel<0, false, T>), grid, dim3(256), 0, st, x, scale, shift, y, C, HW, vec, nt_fwd, (double*)nullptr);
  return check_launch("bn_act_fwd");
}

template <typename T>
int bn_act_bwd_impl(const T* dy, const T* x, const float* gamma, const float* scale, const float* shift,
                    const float* save_mean, const float* save_invstd, const T* res, T* dx, float* dgamma, float* dbeta, int NB,
                    int C, int HW, int act, int training, int accumulate, int phases, void* ws, size_t ws_bytes,
                    wfae_stream_t stream) {
  WFAE_REQUIRE(dy && x && gamma && scale && shift && save_mean && save_invstd, WFAE_ERR_NULL_POINTER,
               "bn_act_bwd: null pointer");
  WFAE_REQUIRE(phases >= 1 && phases <= 3, WFAE_ERR_BAD_SHAPE, "bn_act_bwd: phases must be 1, 2 or 3");
  WFAE_REQUIRE(NB > 0 && C > 0 && HW > 0, WFAE_ERR_BAD_SHAPE, "bn_act_bwd: bad shape");
  // workspace: coef[2C] floats (16-byte aligned) followed by fp64 partials
  const size_t coef_bytes = ((size_t)2 * C * sizeof(float) + 15) / 16 * 16;
  WFAE_REQUIRE(ws && ws_bytes > coef_bytes + sizeof(double) * 2 * (size_t)C, WFAE_ERR_WORKSPACE,
               "bn_act_bwd: workspace too small");
  constexpr int W = ElemW<T>::W;
  float* coef = (float*)ws;
  double* part = (double*)((char*)ws + coef_bytes);
  const long maxp = (long)((ws_bytes - coef_bytes) / (sizeof(double) * 2 * (size_t)C));
  // (round 4, profiles/r04_kbench_bn_reduce_sweep.txt: 1 / 2 / 4 chunks of loads in flight per thread x 512 .. 4096 blocks in the
  // launch all measure the same, 6.0 TB/s on the 2.4 GB tensors and 4.8 - 5.4 on the sub-0.2 ms launches of the narrow / deep
  // stages, whose ~20 us of fixed cost per reduce + finalize pair is what the step's 5.15 TB/s average shows)
  RedGeom g = red_geom(NB, C, HW, maxp < 65535 ? maxp : 65535);
  const int vec = (HW % W == 0) && (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) |
                                      reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(res)) & 15) == 0);
  hipStream_t st = (hipStream_t)stream;
  int rc = WFAE_OK;
  if (phases & 1) {
    if (act == 1)
      hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<1, T>), dim3(C, g.splits), dim3(RT), 0, st, dy, x, scale, shift,
                         save_mean, save_invstd, part, g, vec);
    else if (act == 2)
      hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<2, T>), dim3(C, g.splits), dim3(RT), 0, st, dy, x, scale, shift,
                         save_mean, save_invstd, part, g, vec);
    else
      hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<0, T>), dim3(C, g.splits), dim3(RT), 0, st, dy, x, scale, shift,
                         save_mean, save_invstd, part, g, vec);
    rc = check_launch("bn_act_bwd_reduce");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, (const double*)part, g.splits,
                       C, dgamma, dbeta, coef, accumulate);
    rc = check_launch("bn_bwd_finalize");
    if (rc) return rc;
  }
  if ((phases & 2) && dx) {
    int gy = cdiv(vec ? HW / W : HW, 256 * 4);
    if (gy < 1) gy = 1;
    if (gy > 1024) gy = 1024;
    dim3 grid((unsigned)((long)NB * C), gy);
    const float inv_count = 1.0f / (float)((double)NB * HW);
    // nontemporal loads + stores for the GELU form (measured, tools/kbench.py, B = 32: the nontemporal STORE of dx lifts
    // the kernel pair from 5.0-5.4 to 5.7-6.0 TB/s; the loads alone change nothing)
#define WFAE_DX(ACT_, NT_)                                                                                       \
  hipLaunchKernelGGL((bn_act_bwd_dx_kernel<ACT_, NT_, T>), grid, dim3(256), 0, st, dy, x, gamma, scale, shift, save_mean, \
                     save_invstd, coef, res, dx, C, HW, inv_count, training, vec)
    if (act == 1) WFAE_DX(1, 2);
    else if (act == 2) WFAE_DX(2, 0);
    else WFAE_DX(0, 0);
#undef WFAE_DX
    rc = check_launch("bn_act_bwd_dx");
  }
  return rc;
}

}  // namespace

extern "C" {

int wfae_bn_stats_train(const float* x, int NB, int C, int HW, const float* gamma, const float* beta,
                        float eps, float momentum, float* running_mean, float* running_var,
                        float* save_mean, float* save_invstd, float* scale, float* shift, void* ws,
                        size_t ws_bytes, wfae_stream_t stream) {
  return bn_stats_train_impl(x, NB, C, HW, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, scale,
                             shift, ws, ws_bytes, stream);
}
int wfae_bn_stats_train_bf16(const uint16_t* x, int NB, int C, int HW, const float* gamma, const float* beta,
                             float eps, float momentum, float* running_mean, float* running_var,
                             float* save_mean, float* save_invstd, float* scale, float* shift, void* ws,
                             size_t ws_bytes, wfae_stream_t stream) {
  return bn_stats_train_impl(x, NB, C, HW, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, scale,
                             shift, ws, ws_bytes, stream);
}

int wfae_bn_stats_from_rows(const double* stat_part, int rows, int NB, int C, int HW, const float* gamma,
                            const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                            float* save_mean, float* save_invstd, float* scale, float* shift, void* ws, size_t ws_bytes,
                            wfae_stream_t stream) {
  WFAE_REQUIRE(stat_part && gamma && beta && save_mean && save_invstd && scale && shift, WFAE_ERR_NULL_POINTER,
               "bn_stats_from_rows: null pointer");
  WFAE_REQUIRE(rows > 0 && NB > 0 && C > 0 && HW > 0, WFAE_ERR_BAD_SHAPE, "bn_stats_from_rows: bad shape");
  // enough (channel block, row slice) blocks to fill the chip: the partial rows are ~3 % of the tensor, read once
  int splits = (int)((1024l * 64 + C - 1) / C);
  if (splits > rows / 8) splits = rows / 8;
  if (splits < 1) splits = 1;
  if (splits > 1024) splits = 1024;
  WFAE_REQUIRE(ws && ws_bytes >= sizeof(double) * 2 * (size_t)C * splits, WFAE_ERR_WORKSPACE,
               "bn_stats_from_rows: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(stat_rows_reduce_kernel, dim3(cdiv(C, 64), splits), dim3(256), 0, st, stat_part,
                     stat_part + (long)rows * C, (double*)ws, rows, C, splits);
  int rc = check_launch("stat_rows_reduce");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_finalize_parts_kernel, dim3(C), dim3(64), 0, st, (const double*)ws, splits, 1l, (long)splits,
                     (long)NB * HW, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, scale,
                     shift);
  return check_launch("bn_finalize");
}

int wfae_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, float* save_mean, float* save_invstd,
                      float* scale, float* shift, int C, wfae_stream_t stream) {
  WFAE_REQUIRE(gamma && beta && running_mean && running_var && save_mean && save_invstd && scale && shift,
               WFAE_ERR_NULL_POINTER, "bn_fold_eval: null pointer");
  WFAE_REQUIRE(C > 0, WFAE_ERR_BAD_SHAPE, "bn_fold_eval: bad shape");
  hipLaunchKernelGGL(bn_fold_eval_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, save_mean, save_invstd, scale, shift, C);
  return check_launch("bn_fold_eval");
}

int wfae_bn_act_fwd(const float* x, const float* scale, const float* shift, float* y, int NB, int C,
                    int HW, int act, wfae_stream_t stream) {
  return bn_act_fwd_impl(x, scale, shift, y, NB, C, HW, act, nullptr, 0, nullptr, stream);
}
int wfae_bn_act_fwd_bf16(const uint16_t* x, const float* scale, const float* shift, uint16_t* y, int NB, int C,
                         int HW, int act, wfae_stream_t stream) {
  return bn_act_fwd_impl(x, scale, shift, y, NB, C, HW, act, nullptr, 0, nullptr, stream);
}

int wfae_bn_act_fwd_stats(const float* x, const float* scale, const float* shift, float* y, int NB, int C, int HW, int act,
                          double* part, int64_t part_capacity, int* splits_out, wfae_stream_t stream) {
  WFAE_REQUIRE(part && splits_out, WFAE_ERR_NULL_POINTER, "bn_act_fwd_stats: null pointer");
  return bn_act_fwd_impl(x, scale, shift, y, NB, C, HW, act, part, part_capacity, splits_out, stream);
}
int wfae_bn_act_fwd_stats_bf16(const uint16_t* x, const float* scale, const float* shift, uint16_t* y, int NB, int C, int HW,
                               int act, double* part, int64_t part_capacity, int* splits_out, wfae_stream_t stream) {
  WFAE_REQUIRE(part && splits_out, WFAE_ERR_NULL_POINTER, "bn_act_fwd_stats: null pointer");
  return bn_act_fwd_impl(x, scale, shift, y, NB, C, HW, act, part, part_capacity, splits_out, stream);
}

int wfae_bn_stats_from_parts(const double* part, int splits, int NB, int C, int HW, const float* gamma, const float* beta,
                             float eps, float momentum, float* running_mean, float* running_var, float* save_mean,
                             float* save_invstd, float* scale, float* shift, wfae_stream_t stream) {
  WFAE_REQUIRE(part && gamma && beta && save_mean && save_invstd && scale && shift, WFAE_ERR_NULL_POINTER,
               "bn_stats_from_parts: null pointer");
  WFAE_REQUIRE(splits > 0 && NB > 0 && C > 0 && HW > 0, WFAE_ERR_BAD_SHAPE, "bn_stats_from_parts: bad shape");
  hipLaunchKernelGGL(bn_finalize_parts_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, part, splits, (long)C, 1l, (long)NB * HW,
                     C, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, scale, shift);
  return check_launch("bn_finalize_parts");
}

int wfae_bn_act_bwd_from_rows(const double* part_rows, int rows, int C, float* dgamma, float* dbeta, int accumulate, void* ws,
                              size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(part_rows && ws, WFAE_ERR_NULL_POINTER, "bn_act_bwd_from_rows: null pointer");
  WFAE_REQUIRE(rows > 0 && C > 0, WFAE_ERR_BAD_SHAPE, "bn_act_bwd_from_rows: bad shape");
  // the workspace layout of wfae_bn_act_bwd: coef[2C] floats (16-byte aligned), then fp64 partials part[(c * splits + s) * 2]
  const size_t coef_bytes = ((size_t)2 * C * sizeof(float) + 15) / 16 * 16;
  int splits = (int)((1024l * 64 + C - 1) / C);
  if (splits > rows / 8) splits = rows / 8;
  if (splits < 1) splits = 1;
  if (splits > 1024) splits = 1024;
  WFAE_REQUIRE(ws_bytes >= coef_bytes + sizeof(double) * 2 * (size_t)C * splits, WFAE_ERR_WORKSPACE,
               "bn_act_bwd_from_rows: workspace too small");
  float* coef = (float*)ws;
  double* part = (double*)((char*)ws + coef_bytes);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(stat_rows_reduce_kernel, dim3(cdiv(C, 64), splits), dim3(256), 0, st, part_rows, part_rows + (long)rows * C,
                     part, rows, C, splits);
  int rc = check_launch("stat_rows_reduce");
  if (rc) return rc;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, (const double*)part, splits, C, dgamma, dbeta,
                     coef, accumulate);
  return check_launch("bn_bwd_finalize");
}

int wfae_bn_act_bwd(const float* dy, const float* x, const float* gamma, const float* scale,
                    const float* shift, const float* save_mean, const float* save_invstd,
                    const float* res, float* dx, float* dgamma, float* dbeta, int NB, int C, int HW,
                    int act, int training, int accumulate, int phases, void* ws, size_t ws_bytes,
                    wfae_stream_t stream) {
  return bn_act_bwd_impl(dy, x, gamma, scale, shift, save_mean, save_invstd, res, dx, dgamma, dbeta, NB, C, HW, act, training,
                         accumulate, phases, ws, ws_bytes, stream);
}
int wfae_bn_act_bwd_bf16(const uint16_t* dy, const uint16_t* x, const float* gamma, const float* scale,
                         const float* shift, const float* save_mean, const float* save_invstd,
                         const uint16_t* res, uint16_t* dx, float* dgamma, float* dbeta, int NB, int C, int HW,
                         int act, int training, int accumulate, int phases, void* ws, size_t ws_bytes,
                         wfae_stream_t stream) {
  return bn_act_bwd_impl(dy, x, gamma, scale, shift, save_mean, save_invstd, res, dx, dgamma, dbeta, NB, C, HW, act, training,
                         accumulate, phases, ws, ws_bytes, stream);
}

int wfae_convert_f32_to_bf16(const float* src, uint16_t* dst, int64_t n, wfae_stream_t stream) {
  WFAE_REQUIRE(src && dst, WFAE_ERR_NULL_POINTER, "convert_f32_to_bf16: null pointer");
  WFAE_REQUIRE(n > 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0, WFAE_ERR_BAD_SHAPE,
               "convert_f32_to_bf16: bad size / alignment");
  hipLaunchKernelGGL((convert_kernel<true>), dim3(grid_1d(n, 8)), dim3(256), 0, (hipStream_t)stream, (const void*)src, (void*)dst,
                     (long)n);
  return check_launch("convert_f32_to_bf16");
}
int wfae_convert_bf16_to_f32(const uint16_t* src, float* dst, int64_t n, wfae_stream_t stream) {
  WFAE_REQUIRE(src && dst, WFAE_ERR_NULL_POINTER, "convert_bf16_to_f32: null pointer");
  WFAE_REQUIRE(n > 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0, WFAE_ERR_BAD_SHAPE,
               "convert_bf16_to_f32: bad size / alignment");
  hipLaunchKernelGGL((convert_kernel<false>), dim3(grid_1d(n, 8)), dim3(256), 0, (hipStream_t)stream, (const void*)src, (void*)dst,
                     (long)n);
  return check_launch("convert_bf16_to_f32");
}

int wfae_gelu_fwd(const float* x, float* y, int64_t n, wfae_stream_t s) {
  return launch_ew<EW_GELU>(x, nullptr, y, n, (hipStream_t)s, "gelu_fwd");
}
int wfae_gelu_bwd(const float* dy, const float* x, float* dx, int64_t n, wfae_stream_t s) {
  WFAE_REQUIRE(x, WFAE_ERR_NULL_POINTER, "gelu_bwd: null pointer");
  return launch_ew<EW_GELU_BWD>(dy, x, dx, n, (hipStream_t)s, "gelu_bwd");
}
int wfae_sigmoid_fwd(const float* x, float* y, int64_t n, wfae_stream_t s) {
  return launch_ew<EW_SIGMOID>(x, nullptr, y, n, (hipStream_t)s, "sigmoid_fwd");
}
int wfae_sigmoid_bwd(const float* dy, const float* y, float* dx, int64_t n, wfae_stream_t s) {
  WFAE_REQUIRE(y, WFAE_ERR_NULL_POINTER, "sigmoid_bwd: null pointer");
  return launch_ew<EW_SIGMOID_BWD>(dy, y, dx, n, (hipStream_t)s, "sigmoid_bwd");
}
int wfae_add(const float* a, const float* b, float* out, int64_t n, wfae_stream_t s) {
  WFAE_REQUIRE(b, WFAE_ERR_NULL_POINTER, "add: null pointer");
  return launch_ew<EW_ADD>(a, b, out, n, (hipStream_t)s, "add");
}

int wfae_reduce_sum(const float* x, int outer, int C, int inner, float* out, int accumulate, void* ws,
                    size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(x && out, WFAE_ERR_NULL_POINTER, "reduce_sum: null pointer");
  WFAE_REQUIRE(outer > 0 && C > 0 && inner > 0, WFAE_ERR_BAD_SHAPE, "reduce_sum: bad shape");
  const long maxp = (long)(ws_bytes / (sizeof(double) * 2 * (size_t)C));
  WFAE_REQUIRE(ws && maxp >= 1, WFAE_ERR_WORKSPACE, "reduce_sum: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (inner == 1) {  // x[outer][C]: column sums in two coalesced stages
    int splits = cdiv(outer, 64);
    if (splits > maxp) splits = (int)maxp;
    if (splits > 1024) splits = 1024;
    const int rps = cdiv(outer, splits);
    splits = cdiv(outer, rps);
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, 64), splits), dim3(1024), 0, st, x, outer, C, rps, (double*)ws, splits);
    int rc = check_launch("reduce_sum_cols");
    if (rc) return rc;
    hipLaunchKernelGGL(sum_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, (const double*)ws, splits, C, out,
                       accumulate);
    return check_launch("reduce_sum_finalize");
  }
  // channel index goes to grid.x: fold very wide C (linear bias, pos_emb) into chunks of 65535*... (grid.x is 2^31)
  RedGeom g = red_geom(outer, C, inner, maxp < 65535 ? maxp : 65535);
  const int vec = (inner % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  hipLaunchKernelGGL((chan_reduce_kernel<1>), dim3(C, g.splits), dim3(RT), 0, st, x, (double*)ws, g, vec);
  int rc = check_launch("reduce_sum");
  if (rc) return rc;
  hipLaunchKernelGGL(sum_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, (const double*)ws, g.splits, C, out,
                     accumulate);
  return check_launch("reduce_sum_finalize");
}

static int l1_common(int mode, const float* h, const float* x, float* recon, float* loss, float weight,
                     int64_t n, void* ws, size_t ws_bytes, hipStream_t st) {
  WFAE_REQUIRE(h && x && loss && (mode == 1 || recon), WFAE_ERR_NULL_POINTER, "l1_fwd: null pointer");
  WFAE_REQUIRE(n > 0, WFAE_ERR_BAD_SHAPE, "l1_fwd: bad size");
  int blocks = grid_1d(n, 16);
  if (blocks > 1024) blocks = 1024;
  WFAE_REQUIRE(ws && ws_bytes >= (size_t)blocks * sizeof(double), WFAE_ERR_WORKSPACE, "l1_fwd: workspace too small");
  const int vec = ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(x) |
                    reinterpret_cast<uintptr_t>(recon)) & 15) == 0;
  if (mode == 0)
    hipLaunchKernelGGL((l1_fwd_kernel<0>), dim3(blocks), dim3(RT), 0, st, h, x, recon, (double*)ws, (long)n, vec);
  else
    hipLaunchKernelGGL((l1_fwd_kernel<1>), dim3(blocks), dim3(RT), 0, st, h, x, recon, (double*)ws, (long)n, vec);
  int rc = check_launch("l1_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(scalar_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, blocks,
                     (double)weight / (double)n, loss, (double*)nullptr);
  return check_launch("l1_finalize");
}

int wfae_sigmoid_l1_fwd(const float* h, const float* x, float* recon, float* loss, float weight, int64_t n,
                        void* ws, size_t ws_bytes, wfae_stream_t stream) {
  return l1_common(0, h, x, recon, loss, weight, n, ws, ws_bytes, (hipStream_t)stream);
}
int wfae_l1_fwd(const float* recon, const float* x, float* loss, float weight, int64_t n, void* ws,
                size_t ws_bytes, wfae_stream_t stream) {
  return l1_common(1, recon, x, nullptr, loss, weight, n, ws, ws_bytes, (hipStream_t)stream);
}

int wfae_sigmoid_l1_bwd(const float* recon, const float* x, const float* gloss, float weight, float* dh,
                        int64_t n, wfae_stream_t stream) {
  WFAE_REQUIRE(recon && x && gloss && dh, WFAE_ERR_NULL_POINTER, "sigmoid_l1_bwd: null pointer");
  WFAE_REQUIRE(n > 0, WFAE_ERR_BAD_SHAPE, "sigmoid_l1_bwd: bad size");
  hipLaunchKernelGGL((l1_bwd_kernel<0>), dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, recon, x, gloss,
                     (float)((double)weight / (double)n), dh, (long)n);
  return check_launch("sigmoid_l1_bwd");
}
int wfae_l1_bwd(const float* recon, const float* x, const float* gloss, float weight, float* drecon, int64_t n,
                wfae_stream_t stream) {
  WFAE_REQUIRE(recon && x && gloss && drecon, WFAE_ERR_NULL_POINTER, "l1_bwd: null pointer");
  WFAE_REQUIRE(n > 0, WFAE_ERR_BAD_SHAPE, "l1_bwd: bad size");
  hipLaunchKernelGGL((l1_bwd_kernel<1>), dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, recon, x, gloss,
                     (float)((double)weight / (double)n), drecon, (long)n);
  return check_launch("l1_bwd");
}

int wfae_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
               float eps, float weight_decay, float bias_corr1, float bias_corr2, float grad_scale,
               wfae_stream_t stream) {
  WFAE_REQUIRE(p && g && m && v, WFAE_ERR_NULL_POINTER, "adamw: null pointer");
  WFAE_REQUIRE(n >= 0, WFAE_ERR_BAD_SHAPE, "adamw: bad size");
  if (n == 0) return WFAE_OK;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, lr,
                     beta1, beta2, eps, weight_decay, bias_corr1, sqrtf(bias_corr2), grad_scale);
  return check_launch("adamw");
}

int wfae_sumsq(const float* x, int64_t n, double* out, void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(x && out, WFAE_ERR_NULL_POINTER, "sumsq: null pointer");
  WFAE_REQUIRE(n > 0, WFAE_ERR_BAD_SHAPE, "sumsq: bad size");
  int blocks = grid_1d(n, 16);
  if (blocks > 1024) blocks = 1024;
  WFAE_REQUIRE(ws && ws_bytes >= (size_t)blocks * sizeof(double), WFAE_ERR_WORKSPACE, "sumsq: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(RT), 0, st, x, (double*)ws, (long)n);
  int rc = check_launch("sumsq");
  if (rc) return rc;
  hipLaunchKernelGGL(scalar_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, blocks, 1.0,
                     (float*)nullptr, out);
  return check_launch("sumsq_finalize");
}

int wfae_vil_u8_to_f32(const uint8_t* src, float* dst, int NB, int H, int W, int T, float scale,
                       wfae_stream_t stream) {
  WFAE_REQUIRE(src && dst, WFAE_ERR_NULL_POINTER, "vil_u8_to_f32: null pointer");
  WFAE_REQUIRE(NB > 0 && H > 0 && W > 0 && T > 0, WFAE_ERR_BAD_SHAPE, "vil_u8_to_f32: bad shape");
  const long total = (long)NB * T * H * W;
  hipLaunchKernelGGL(vil_u8_to_f32_kernel, dim3(grid_1d(total, 1)), dim3(256), 0, (hipStream_t)stream, src, dst, H,
                     W, T, scale, total);
  return check_launch("vil_u8_to_f32");
}

int wfae_leaky_relu_fwd(const float* x, float* y, int64_t n, wfae_stream_t s) {
  return launch_ew<EW_LRELU>(x, nullptr, y, n, (hipStream_t)s, "leaky_relu_fwd");
}
int wfae_leaky_relu_bwd(const float* dy, const float* x, float* dx, int64_t n, wfae_stream_t s) {
  WFAE_REQUIRE(x, WFAE_ERR_NULL_POINTER, "leaky_relu_bwd: null pointer");
  return launch_ew<EW_LRELU_BWD>(dy, x, dx, n, (hipStream_t)s, "leaky_relu_bwd");
}

int wfae_pad2d(const float* x, float* y, int64_t planes, int H, int W, int pad, int crop, wfae_stream_t stream) {
  WFAE_REQUIRE(x && y, WFAE_ERR_NULL_POINTER, "pad2d: null pointer");
  WFAE_REQUIRE(planes > 0 && H > 0 && W > 0 && pad >= 0, WFAE_ERR_BAD_SHAPE, "pad2d: bad shape");
  const long total = crop ? planes * H * W : planes * (H + 2 * pad) * (W + 2 * pad);
  if (crop)
    hipLaunchKernelGGL((pad2d_kernel<1>), dim3(grid_1d(total, 1)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, pad, total);
  else
    hipLaunchKernelGGL((pad2d_kernel<0>), dim3(grid_1d(total, 1)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, pad, total);
  return check_launch("pad2d");
}

int wfae_mean_fwd(const float* x, float* out, int64_t n, int hinge, float sign, float weight, void* ws, size_t ws_bytes,
                  wfae_stream_t stream) {
  WFAE_REQUIRE(x && out, WFAE_ERR_NULL_POINTER, "mean_fwd: null pointer");
  WFAE_REQUIRE(n > 0, WFAE_ERR_BAD_SHAPE, "mean_fwd: bad size");
  int blocks = grid_1d(n, 16);
  if (blocks > 1024) blocks = 1024;
  WFAE_REQUIRE(ws && ws_bytes >= (size_t)blocks * sizeof(double), WFAE_ERR_WORKSPACE, "mean_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (hinge)
    hipLaunchKernelGGL((mean_part_kernel<1>), dim3(blocks), dim3(RT), 0, st, x, (double*)ws, (long)n, sign);
  else
    hipLaunchKernelGGL((mean_part_kernel<0>), dim3(blocks), dim3(RT), 0, st, x, (double*)ws, (long)n, sign);
  int rc = check_launch("mean_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(scalar_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, blocks,
                     (double)weight / (double)n, out, (double*)nullptr);
  return check_launch("mean_finalize");
}

int wfae_mean_bwd(const float* x, const float* gout, float* dx, int64_t n, int hinge, float sign, float weight,
                  wfae_stream_t stream) {
  WFAE_REQUIRE(x && gout && dx, WFAE_ERR_NULL_POINTER, "mean_bwd: null pointer");
  WFAE_REQUIRE(n > 0, WFAE_ERR_BAD_SHAPE, "mean_bwd: bad size");
  const float w = (float)((double)weight / (double)n);
  if (hinge)
    hipLaunchKernelGGL((mean_bwd_kernel<1>), dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, x, gout, w, sign, dx, (long)n);
  else
    hipLaunchKernelGGL((mean_bwd_kernel<0>), dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, x, gout, w, sign, dx, (long)n);
  return check_launch("mean_bwd");
}

int wfae_scale(const float* x, const float* scale_dev, float scale, float* y, int64_t n, wfae_stream_t stream) {
  WFAE_REQUIRE(x && y, WFAE_ERR_NULL_POINTER, "scale: null pointer");
  if (n <= 0) return WFAE_OK;
  hipLaunchKernelGGL(scale_kernel, dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, x, scale_dev, scale, y, (long)n);
  return check_launch("scale");
}

int wfae_adaptive_weight(const double* sumsq_rec, const double* sumsq_disc, float disc_weight, float* out,
                         wfae_stream_t stream) {
  WFAE_REQUIRE(sumsq_rec && sumsq_disc && out, WFAE_ERR_NULL_POINTER, "adaptive_weight: null pointer");
  hipLaunchKernelGGL(adaptive_weight_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sumsq_rec, sumsq_disc,
                     disc_weight, out);
  return check_launch("adaptive_weight");
}

int wfae_clip_coef(const double* sumsq_parts, int n_parts, float max_norm, float pre_scale, float* out2,
                   wfae_stream_t stream) {
  WFAE_REQUIRE(sumsq_parts && out2, WFAE_ERR_NULL_POINTER, "clip_coef: null pointer");
  WFAE_REQUIRE(n_parts > 0 && max_norm > 0.f, WFAE_ERR_BAD_SHAPE, "clip_coef: bad arguments");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sumsq_parts, n_parts, max_norm,
                     pre_scale, out2);
  return check_launch("clip_coef");
}

}  // extern "C"
