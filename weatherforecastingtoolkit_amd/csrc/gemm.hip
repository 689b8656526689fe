// gemm.hip — fp32 MFMA (v_mfma_f32_32x32x2_f32) tiled GEMM core with NCHW-aware
// operand loaders.  One kernel template serves
//   * 1x1 convolutions fwd / bwd-data / bwd-weight   (ae_64x8x8_lin.py:15,19,69,79)
//   * nn.Linear fwd / bwd-data / bwd-weight          (ae_64x8x8_lin.py:74-75)
//   * 4x4 stride-2 Conv2d / ConvTranspose2d in their three roles (down, up,
//     wgrad) as IMPLICIT GEMMs: the B operand is gathered straight from the
//     NCHW activation tensor into LDS — no im2col buffer ever exists in HBM.
//
// Numerics: v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fp32 fmaf chain
// (MI355X guide §3 "FP32-input MFMA"), so results are deterministic and match
// an fp32 reference to accumulation-order rounding.
//
// Tile: BM x 128 x 16, 256 threads = 4 waves, LDS images As[k][m], Bs[k][n]
// (m / n contiguous => conflict-free ds_read_b32 for both MFMA operands),
// double-buffered through registers (global loads of stage t+1 are in flight
// while stage t is multiplied).
#include "common.h"

using namespace wfae;

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 16;
constexpr int BN = 128;
constexpr int NT = 256;

enum AKind { A_KCONTIG = 0, A_MCONTIG = 1 };
enum BKind { B_NCONTIG = 0, B_KCONTIG = 1, B_DOWN = 2, B_UP = 3, B_WGRAD = 4 };
enum EKind { E_BATCHED = 0, E_SLAB = 1, E_UP = 2 };

struct GemmP {
  const float* A;
  const float* B;
  float* C;
  const float* bias;  // [M] or null
  const float* res;   // residual, E_BATCHED only
  int M, N, K;
  int k_per_split;  // multiple of BK; == K rounded up when not split
  // A(m,k): KCONTIG  A[(k / a_hw) * a_img + m * a_ld + k % a_hw]
  //         MCONTIG  A[k * a_ld + m]
  int a_hw;
  long a_img;
  int a_ld;
  // B(k,n): NCONTIG  B[(n / b_hw) * b_img + k * b_ld + n % b_hw]
  //         KCONTIG  B[(k / b_hw) * b_img + n * b_ld + k % b_hw]
  int b_hw;
  long b_img;
  int b_ld;
  // C(m,n): BATCHED  C[(n / c_hw) * c_img + m * c_ld + n % c_hw]
  int c_hw;
  long c_img;
  int c_ld;
  long res_img;
  int beta;          // 1: C += result
  int a_vec, b_vec;  // 16-byte vector loads are legal for this operand
  // 4x4 s2 geometry (gather kinds): lo side Hlo x Wlo, hi side 2Hlo x 2Wlo
  int Chi, Clo, Hlo, Wlo;
};

template <int BM, int WMW, int WNW, int AK, int BKD, int EK>
__global__ __launch_bounds__(NT) void gemm_kernel(GemmP p) {
  static_assert(WMW * WNW == 4, "4 waves");
  constexpr int TM = BM / (WMW * 32);
  constexpr int TN = BN / (WNW * 32);
  static_assert(TM >= 1 && TN >= 1, "tile");
  constexpr int LDA_S = BM + 4;
  constexpr int LDB_S = BN + 4;
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDA_S];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB_S];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int mtiles = (p.M + BM - 1) / BM;
  const int m0 = (blockIdx.x % mtiles) * BM;
  const int n0 = (blockIdx.x / mtiles) * BN;
  const int z = blockIdx.z;

  int k_begin = 0, k_end = p.K;
  int py = 0, px = 0;
  const float* __restrict__ Ap = p.A;
  if constexpr (BKD == B_UP) {
    py = z >> 1;
    px = z & 1;
    Ap += (long)z * p.K * p.M;  // packed per-phase weights [phase][k][m]
  } else {
    k_begin = z * p.k_per_split;
    k_end = min(p.K, k_begin + p.k_per_split);
  }
  const float* __restrict__ Bp = p.B;

  // ------------------------------------------------ per-thread loader state
  constexpr int A_IT = (BM * BK / 4 + NT - 1) / NT;
  constexpr int B_IT = (BN * BK / 4) / NT;  // 2
  float4 ra[A_IT];
  float4 rb[B_IT];
  float rg[8];  // gather kinds

  const int H = 2 * p.Hlo, W = 2 * p.Wlo;
  const int HWlo = p.Hlo * p.Wlo;

  // B_NCONTIG: n is fixed per thread
  long bn_base = 0;
  bool bn_ok = false;
  // gather kinds
  const int g_nl = t & (BN - 1);
  const int g_kh = t >> 7;
  long g_base = 0;
  unsigned g_rmask = 0, g_cmask = 0;
  bool g_ok = false;
  int g_hi = 0, g_ky = 0, g_kx = 0;

  if constexpr (BKD == B_NCONTIG) {
    const int n = n0 + (t & 31) * 4;
    bn_ok = n < p.N;
    if (bn_ok) {
      const int img = n / p.b_hw;
      bn_base = (long)img * p.b_img + (n - img * p.b_hw);
    }
  } else if constexpr (BKD == B_DOWN) {
    const int n = n0 + g_nl;
    g_ok = n < p.N;
    if (g_ok) {
      const int img = n / HWlo;
      const int r = n - img * HWlo;
      const int oy = r / p.Wlo, ox = r - oy * p.Wlo;
      const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
      g_base = (long)img * p.Chi * H * W + (long)iy0 * W + ix0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int iy = iy0 + g_kh * 2 + i;
        if (iy >= 0 && iy < H) g_rmask |= 1u << i;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ix = ix0 + i;
        if (ix >= 0 && ix < W) g_cmask |= 1u << i;
      }
    }
  } else if constexpr (BKD == B_UP) {
    const int n = n0 + g_nl;
    g_ok = n < p.N;
    if (g_ok) {
      const int img = n / HWlo;
      const int r = n - img * HWlo;
      const int a = r / p.Wlo, b = r - a * p.Wlo;
      g_base = (long)img * p.Clo * HWlo + (long)(a + py) * p.Wlo + (b + px);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = a + py - i, col = b + px - i;
        if (row >= 0 && row < p.Hlo) g_rmask |= 1u << i;
        if (col >= 0 && col < p.Wlo) g_cmask |= 1u << i;
      }
    }
  } else if constexpr (BKD == B_WGRAD) {
    const int n = n0 + g_nl;
    g_ok = n < p.N;
    g_hi = n >> 4;
    g_ky = (n >> 2) & 3;
    g_kx = n & 3;
  }

  auto load_a = [&](int k0) {
    if constexpr (AK == A_KCONTIG) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int idx = t + i * NT;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < BM * BK / 4) {
          const int m = m0 + (idx >> 2);
          const int k = k0 + (idx & 3) * 4;
          if (m < p.M && k < k_end) {
            if (p.a_vec && k + 3 < k_end) {
              const int img = k / p.a_hw;
              v = *reinterpret_cast<const float4*>(Ap + (long)img * p.a_img + (long)m * p.a_ld +
                                                   (k - img * p.a_hw));
            } else {
              float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int kk = k + j;
                if (kk < k_end) {
                  const int img = kk / p.a_hw;
                  e[j] = Ap[(long)img * p.a_img + (long)m * p.a_ld + (kk - img * p.a_hw)];
                }
              }
              v = make_float4(e[0], e[1], e[2], e[3]);
            }
          }
        }
        ra[i] = v;
      }
    } else {  // A_MCONTIG
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int idx = t + i * NT;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < BM * BK / 4) {
          const int kr = idx / (BM / 4);
          const int m = m0 + (idx % (BM / 4)) * 4;
          const int k = k0 + kr;
          if (k < k_end && m < p.M) {
            const float* src = Ap + (long)k * p.a_ld + m;
            if (p.a_vec && m + 3 < p.M) {
              v = *reinterpret_cast<const float4*>(src);
            } else {
              v.x = src[0];
              if (m + 1 < p.M) v.y = src[1];
              if (m + 2 < p.M) v.z = src[2];
              if (m + 3 < p.M) v.w = src[3];
            }
          }
        }
        ra[i] = v;
      }
    }
  };

  auto store_a = [&](int buf) {
    if constexpr (AK == A_KCONTIG) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int idx = t + i * NT;
        if (idx < BM * BK / 4) {
          const int ml = idx >> 2, kq = (idx & 3) * 4;
          As[buf][kq + 0][ml] = ra[i].x;
          As[buf][kq + 1][ml] = ra[i].y;
          As[buf][kq + 2][ml] = ra[i].z;
          As[buf][kq + 3][ml] = ra[i].w;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int idx = t + i * NT;
        if (idx < BM * BK / 4) {
          const int kr = idx / (BM / 4), ml = (idx % (BM / 4)) * 4;
          *reinterpret_cast<float4*>(&As[buf][kr][ml]) = ra[i];
        }
      }
    }
  };

  auto load_b = [&](int k0) {
    if constexpr (BKD == B_NCONTIG) {
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int k = k0 + (t >> 5) + i * 8;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bn_ok && k < k_end) {
          if (p.b_vec) {
            v = *reinterpret_cast<const float4*>(Bp + bn_base + (long)k * p.b_ld);
          } else {
            const int n = n0 + (t & 31) * 4;
            float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int nn = n + j;
              if (nn < p.N) {
                const int img = nn / p.b_hw;
                e[j] = Bp[(long)img * p.b_img + (long)k * p.b_ld + (nn - img * p.b_hw)];
              }
            }
            v = make_float4(e[0], e[1], e[2], e[3]);
          }
        }
        rb[i] = v;
      }
    } else if constexpr (BKD == B_KCONTIG) {
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int idx = t + i * NT;
        const int n = n0 + (idx >> 2);
        const int k = k0 + (idx & 3) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < p.N && k < k_end) {
          if (p.b_vec && k + 3 < k_end) {
            const int img = k / p.b_hw;
            v = *reinterpret_cast<const float4*>(Bp + (long)img * p.b_img + (long)n * p.b_ld +
                                                 (k - img * p.b_hw));
          } else {
            float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int kk = k + j;
              if (kk < k_end) {
                const int img = kk / p.b_hw;
                e[j] = Bp[(long)img * p.b_img + (long)n * p.b_ld + (kk - img * p.b_hw)];
              }
            }
            v = make_float4(e[0], e[1], e[2], e[3]);
          }
        }
        rb[i] = v;
      }
    } else if constexpr (BKD == B_DOWN) {
      // k = hi*16 + ky*4 + kx ; one hi channel per stage
      const int hi = k0 >> 4;
      const float* src = Bp + g_base + (long)hi * H * W + (long)(g_kh * 2) * W;
      const bool ok = g_ok && hi < p.Chi;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = 0.f;
          if (ok && ((g_rmask >> i) & 1u) && ((g_cmask >> j) & 1u)) v = src[i * W + j];
          rg[i * 4 + j] = v;
        }
    } else if constexpr (BKD == B_UP) {
      // k = lo*4 + ty*2 + tx ; four lo channels per stage, this thread two of them
      const int lo0 = (k0 >> 2) + g_kh * 2;
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const int lo = lo0 + l;
        const float* src = Bp + g_base + (long)lo * HWlo;
        const bool ok = g_ok && lo < p.Clo;
#pragma unroll
        for (int ty = 0; ty < 2; ++ty)
#pragma unroll
          for (int tx = 0; tx < 2; ++tx) {
            float v = 0.f;
            if (ok && ((g_rmask >> ty) & 1u) && ((g_cmask >> tx) & 1u)) v = src[-ty * p.Wlo - tx];
            rg[l * 4 + ty * 2 + tx] = v;
          }
      }
    } else {  // B_WGRAD: k = (img, oy, ox) pixel index, n = (hi, ky, kx)
      const int kb = k0 + g_kh * 8;
      if ((p.Wlo & 7) == 0) {
        // 8 consecutive pixels share (img, oy)
        const int img = kb / HWlo;
        const int r = kb - img * HWlo;
        const int oy = r / p.Wlo, ox0 = r - oy * p.Wlo;
        const int iy = 2 * oy - 1 + g_ky;
        const bool ok = g_ok && kb < k_end && iy >= 0 && iy < H;
        const int ixb = 2 * ox0 - 1 + g_kx;
        const float* src = Bp + ((long)img * p.Chi + g_hi) * H * W + (long)iy * W + ixb;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ix = ixb + 2 * j;
          float v = 0.f;
          if (ok && ix >= 0 && ix < W) v = src[2 * j];
          rg[j] = v;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = kb + j;
          float v = 0.f;
          if (g_ok && k < k_end) {
            const int img = k / HWlo;
            const int r = k - img * HWlo;
            const int oy = r / p.Wlo, ox = r - oy * p.Wlo;
            const int iy = 2 * oy - 1 + g_ky, ix = 2 * ox - 1 + g_kx;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W)
              v = Bp[((long)img * p.Chi + g_hi) * H * W + (long)iy * W + ix];
          }
          rg[j] = v;
        }
      }
    }
  };

  auto store_b = [&](int buf) {
    if constexpr (BKD == B_NCONTIG) {
#pragma unroll
      for (int i = 0; i < B_IT; ++i)
        *reinterpret_cast<float4*>(&Bs[buf][(t >> 5) + i * 8][(t & 31) * 4]) = rb[i];
    } else if constexpr (BKD == B_KCONTIG) {
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int idx = t + i * NT;
        const int nl = idx >> 2, kq = (idx & 3) * 4;
        Bs[buf][kq + 0][nl] = rb[i].x;
        Bs[buf][kq + 1][nl] = rb[i].y;
        Bs[buf][kq + 2][nl] = rb[i].z;
        Bs[buf][kq + 3][nl] = rb[i].w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) Bs[buf][g_kh * 8 + j][g_nl] = rg[j];
    }
  };

  // ------------------------------------------------------------- main loop
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wm0 = (wave / WNW) * (TM * 32);
  const int wn0 = (wave % WNW) * (TN * 32);
  const int l31 = lane & 31, lh = lane >> 5;

  const int nstages = (k_end - k_begin + BK - 1) / BK;
  if (nstages > 0) {
    load_a(k_begin);
    load_b(k_begin);
    store_a(0);
    store_b(0);
  }
  __syncthreads();
  for (int s = 0; s < nstages; ++s) {
    const int buf = s & 1;
    if (s + 1 < nstages) {
      load_a(k_begin + (s + 1) * BK);
      load_b(k_begin + (s + 1) * BK);
    }
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
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < nstages) {
      store_a(buf ^ 1);
      store_b(buf ^ 1);
    }
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue
  // C/D map of 32x32 MFMA: col(n) = lane & 31, row(m) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn0 + j * 32 + l31;
    if (n >= p.N) continue;
    float* cb;
    const float* rbp = nullptr;
    long mstride;
    if constexpr (EK == E_BATCHED) {
      const int img = n / p.c_hw;
      const int pn = n - img * p.c_hw;
      cb = p.C + (long)img * p.c_img + pn;
      if (p.res) rbp = p.res + (long)img * p.res_img + pn;
      mstride = p.c_ld;
    } else if constexpr (EK == E_SLAB) {
      cb = p.C + (long)z * p.M * p.N + n;
      mstride = p.N;
    } else {  // E_UP
      const int img = n / HWlo;
      const int r = n - img * HWlo;
      const int a = r / p.Wlo, b = r - a * p.Wlo;
      cb = p.C + (long)img * p.Chi * H * W + (long)(2 * a + py) * W + (2 * b + px);
      mstride = (long)H * W;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < p.M) {
          float v = acc[i][j][r];
          if constexpr (EK == E_BATCHED) {
            if (p.bias) v += p.bias[m];
            if (rbp) v += rbp[(long)m * mstride];
            if (p.beta) v += cb[(long)m * mstride];
          }
          cb[(long)m * mstride] = v;
        }
      }
  }
}

// out[i] = (beta ? out[i] : 0) + sum_z slab[z][i] (+ bias_n[i % N])   (fixed order => deterministic)
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                     const float* __restrict__ bias_n, long MN, int N, int splits,
                                     int beta) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= MN) return;
  float s = 0.f;
  for (int z = 0; z < splits; ++z) s += slab[(long)z * MN + i];
  if (bias_n) s += bias_n[i % N];
  if (beta) s += out[i];
  out[i] = s;
}

// Wp[phase][lo*4 + ty*2 + tx][hi] = W[lo][hi][1-py+2ty][1-px+2tx]
__global__ void pack_up_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Clo,
                                       int Chi) {
  const long total = (long)16 * Clo * Chi;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int hi = (int)(i % Chi);
  long r = i / Chi;
  const int tx = (int)(r & 1);
  const int ty = (int)((r >> 1) & 1);
  r >>= 2;
  const int lo = (int)(r % Clo);
  const int phase = (int)(r / Clo);
  const int py = phase >> 1, px = phase & 1;
  const int ky = 1 - py + 2 * ty, kx = 1 - px + 2 * tx;
  wp[i] = w[((long)lo * Chi + hi) * 16 + ky * 4 + kx];
}

template <int AK, int BKD, int EK>
int launch_gemm(const GemmP& p, int zdim, hipStream_t st, const char* what) {
  const int ntiles = cdiv(p.N, BN);
  dim3 block(NT);
  if (p.M > 64) {
    dim3 grid(cdiv(p.M, 128) * ntiles, 1, zdim);
    hipLaunchKernelGGL((gemm_kernel<128, 2, 2, AK, BKD, EK>), grid, block, 0, st, p);
  } else if (p.M > 32) {
    dim3 grid(cdiv(p.M, 64) * ntiles, 1, zdim);
    hipLaunchKernelGGL((gemm_kernel<64, 2, 2, AK, BKD, EK>), grid, block, 0, st, p);
  } else {
    dim3 grid(ntiles, 1, zdim);
    hipLaunchKernelGGL((gemm_kernel<32, 1, 4, AK, BKD, EK>), grid, block, 0, st, p);
  }
  return check_launch(what);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// choose split count so tiles*splits covers the chip a few times over
inline int pick_splits(int M, int N, int K, size_t ws_bytes, int* k_per_split) {
  const long tiles = (long)cdiv(M, M > 64 ? 128 : (M > 32 ? 64 : 32)) * cdiv(N, BN);
  const int stages = cdiv(K, BK);
  long want = (2048 + tiles - 1) / tiles;
  if (want < 1) want = 1;
  if (want > stages) want = stages;
  const size_t slab = (size_t)M * N * sizeof(float);
  while (want > 1 && (size_t)want * slab > ws_bytes) --want;
  int per = cdiv(stages, (int)want) * BK;
  *k_per_split = per;
  return cdiv(K, per);
}

int splitk_finish(const float* slab, float* out, const float* bias_n, long MN, int N, int splits,
                  int beta, hipStream_t st) {
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(MN, 256)), dim3(256), 0, st, slab, out, bias_n,
                     MN, N, splits, beta);
  return check_launch("splitk_reduce");
}

}  // namespace

extern "C" {

int wfae_conv1x1_fwd(const float* x, const float* w, const float* bias, const float* res,
                     int64_t res_img_stride, float* y, int NB, int Cin, int Cout, int HW,
                     wfae_stream_t stream) {
  WFAE_REQUIRE(x && w && y, WFAE_ERR_NULL_POINTER, "conv1x1_fwd: null pointer");
  WFAE_REQUIRE(NB > 0 && Cin > 0 && Cout > 0 && HW > 0, WFAE_ERR_BAD_SHAPE, "conv1x1_fwd: bad shape");
  WFAE_REQUIRE((int64_t)NB * HW < (1ll << 31), WFAE_ERR_BAD_SHAPE, "conv1x1_fwd: NB*HW too large");
  GemmP p = {};
  p.A = w; p.B = x; p.C = y; p.bias = bias; p.res = res;
  p.M = Cout; p.N = NB * HW; p.K = Cin; p.k_per_split = cdiv(Cin, BK) * BK;
  p.a_hw = Cin; p.a_img = 0; p.a_ld = Cin;
  p.b_hw = HW; p.b_img = (long)Cin * HW; p.b_ld = HW;
  p.c_hw = HW; p.c_img = (long)Cout * HW; p.c_ld = HW; p.res_img = res_img_stride;
  p.a_vec = (Cin % 4 == 0) && aligned16(w);
  p.b_vec = (HW % 4 == 0) && aligned16(x);
  return launch_gemm<A_KCONTIG, B_NCONTIG, E_BATCHED>(p, 1, (hipStream_t)stream, "conv1x1_fwd");
}

int wfae_conv1x1_bwd_data(const float* dy, const float* w, float* dx, int NB, int Cin, int Cout,
                          int HW, wfae_stream_t stream) {
  WFAE_REQUIRE(dy && w && dx, WFAE_ERR_NULL_POINTER, "conv1x1_bwd_data: null pointer");
  WFAE_REQUIRE(NB > 0 && Cin > 0 && Cout > 0 && HW > 0, WFAE_ERR_BAD_SHAPE, "conv1x1_bwd_data: bad shape");
  WFAE_REQUIRE((int64_t)NB * HW < (1ll << 31), WFAE_ERR_BAD_SHAPE, "conv1x1_bwd_data: NB*HW too large");
  GemmP p = {};
  p.A = w; p.B = dy; p.C = dx;
  p.M = Cin; p.N = NB * HW; p.K = Cout; p.k_per_split = cdiv(Cout, BK) * BK;
  p.a_ld = Cin;  // A(m=ci,k=co) = w[co*Cin + ci]
  p.b_hw = HW; p.b_img = (long)Cout * HW; p.b_ld = HW;
  p.c_hw = HW; p.c_img = (long)Cin * HW; p.c_ld = HW;
  p.a_vec = (Cin % 4 == 0) && aligned16(w);
  p.b_vec = (HW % 4 == 0) && aligned16(dy);
  return launch_gemm<A_MCONTIG, B_NCONTIG, E_BATCHED>(p, 1, (hipStream_t)stream, "conv1x1_bwd_data");
}

int wfae_conv1x1_bwd_weight(const float* dy, const float* x, float* dw, int NB, int Cin, int Cout,
                            int HW, int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(dy && x && dw, WFAE_ERR_NULL_POINTER, "conv1x1_bwd_weight: null pointer");
  WFAE_REQUIRE(NB > 0 && Cin > 0 && Cout > 0 && HW > 0, WFAE_ERR_BAD_SHAPE, "conv1x1_bwd_weight: bad shape");
  WFAE_REQUIRE((int64_t)NB * HW < (1ll << 31), WFAE_ERR_BAD_SHAPE, "conv1x1_bwd_weight: NB*HW too large");
  const size_t slab = (size_t)Cout * Cin * sizeof(float);
  WFAE_REQUIRE(ws && ws_bytes >= slab, WFAE_ERR_WORKSPACE, "conv1x1_bwd_weight: workspace %zu < %zu", ws_bytes, slab);
  GemmP p = {};
  p.A = dy; p.B = x; p.C = (float*)ws;
  p.M = Cout; p.N = Cin; p.K = NB * HW;
  p.a_hw = HW; p.a_img = (long)Cout * HW; p.a_ld = HW;
  p.b_hw = HW; p.b_img = (long)Cin * HW; p.b_ld = HW;
  p.a_vec = (HW % 4 == 0) && aligned16(dy);
  p.b_vec = (HW % 4 == 0) && aligned16(x);
  const int splits = pick_splits(p.M, p.N, p.K, ws_bytes, &p.k_per_split);
  int rc = launch_gemm<A_KCONTIG, B_KCONTIG, E_SLAB>(p, splits, (hipStream_t)stream, "conv1x1_bwd_weight");
  if (rc) return rc;
  return splitk_finish((float*)ws, dw, nullptr, (long)Cout * Cin, Cin, splits, accumulate, (hipStream_t)stream);
}

int wfae_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int In,
                    int Out, void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(x && w && y, WFAE_ERR_NULL_POINTER, "linear_fwd: null pointer");
  WFAE_REQUIRE(B > 0 && In > 0 && Out > 0, WFAE_ERR_BAD_SHAPE, "linear_fwd: bad shape");
  const size_t slab = (size_t)B * Out * sizeof(float);
  WFAE_REQUIRE(ws && ws_bytes >= slab, WFAE_ERR_WORKSPACE, "linear_fwd: workspace %zu < %zu", ws_bytes, slab);
  GemmP p = {};  // C(m=b, n=o) = sum_i x[b,i] w[o,i]
  p.A = x; p.B = w; p.C = (float*)ws;
  p.M = B; p.N = Out; p.K = In;
  p.a_hw = In; p.a_img = 0; p.a_ld = In;
  p.b_hw = In; p.b_img = 0; p.b_ld = In;
  p.a_vec = (In % 4 == 0) && aligned16(x);
  p.b_vec = (In % 4 == 0) && aligned16(w);
  const int splits = pick_splits(p.M, p.N, p.K, ws_bytes, &p.k_per_split);
  int rc = launch_gemm<A_KCONTIG, B_KCONTIG, E_SLAB>(p, splits, (hipStream_t)stream, "linear_fwd");
  if (rc) return rc;
  return splitk_finish((float*)ws, y, bias, (long)B * Out, Out, splits, 0, (hipStream_t)stream);
}

int wfae_linear_bwd_data(const float* dy, const float* w, float* dx, int B, int In, int Out,
                         wfae_stream_t stream) {
  WFAE_REQUIRE(dy && w && dx, WFAE_ERR_NULL_POINTER, "linear_bwd_data: null pointer");
  WFAE_REQUIRE(B > 0 && In > 0 && Out > 0, WFAE_ERR_BAD_SHAPE, "linear_bwd_data: bad shape");
  GemmP p = {};  // dx(m=b, n=i) = sum_o dy[b,o] w[o,i]
  p.A = dy; p.B = w; p.C = dx;
  p.M = B; p.N = In; p.K = Out; p.k_per_split = cdiv(Out, BK) * BK;
  p.a_hw = Out; p.a_img = 0; p.a_ld = Out;
  p.b_hw = In; p.b_img = 0; p.b_ld = In;
  p.c_hw = In; p.c_img = 0; p.c_ld = In;
  p.a_vec = (Out % 4 == 0) && aligned16(dy);
  p.b_vec = (In % 4 == 0) && aligned16(w);
  return launch_gemm<A_KCONTIG, B_NCONTIG, E_BATCHED>(p, 1, (hipStream_t)stream, "linear_bwd_data");
}

int wfae_linear_bwd_weight(const float* dy, const float* x, float* dw, int B, int In, int Out,
                           int accumulate, wfae_stream_t stream) {
  WFAE_REQUIRE(dy && x && dw, WFAE_ERR_NULL_POINTER, "linear_bwd_weight: null pointer");
  WFAE_REQUIRE(B > 0 && In > 0 && Out > 0, WFAE_ERR_BAD_SHAPE, "linear_bwd_weight: bad shape");
  GemmP p = {};  // dw(m=o, n=i) = sum_b dy[b,o] x[b,i]
  p.A = dy; p.B = x; p.C = dw;
  p.M = Out; p.N = In; p.K = B; p.k_per_split = cdiv(B, BK) * BK;
  p.a_ld = Out;
  p.b_hw = In; p.b_img = 0; p.b_ld = In;
  p.c_hw = In; p.c_img = 0; p.c_ld = In;
  p.beta = accumulate ? 1 : 0;
  p.a_vec = (Out % 4 == 0) && aligned16(dy);
  p.b_vec = (In % 4 == 0) && aligned16(x);
  return launch_gemm<A_MCONTIG, B_NCONTIG, E_BATCHED>(p, 1, (hipStream_t)stream, "linear_bwd_weight");
}

int wfae_conv4x4s2_down(const float* hi, const float* w, float* lo, int NB, int Chi, int Clo,
                        int Hlo, int Wlo, wfae_stream_t stream) {
  WFAE_REQUIRE(hi && w && lo, WFAE_ERR_NULL_POINTER, "conv4x4s2_down: null pointer");
  WFAE_REQUIRE(NB > 0 && Chi > 0 && Clo > 0 && Hlo > 0 && Wlo > 0, WFAE_ERR_BAD_SHAPE, "conv4x4s2_down: bad shape");
  WFAE_REQUIRE((int64_t)NB * Hlo * Wlo < (1ll << 31) && (int64_t)Chi * 16 < (1ll << 31), WFAE_ERR_BAD_SHAPE,
               "conv4x4s2_down: too large");
  GemmP p = {};
  p.A = w; p.B = hi; p.C = lo;
  p.M = Clo; p.N = NB * Hlo * Wlo; p.K = Chi * 16; p.k_per_split = p.K;
  p.a_hw = p.K; p.a_img = 0; p.a_ld = p.K;
  p.c_hw = Hlo * Wlo; p.c_img = (long)Clo * Hlo * Wlo; p.c_ld = Hlo * Wlo;
  p.a_vec = aligned16(w);
  p.Chi = Chi; p.Clo = Clo; p.Hlo = Hlo; p.Wlo = Wlo;
  return launch_gemm<A_KCONTIG, B_DOWN, E_BATCHED>(p, 1, (hipStream_t)stream, "conv4x4s2_down");
}

int wfae_conv4x4s2_up(const float* lo, const float* w, float* hi, int NB, int Chi, int Clo, int Hlo,
                      int Wlo, void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(lo && w && hi, WFAE_ERR_NULL_POINTER, "conv4x4s2_up: null pointer");
  WFAE_REQUIRE(NB > 0 && Chi > 0 && Clo > 0 && Hlo > 0 && Wlo > 0, WFAE_ERR_BAD_SHAPE, "conv4x4s2_up: bad shape");
  WFAE_REQUIRE((int64_t)NB * Hlo * Wlo < (1ll << 31), WFAE_ERR_BAD_SHAPE, "conv4x4s2_up: too large");
  const size_t need = (size_t)16 * Clo * Chi * sizeof(float);
  WFAE_REQUIRE(ws && ws_bytes >= need, WFAE_ERR_WORKSPACE, "conv4x4s2_up: workspace %zu < %zu", ws_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(pack_up_weights_kernel, dim3(cdiv((long)16 * Clo * Chi, 256)), dim3(256), 0, st, w,
                     (float*)ws, Clo, Chi);
  int rc = check_launch("pack_up_weights");
  if (rc) return rc;
  GemmP p = {};
  p.A = (const float*)ws; p.B = lo; p.C = hi;
  p.M = Chi; p.N = NB * Hlo * Wlo; p.K = Clo * 4; p.k_per_split = p.K;
  p.a_ld = Chi;
  p.a_vec = (Chi % 4 == 0) && aligned16(ws);
  p.Chi = Chi; p.Clo = Clo; p.Hlo = Hlo; p.Wlo = Wlo;
  return launch_gemm<A_MCONTIG, B_UP, E_UP>(p, 4, st, "conv4x4s2_up");
}

int wfae_conv4x4s2_wgrad(const float* lo, const float* hi, float* dw, int NB, int Chi, int Clo,
                         int Hlo, int Wlo, int accumulate, void* ws, size_t ws_bytes,
                         wfae_stream_t stream) {
  WFAE_REQUIRE(lo && hi && dw, WFAE_ERR_NULL_POINTER, "conv4x4s2_wgrad: null pointer");
  WFAE_REQUIRE(NB > 0 && Chi > 0 && Clo > 0 && Hlo > 0 && Wlo > 0, WFAE_ERR_BAD_SHAPE, "conv4x4s2_wgrad: bad shape");
  WFAE_REQUIRE((int64_t)NB * Hlo * Wlo < (1ll << 31), WFAE_ERR_BAD_SHAPE, "conv4x4s2_wgrad: too large");
  const size_t slab = (size_t)Clo * Chi * 16 * sizeof(float);
  WFAE_REQUIRE(ws && ws_bytes >= slab, WFAE_ERR_WORKSPACE, "conv4x4s2_wgrad: workspace %zu < %zu", ws_bytes, slab);
  const int HWlo = Hlo * Wlo;
  GemmP p = {};
  p.A = lo; p.B = hi; p.C = (float*)ws;
  p.M = Clo; p.N = Chi * 16; p.K = NB * HWlo;
  p.a_hw = HWlo; p.a_img = (long)Clo * HWlo; p.a_ld = HWlo;
  p.a_vec = (HWlo % 4 == 0) && aligned16(lo);
  p.Chi = Chi; p.Clo = Clo; p.Hlo = Hlo; p.Wlo = Wlo;
  const int splits = pick_splits(p.M, p.N, p.K, ws_bytes, &p.k_per_split);
  int rc = launch_gemm<A_KCONTIG, B_WGRAD, E_SLAB>(p, splits, (hipStream_t)stream, "conv4x4s2_wgrad");
  if (rc) return rc;
  return splitk_finish((float*)ws, dw, nullptr, (long)Clo * Chi * 16, Chi * 16, splits, accumulate,
                       (hipStream_t)stream);
}

}  // extern "C"
