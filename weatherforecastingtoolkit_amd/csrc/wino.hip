// wino.hip — Winograd F(2x2, 2x2) transforms for the 4x4 stride-2 convolutions.
//
// A 4x4 stride-2 pad-1 convolution is the sum of four 2x2 stride-1 "valid" convolutions, one per input
// parity phase:   lo[oy][ox] = sum_{p,q} sum_{a,b} w[2a+p][2b+q] * P_pq[oy+a][ox+b],
// P_pq[i][j] = hi[2i-1+p][2j-1+q]  (zero outside the image), i in [0,Hlo], j in [0,Wlo].
// Each phase goes through F(2x2,2x2): per 2x2 output tile 9 multiplies instead of 16, i.e. 9 GEMMs
// M_xi = U_xi (Clo x 4Chi) * V_xi (4Chi x T) over T = N*Hlo*Wlo/4 tiles instead of one implicit GEMM with
// K = 16 Chi:  18 instead of 32 FLOP per (lo channel, hi channel, lo pixel).
//   V = B^T d B   (input tile d 3x3 per phase; the four phases' tiles are the 6x6 hi patch at stride 4)
//   U = G g G^T   (g = the phase's 2x2 filter)
//   Y = A^T M A
//   B^T = [1 -1 0; 0 1 0; 0 -1 1]   G = [1 0; 1 1; 0 1]   A^T = [1 1 0; 0 1 1]
// The transposed convolution / data gradient is the exact adjoint  hi = In^T( U^T * Out^T(lo) ), the
// weight gradient  dU_xi = Out^T(dlo)_xi * In(hi)_xi^T,  dw = G^T dU G.
// Layouts: V[xi][c][t], c = (p*2+q)*Chi + h;  M[xi][l][t];  U[xi][l][c];  t = (n*Hlo/2 + ty)*Wlo/2 + tx.
// All kernels are memory-bound streaming kernels: consecutive threads own consecutive tiles tx.
#include "common.h"

using namespace wfae;

namespace {

// hi (N,Chi,2Hlo,2Wlo) -> V[9][4Chi][T]
__global__ __launch_bounds__(256) void wino_in_kernel(const float* __restrict__ hi, float* __restrict__ V, int Chi,
                                                      int Hlo, int Wlo, long T) {
  const int TW = Wlo >> 1, TH = Hlo >> 1, Timg = TW * TH;
  const int tl = blockIdx.x * blockDim.x + threadIdx.x;
  if (tl >= Timg) return;
  const int h = blockIdx.y, n = blockIdx.z;
  const int ty = tl / TW, tx = tl - ty * TW;
  const int H = 2 * Hlo, W = 2 * Wlo;
  const float* __restrict__ src = hi + (long)(n * Chi + h) * H * W;
  float d[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const int Y = 4 * ty - 1 + r;
    const bool rok = Y >= 0 && Y < H;
    const float* __restrict__ row = src + (long)(rok ? Y : 0) * W;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int X = 4 * tx - 1 + c;
      const bool ok = rok && X >= 0 && X < W;
      const float v = row[ok ? X : 0];
      d[r][c] = ok ? v : 0.f;
    }
  }
  const long t = (long)n * Timg + tl;
  const long xi_stride = 4L * Chi * T;
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float tr[3][3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float d0 = d[p][2 * j + q], d1 = d[2 + p][2 * j + q], d2 = d[4 + p][2 * j + q];
        tr[0][j] = d0 - d1;
        tr[1][j] = d1;
        tr[2][j] = d2 - d1;
      }
      float* __restrict__ dst = V + ((long)(p * 2 + q) * Chi + h) * T + t;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        dst[(long)(3 * u + 0) * xi_stride] = tr[u][0] - tr[u][1];
        dst[(long)(3 * u + 1) * xi_stride] = tr[u][1];
        dst[(long)(3 * u + 2) * xi_stride] = tr[u][2] - tr[u][1];
      }
    }
}

// adjoint of wino_in: dV[9][4Chi][T] -> hi (N,Chi,2Hlo,2Wlo); every thread produces one 4x4 block of hi,
// gathering the overlapping contributions of its own and the neighbouring tiles (no atomics).
__global__ __launch_bounds__(256) void wino_in_t_kernel(const float* __restrict__ dV, float* __restrict__ hi, int Chi,
                                                        int Hlo, int Wlo, long T) {
  const int TW = Wlo >> 1, TH = Hlo >> 1, Timg = TW * TH;
  const int tl = blockIdx.x * blockDim.x + threadIdx.x;
  if (tl >= Timg) return;
  const int h = blockIdx.y, n = blockIdx.z;
  const int ty = tl / TW, tx = tl - ty * TW;
  const int W = 2 * Wlo;
  const long xi_stride = 4L * Chi * T;
  // hi row 4ty + r belongs to phase p = (r & 1) ^ 1 and receives tile rows:
  //   r = 0: (ty, i=0), (ty-1, i=2)   r = 1: (ty, 1)   r = 2: (ty, 1)   r = 3: (ty, 2), (ty+1, 0)
  constexpr int NCON[4] = {2, 1, 1, 2};
  constexpr int CDT[4][2] = {{0, -1}, {0, 0}, {0, 0}, {0, 1}};
  constexpr int CI[4][2] = {{0, 2}, {1, 1}, {1, 1}, {2, 0}};
  // B = (B^T)^T rows: e[i][.] = sum_u Bm[i][u] dV[u][.]
  constexpr float Bm[3][3] = {{1.f, 0.f, 0.f}, {-1.f, 1.f, -1.f}, {0.f, 0.f, 1.f}};
  float* __restrict__ out = hi + ((long)(n * Chi + h) * (2 * Hlo) + 4 * ty) * W + 4 * tx;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int p = (r & 1) ^ 1;
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int q = (c & 1) ^ 1;
      float acc = 0.f;
#pragma unroll
      for (int a = 0; a < NCON[r]; ++a) {
        const int tyy = ty + CDT[r][a], i = CI[r][a];
        if (tyy < 0 || tyy >= TH) continue;
#pragma unroll
        for (int b = 0; b < NCON[c]; ++b) {
          const int txx = tx + CDT[c][b], j = CI[c][b];
          if (txx < 0 || txx >= TW) continue;
          const float* __restrict__ base = dV + ((long)(p * 2 + q) * Chi + h) * T + (long)n * Timg + tyy * TW + txx;
#pragma unroll
          for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) {
              const float cf = Bm[i][u] * Bm[j][v];
              if (cf != 0.f) acc += cf * base[(long)(3 * u + v) * xi_stride];
            }
        }
      }
      o[c] = acc;
    }
    *reinterpret_cast<float4*>(out + (long)r * W) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// M[9][Clo][T] -> lo (N,Clo,Hlo,Wlo):  Y = A^T M A
__global__ __launch_bounds__(256) void wino_out_kernel(const float* __restrict__ M, float* __restrict__ lo, int Clo,
                                                       int Hlo, int Wlo, long T) {
  const int TW = Wlo >> 1, TH = Hlo >> 1, Timg = TW * TH;
  const int tl = blockIdx.x * blockDim.x + threadIdx.x;
  if (tl >= Timg) return;
  const int l = blockIdx.y, n = blockIdx.z;
  const int ty = tl / TW, tx = tl - ty * TW;
  const long xi_stride = (long)Clo * T;
  const float* __restrict__ src = M + (long)l * T + (long)n * Timg + tl;
  float m[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) m[k] = src[(long)k * xi_stride];
  float* __restrict__ dst = lo + ((long)(n * Clo + l) * Hlo + 2 * ty) * Wlo + 2 * tx;
  *reinterpret_cast<float2*>(dst) = make_float2((m[0] + m[1]) + (m[3] + m[4]), (m[1] + m[2]) + (m[4] + m[5]));
  *reinterpret_cast<float2*>(dst + Wlo) = make_float2((m[3] + m[4]) + (m[6] + m[7]), (m[4] + m[5]) + (m[7] + m[8]));
}

// adjoint of wino_out: lo (N,Clo,Hlo,Wlo) -> Mt[9][Clo][T]:  Mt = A Y A^T
__global__ __launch_bounds__(256) void wino_out_t_kernel(const float* __restrict__ lo, float* __restrict__ Mt, int Clo,
                                                         int Hlo, int Wlo, long T) {
  const int TW = Wlo >> 1, TH = Hlo >> 1, Timg = TW * TH;
  const int tl = blockIdx.x * blockDim.x + threadIdx.x;
  if (tl >= Timg) return;
  const int l = blockIdx.y, n = blockIdx.z;
  const int ty = tl / TW, tx = tl - ty * TW;
  const float* __restrict__ src = lo + ((long)(n * Clo + l) * Hlo + 2 * ty) * Wlo + 2 * tx;
  const float2 y0 = *reinterpret_cast<const float2*>(src);
  const float2 y1 = *reinterpret_cast<const float2*>(src + Wlo);
  const long xi_stride = (long)Clo * T;
  float* __restrict__ dst = Mt + (long)l * T + (long)n * Timg + tl;
  dst[0 * xi_stride] = y0.x;
  dst[1 * xi_stride] = y0.x + y0.y;
  dst[2 * xi_stride] = y0.y;
  dst[3 * xi_stride] = y0.x + y1.x;
  dst[4 * xi_stride] = (y0.x + y0.y) + (y1.x + y1.y);
  dst[5 * xi_stride] = y0.y + y1.y;
  dst[6 * xi_stride] = y1.x;
  dst[7 * xi_stride] = y1.x + y1.y;
  dst[8 * xi_stride] = y1.y;
}

// w (Clo,Chi,4,4) -> U[9][Clo][4Chi]:  U = G g G^T per phase, g[a][b] = w[2a+p][2b+q]
__global__ __launch_bounds__(256) void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ U, int Clo,
                                                           int Chi) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)Clo * Chi) return;
  const int h = (int)(i % Chi), l = (int)(i / Chi);
  float g[16];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float4 v = reinterpret_cast<const float4*>(w + i * 16)[k];
    g[4 * k] = v.x; g[4 * k + 1] = v.y; g[4 * k + 2] = v.z; g[4 * k + 3] = v.w;
  }
  const long K4 = 4L * Chi, xi_stride = (long)Clo * K4;
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float g00 = g[p * 4 + q], g01 = g[p * 4 + 2 + q], g10 = g[(2 + p) * 4 + q], g11 = g[(2 + p) * 4 + 2 + q];
      float* __restrict__ dst = U + (long)l * K4 + (long)(p * 2 + q) * Chi + h;
      dst[0 * xi_stride] = g00;
      dst[1 * xi_stride] = g00 + g01;
      dst[2 * xi_stride] = g01;
      dst[3 * xi_stride] = g00 + g10;
      dst[4 * xi_stride] = (g00 + g01) + (g10 + g11);
      dst[5 * xi_stride] = g01 + g11;
      dst[6 * xi_stride] = g10;
      dst[7 * xi_stride] = g10 + g11;
      dst[8 * xi_stride] = g11;
    }
}

// dU[9][Clo][4Chi] -> dw (Clo,Chi,4,4) (+= when beta):  dg = G^T dU G per phase
__global__ __launch_bounds__(256) void wino_weights_t_kernel(const float* __restrict__ dU, float* __restrict__ dw,
                                                             int Clo, int Chi, int beta) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)Clo * Chi) return;
  const int h = (int)(i % Chi), l = (int)(i / Chi);
  const long K4 = 4L * Chi, xi_stride = (long)Clo * K4;
  float g[16];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float* __restrict__ src = dU + (long)l * K4 + (long)(p * 2 + q) * Chi + h;
      float m[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) m[k] = src[(long)k * xi_stride];
      g[p * 4 + q] = (m[0] + m[1]) + (m[3] + m[4]);            // a = 0, b = 0
      g[p * 4 + 2 + q] = (m[1] + m[2]) + (m[4] + m[5]);        // a = 0, b = 1
      g[(2 + p) * 4 + q] = (m[3] + m[4]) + (m[6] + m[7]);      // a = 1, b = 0
      g[(2 + p) * 4 + 2 + q] = (m[4] + m[5]) + (m[7] + m[8]);  // a = 1, b = 1
    }
  float4* __restrict__ dst = reinterpret_cast<float4*>(dw + i * 16);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float4 v = make_float4(g[4 * k], g[4 * k + 1], g[4 * k + 2], g[4 * k + 3]);
    if (beta) {
      const float4 o = dst[k];
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    dst[k] = v;
  }
}

inline dim3 tile_grid(int NB, int C, int Hlo, int Wlo) { return dim3(cdiv((long)(Hlo / 2) * (Wlo / 2), 256), C, NB); }

}  // namespace

namespace wfae {

int wino_in(const float* hi, float* V, int NB, int Chi, int Hlo, int Wlo, hipStream_t st) {
  const long T = (long)NB * (Hlo / 2) * (Wlo / 2);
  hipLaunchKernelGGL(wino_in_kernel, tile_grid(NB, Chi, Hlo, Wlo), dim3(256), 0, st, hi, V, Chi, Hlo, Wlo, T);
  return check_launch("wino_in");
}
int wino_in_t(const float* dV, float* hi, int NB, int Chi, int Hlo, int Wlo, hipStream_t st) {
  const long T = (long)NB * (Hlo / 2) * (Wlo / 2);
  hipLaunchKernelGGL(wino_in_t_kernel, tile_grid(NB, Chi, Hlo, Wlo), dim3(256), 0, st, dV, hi, Chi, Hlo, Wlo, T);
  return check_launch("wino_in_t");
}
int wino_out(const float* M, float* lo, int NB, int Clo, int Hlo, int Wlo, hipStream_t st) {
  const long T = (long)NB * (Hlo / 2) * (Wlo / 2);
  hipLaunchKernelGGL(wino_out_kernel, tile_grid(NB, Clo, Hlo, Wlo), dim3(256), 0, st, M, lo, Clo, Hlo, Wlo, T);
  return check_launch("wino_out");
}
int wino_out_t(const float* lo, float* Mt, int NB, int Clo, int Hlo, int Wlo, hipStream_t st) {
  const long T = (long)NB * (Hlo / 2) * (Wlo / 2);
  hipLaunchKernelGGL(wino_out_t_kernel, tile_grid(NB, Clo, Hlo, Wlo), dim3(256), 0, st, lo, Mt, Clo, Hlo, Wlo, T);
  return check_launch("wino_out_t");
}
int wino_weights(const float* w, float* U, int Clo, int Chi, hipStream_t st) {
  hipLaunchKernelGGL(wino_weights_kernel, dim3(cdiv((long)Clo * Chi, 256)), dim3(256), 0, st, w, U, Clo, Chi);
  return check_launch("wino_weights");
}
int wino_weights_t(const float* dU, float* dw, int Clo, int Chi, int beta, hipStream_t st) {
  hipLaunchKernelGGL(wino_weights_t_kernel, dim3(cdiv((long)Clo * Chi, 256)), dim3(256), 0, st, dU, dw, Clo, Chi, beta);
  return check_launch("wino_weights_t");
}

}  // namespace wfae
