// wino.hip — Winograd transforms for the 4x4 stride-2 convolutions.
//
// A 4x4 stride-2 pad-1 convolution is the sum of four 2x2 stride-1 "valid" convolutions, one per input
// parity phase:   lo[oy][ox] = sum_{p,q} sum_{a,b} w[2a+p][2b+q] * P_pq[oy+a][ox+b],
// P_pq[i][j] = hi[2i-1+p][2j-1+q]  (zero outside the image), i in [0,Hlo], j in [0,Wlo].
// Each phase goes through a Winograd algorithm F(MxM, 2x2) on N x N input tiles (N = M + 1): N*N GEMMs
// M_xi = U_xi (Clo x 4Chi) * V_xi (4Chi x T) over T = NB*Hlo*Wlo/M^2 tiles instead of one implicit GEMM with
// K = 16 Chi.
//   variant 0: F(2x2,2x2), N = 3:  9 multiplies per 2x2 outputs  -> 18 FLOP per (lo ch, hi ch, lo pixel) vs 32
//   variant 1: F(4x4,2x2), N = 5: 25 multiplies per 4x4 outputs  -> 12.5 FLOP
//   V = B^T d B   (d = N x N tile of a phase; the four phases' tiles are one 2N x 2N hi patch at stride 2M)
//   U = G g G^T   (g = the phase's 2x2 filter)
//   Y = A^T M A
// F(4,2) uses the interpolation points {0, 1, -1, 2, -2} (Toom-Cook; matrices derived and checked in
// tests/test_host_logic_cpu.py): in fp32 its error is that of the direct form (1.7e-6 vs 1.3e-6 max-rel
// over K = 2048 products), unlike the {0, +-1, 2, inf} set (4e-6).
// The transposed convolution / data gradient is the exact adjoint  hi = In^T( U^T * Out^T(lo) ), the
// weight gradient  dU_xi = Out^T(dlo)_xi * In(hi)_xi^T,  dw = G^T dU G.
// Layouts: V[xi][c][t], c = (p*2+q)*Chi + h;  M[xi][l][t];  U[xi][l][c];  t = (n*Hlo/M + ty)*Wlo/M + tx.
// All kernels are memory-bound streaming kernels: consecutive threads own consecutive tiles tx.
#include "common.h"

using namespace wfae;

namespace {

struct W22 {
  static constexpr int N = 3, M = 2;
  static constexpr float BT[3][3] = {{1, -1, 0}, {0, 1, 0}, {0, -1, 1}};
  static constexpr float G[3][2] = {{1, 0}, {1, 1}, {0, 1}};
  static constexpr float AT[2][3] = {{1, 1, 0}, {0, 1, 1}};
};
struct W42 {
  static constexpr int N = 5, M = 4;
  static constexpr float BT[5][5] = {{4, 0, -5, 0, 1}, {0, 4, 4, -1, -1}, {0, -4, 4, 1, -1}, {0, -2, -1, 2, 1}, {0, 2, -1, -2, 1}};
  static constexpr float G[5][2] = {{0.25f, 0.f}, {1.f / 6, 1.f / 6}, {1.f / 6, -1.f / 6}, {1.f / 24, 1.f / 12}, {1.f / 24, -1.f / 12}};
  static constexpr float AT[4][5] = {{1, 1, 1, 1, 1}, {0, 1, -1, 2, -2}, {0, 1, 1, 4, 4}, {0, 1, -1, 8, -8}};
};

// out[a][b] = sum_{i,j} L[a][i] * in[i][j] * L[b][j]   (L is RA x CA, compile-time, zeros skipped)
template <int RA, int CA, typename Lm>
__device__ __forceinline__ void sandwich(const float (&in)[CA][CA], float (&out)[RA][RA], Lm L) {
  float tmp[RA][CA];
#pragma unroll
  for (int a = 0; a < RA; ++a)
#pragma unroll
    for (int j = 0; j < CA; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < CA; ++i)
        if (L(a, i) != 0.f) s = fmaf(L(a, i), in[i][j], s);
      tmp[a][j] = s;
    }
#pragma unroll
  for (int a = 0; a < RA; ++a)
#pragma unroll
    for (int b = 0; b < RA; ++b) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < CA; ++j)
        if (L(b, j) != 0.f) s = fmaf(L(b, j), tmp[a][j], s);
      out[a][b] = s;
    }
}

// operand store of the transforms that feed GEMMs: fp32, or (SPLIT) the three bf16 planes h + m + l == v of
// splitgemm.hip, `plane` elements apart
template <bool SPLIT>
__device__ __forceinline__ void put_operand(void* __restrict__ base, long idx, long plane, float v) {
  if constexpr (!SPLIT) {
    static_cast<float*>(base)[idx] = v;
  } else {
    unsigned short h, m, l;
    split3(v, h, m, l);
    unsigned short* __restrict__ o = static_cast<unsigned short*>(base) + idx;
    o[0] = h;
    if (plane) {   // plane == 0: the h plane alone (bf16-rounded operands, 'medium' precision)
      o[plane] = m;
      o[2 * plane] = l;
    }
  }
}

// The NX = N * N operand values of one tile and phase -> base[idx + xi * xi_stride (+ plane * pl)], xi < NX.  Consecutive lanes
// own consecutive tiles, so per (xi, plane) a wave writes 64 consecutive bf16 = 128 bytes — as 64 two-byte stores (300 store
// instructions per tile and thread for F(4x4,2x2): the transforms ran at 4.0 - 4.9 TB/s against 6.1 for the BatchNorm passes).
// PACKED form (a full wave on one operand row segment that starts on a 16-byte boundary): one plane at a time goes through a
// wave-private LDS image tb[xi][64] (ds_write_b16, lane-linear) and leaves as 16-byte pieces — piece j = (xi = j >> 3, eight
// tiles 8 (j & 7) ..) is read by lane j (a linear, conflict-free ds_read_b128) and stored whole: 8 rows x 128 bytes per store
// instruction, 4 (2) store instructions per plane instead of 25 (9).
template <bool SPLIT, int NX>
__device__ __forceinline__ void put_operands(void* __restrict__ base, long idx, long xi_stride, long plane, const float (&v)[NX],
                                             unsigned short* __restrict__ tb) {
  if constexpr (!SPLIT) {
#pragma unroll
    for (int xi = 0; xi < NX; ++xi) static_cast<float*>(base)[idx + xi * xi_stride] = v[xi];
  } else {
    const int lane = threadIdx.x & 63;
    const long idx0 = __shfl(idx, 0, 64);
    const bool full = __builtin_amdgcn_ballot_w64(idx == idx0 + lane) == ~0ull && (idx0 & 7) == 0 && (xi_stride & 7) == 0 &&
                      (plane & 7) == 0;
    if (!full) {
#pragma unroll
      for (int xi = 0; xi < NX; ++xi) put_operand<true>(base, idx + xi * xi_stride, plane, v[xi]);
      return;
    }
    unsigned short* __restrict__ o = static_cast<unsigned short*>(base) + idx0;
    const int nplanes = plane ? 3 : 1;
    for (int pl = 0; pl < nplanes; ++pl) {
#pragma unroll
      for (int xi = 0; xi < NX; ++xi) {
        unsigned short h, m, l;
        split3(v[xi], h, m, l);
        tb[xi * 64 + lane] = pl == 0 ? h : (pl == 1 ? m : l);
      }
#pragma unroll
      for (int r = 0; r < (NX * 8 + 63) / 64; ++r) {
        const int j = r * 64 + lane;
        if (j < NX * 8) {
          const wfae_vu4 piece = *reinterpret_cast<const wfae_vu4*>(tb + j * 8);
          *reinterpret_cast<wfae_vu4*>(o + pl * plane + (long)(j >> 3) * xi_stride + (j & 7) * 8) = piece;
        }
      }
    }
  }
}

// thread -> (channel c, image n, tile tl) of the transform kernels without a block reduction.  Two launch shapes
// (pick_grid): large images — one block per (<= 256 tiles, channel, image), the blocks in flight work on neighbouring
// channels of one image (contiguous input); small images (< 256 tiles) — consecutive threads own consecutive t = (image,
// tile) of one channel, so that a wave still writes 64 consecutive elements of an operand row: a 24 x 24 image has 36 tiles,
// 72 bytes of a row, which left 220 of 256 lanes idle and every cache line of the operand shared by two blocks
// (wino_in on 1024 ch @ 48 x 48, B = 32: 0.51 -> 0.25 ms).  false: nothing to do for this thread.
__device__ __forceinline__ bool tile_of_thread(int C, int Timg, long T, int& c, int& n, int& tl) {
  if (gridDim.y * gridDim.z > 1) {
    tl = blockIdx.x * blockDim.x + threadIdx.x;
    c = blockIdx.y;
    n = blockIdx.z;
    return tl < Timg;
  }
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  c = (int)(idx / T);
  if (c >= C) return false;
  const int tt = (int)(idx - (long)c * T);
  n = tt / Timg;
  tl = tt - n * Timg;
  return true;
}

// hi (N,Chi,2Hlo,2Wlo) -> V[N*N][4Chi][T]
template <typename WV, bool SPLIT = false, typename ET = float>
__global__ __launch_bounds__(256) void wino_in_kernel(const ET* __restrict__ hi, void* __restrict__ V, int Chi,
                                                      int Hlo, int Wlo, long T, int planes = 3) {
  constexpr int N = WV::N, M = WV::M, PSZ = 2 * N;
  const int TW = Wlo / M, TH = Hlo / M, Timg = TW * TH;
  int h, n, tl;
  if (!tile_of_thread(Chi, Timg, T, h, n, tl)) return;
  const int ty = tl / TW, tx = tl - ty * TW;
  const int H = 2 * Hlo, W = 2 * Wlo;
  const ET* __restrict__ src = hi + (long)(n * Chi + h) * H * W;
  const long t = (long)n * Timg + tl;
  const long xi_stride = 4L * Chi * T;
  const int Y0 = 2 * M * ty - 1, X0 = 2 * M * tx - 1;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    // the two phases (p, 0) and (p, 1) share their rows: load N rows of the 2N-wide patch once
    float rowbuf[N][PSZ];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int Y = Y0 + 2 * i + p;
      const bool rok = Y >= 0 && Y < H;
      const ET* __restrict__ row = src + (long)(rok ? Y : 0) * W;
      // the patch row is hi[X0 .. X0 + 2M + 1] with X0 = 2M tx - 1: one halo element, 2M elements that start on a
      // 16-byte boundary (W % 4 == 0, the tiles cover the row exactly), one halo element -> 2 + M/2 loads instead of
      // 2M + 2 (the kernel was bound by load instructions: lanes are 8 M bytes apart, every dword load of a wave
      // touches 16 cache lines)
      const float v0 = ld1(row + (X0 >= 0 ? X0 : 0));
      rowbuf[i][0] = (rok && X0 >= 0) ? v0 : 0.f;
#pragma unroll
      for (int c4 = 0; c4 < 2 * M; c4 += 4) {
        const float4 v = ld4(row + X0 + 1 + c4);
        rowbuf[i][1 + c4] = rok ? v.x : 0.f;
        rowbuf[i][2 + c4] = rok ? v.y : 0.f;
        rowbuf[i][3 + c4] = rok ? v.z : 0.f;
        rowbuf[i][4 + c4] = rok ? v.w : 0.f;
      }
      const int XL = X0 + PSZ - 1;
      const float vl = ld1(row + (XL < W ? XL : 0));
      rowbuf[i][PSZ - 1] = (rok && XL < W) ? vl : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float d[N][N], v[N][N];
#pragma unroll
      for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) d[i][j] = rowbuf[i][2 * j + q];
      sandwich<N, N>(d, v, [](int a, int i) { return WV::BT[a][i]; });
      const long dst = ((long)(p * 2 + q) * Chi + h) * T + t;
      // (the packed 16-byte stores of put_operands were tried here too, round 4: 124 instead of 72 registers, 4 instead of 7 waves
      // per SIMD, and no faster — 0.95 / 0.47 / 1.77 ms against 0.88 / 0.75 / 1.67 of the two-byte stores and 0.72 / 0.36 /
      // 1.30 ms for the fp32 operand form: this kernel is bound by its patch reads, whose lanes sit 32 bytes apart)
#pragma unroll
      for (int u = 0; u < N; ++u)
#pragma unroll
        for (int w = 0; w < N; ++w) put_operand<SPLIT>(V, dst + (long)(u * N + w) * xi_stride, planes == 3 ? (long)(N * N) * xi_stride : 0, v[u][w]);
    }
  }
}

// adjoint of wino_in: dV[N*N][4Chi][T] -> hi (N,Chi,2Hlo,2Wlo).  Every thread produces one 2M x 2M block of hi:
// the inverse transform E = B dV B^T of its own tile gives the interior, the rows/columns shared with the
// neighbouring tiles (tile patches are 2N = 2M+2 wide at stride 2M) are gathered from them — no atomics.
// hi row 2M*ty + r belongs to phase p = (r & 1) ^ 1 and phase-row i = (r + 1 - p) / 2 of tile ty; row r = 0 also is
// phase-row M of tile ty-1, row r = 2M-1 also phase-row 0 of tile ty+1 (same for columns).
// STATS: the block also leaves the sum / sum of squares of the values it writes (one channel of one image, <= 256 blocks
// of 2M x 2M) in part[(split * Chi + h) * 2 + {0,1}], split = n * gridDim.x + blockIdx.x — for the BatchNorm that follows
// a ConvTranspose2d (DecBlock.up)
template <typename WV, bool STATS = false, typename ET = float>
__global__ __launch_bounds__(256) void wino_in_t_kernel(const float* __restrict__ dV, ET* __restrict__ hi, int Chi,
                                                        int Hlo, int Wlo, long T, double* __restrict__ part = nullptr) {
  constexpr int N = WV::N, M = WV::M, BS = 2 * M;
  const int TW = Wlo / M, TH = Hlo / M, Timg = TW * TH;
  // STATS: one block per (<= 256 tiles, channel, image) for the block reduction; else the flat (channel, tile) index
  int h, n, tl0;
  bool active;
  if constexpr (STATS) {
    tl0 = blockIdx.x * blockDim.x + threadIdx.x;
    h = blockIdx.y;
    n = blockIdx.z;
    active = tl0 < Timg;
  } else {
    active = tile_of_thread(Chi, Timg, T, h, n, tl0);
  }
  if (!STATS && !active) return;
  const int tl = active ? tl0 : 0;   // STATS: idle lanes of the last block recompute tile 0 (stores and sums masked) and
                                     // take part in the block reduction
  const int ty = tl / TW, tx = tl - ty * TW;
  const int W = 2 * Wlo;
  const long xi_stride = 4L * Chi * T;
  float o[BS][BS];
#pragma unroll
  for (int r = 0; r < BS; ++r)
#pragma unroll
    for (int c = 0; c < BS; ++c) o[r][c] = 0.f;
  auto Bm = [](int i, int u) { return WV::BT[u][i]; };   // B = (B^T)^T
  // own tile: full inverse transform per phase
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float* __restrict__ base = dV + ((long)(p * 2 + q) * Chi + h) * T + (long)n * Timg + tl;
      float dv[N][N], e[N][N];
#pragma unroll
      for (int u = 0; u < N; ++u)
#pragma unroll
        for (int v = 0; v < N; ++v) dv[u][v] = base[(long)(u * N + v) * xi_stride];
      sandwich<N, N>(dv, e, Bm);
      // phase-row i lands on block row r = 2i - 1 + p (0 <= r < 2M), same for columns
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int r = 2 * i - 1 + p;
        if (r < 0 || r >= BS) continue;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const int c = 2 * j - 1 + q;
          if (c < 0 || c >= BS) continue;
          o[r][c] += e[i][j];
        }
      }
    }
  // neighbours.  A tile above contributes its phase-row M (row phase p = 1) to block row 0, a tile below its
  // phase-row 0 (p = 0) to block row 2M-1; the same for columns with q.
  auto base_of = [&](int p, int q, int tyy, int txx) -> const float* {
    return dV + ((long)(p * 2 + q) * Chi + h) * T + (long)n * Timg + (long)tyy * TW + txx;
  };
  // vertical neighbours: one block row, all columns
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int tyy = s == 0 ? ty - 1 : ty + 1;
    if (tyy < 0 || tyy >= TH) continue;
    const int p = s == 0 ? 1 : 0, i = s == 0 ? M : 0, r = s == 0 ? 0 : BS - 1;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float* __restrict__ base = base_of(p, q, tyy, tx);
      float tv[N];
#pragma unroll
      for (int v = 0; v < N; ++v) {
        float a = 0.f;
#pragma unroll
        for (int u = 0; u < N; ++u)
          if (Bm(i, u) != 0.f) a = fmaf(Bm(i, u), base[(long)(u * N + v) * xi_stride], a);
        tv[v] = a;
      }
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const int c = 2 * j - 1 + q;
        if (c < 0 || c >= BS) continue;
        float a = 0.f;
#pragma unroll
        for (int v = 0; v < N; ++v)
          if (Bm(j, v) != 0.f) a = fmaf(Bm(j, v), tv[v], a);
        o[r][c] += a;
      }
    }
  }
  // horizontal neighbours: one block column, all rows
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int txx = s == 0 ? tx - 1 : tx + 1;
    if (txx < 0 || txx >= TW) continue;
    const int q = s == 0 ? 1 : 0, j = s == 0 ? M : 0, c = s == 0 ? 0 : BS - 1;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const float* __restrict__ base = base_of(p, q, ty, txx);
      float tu[N];
#pragma unroll
      for (int u = 0; u < N; ++u) {
        float a = 0.f;
#pragma unroll
        for (int v = 0; v < N; ++v)
          if (Bm(j, v) != 0.f) a = fmaf(Bm(j, v), base[(long)(u * N + v) * xi_stride], a);
        tu[u] = a;
      }
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int r = 2 * i - 1 + p;
        if (r < 0 || r >= BS) continue;
        float a = 0.f;
#pragma unroll
        for (int u = 0; u < N; ++u)
          if (Bm(i, u) != 0.f) a = fmaf(Bm(i, u), tu[u], a);
        o[r][c] += a;
      }
    }
  }
  // corner neighbours: one element each
#pragma unroll
  for (int sy = 0; sy < 2; ++sy)
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      const int tyy = sy == 0 ? ty - 1 : ty + 1, txx = sx == 0 ? tx - 1 : tx + 1;
      if (tyy < 0 || tyy >= TH || txx < 0 || txx >= TW) continue;
      const int p = sy == 0 ? 1 : 0, i = sy == 0 ? M : 0, r = sy == 0 ? 0 : BS - 1;
      const int q = sx == 0 ? 1 : 0, j = sx == 0 ? M : 0, c = sx == 0 ? 0 : BS - 1;
      const float* __restrict__ base = base_of(p, q, tyy, txx);
      float a = 0.f;
#pragma unroll
      for (int u = 0; u < N; ++u) {
        if (Bm(i, u) == 0.f) continue;
#pragma unroll
        for (int v = 0; v < N; ++v)
          if (Bm(j, v) != 0.f) a = fmaf(Bm(i, u) * Bm(j, v), base[(long)(u * N + v) * xi_stride], a);
      }
      o[r][c] += a;
    }
  ET* __restrict__ out = hi + ((long)(n * Chi + h) * (2 * Hlo) + BS * ty) * W + BS * tx;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int r = 0; r < BS; ++r)
#pragma unroll
    for (int c4 = 0; c4 < BS; c4 += 4) {
      if constexpr (STATS && sizeof(ET) == 2) {   // the sums are those of the rounded values the BatchNorm will read
#pragma unroll
        for (int q = 0; q < 4; ++q) o[r][c4 + q] = rounded_as(out, o[r][c4 + q]);
      }
      if (active) st4(out + (long)r * W + c4, make_float4(o[r][c4], o[r][c4 + 1], o[r][c4 + 2], o[r][c4 + 3]));
      if (STATS && active) {
        s1 += (double)((o[r][c4] + o[r][c4 + 1]) + (o[r][c4 + 2] + o[r][c4 + 3]));
        s2 += (double)(fmaf(o[r][c4], o[r][c4], o[r][c4 + 1] * o[r][c4 + 1]) + fmaf(o[r][c4 + 2], o[r][c4 + 2], o[r][c4 + 3] * o[r][c4 + 3]));
      }
    }
  if constexpr (STATS) {
    __shared__ double sm[16];
    const double r1 = block_sum(s1, sm);
    const double r2 = block_sum(s2, sm);
    if (threadIdx.x == 0) {
      const long split = (long)n * gridDim.x + blockIdx.x;
      part[(split * Chi + h) * 2 + 0] = r1;
      part[(split * Chi + h) * 2 + 1] = r2;
    }
  }
}

// M[N*N][Clo][T] -> lo (N,Clo,Hlo,Wlo):  Y = A^T M A
// STATS: the block also leaves the sum / sum of squares of the values it writes (one channel, <= 256 tiles) in
// part[(split * Clo + l) * 2 + {0,1}], split = n * gridDim.x + blockIdx.x, for the BatchNorm that follows the convolution
template <typename WV, bool STATS = false, typename ET = float>
__global__ __launch_bounds__(256) void wino_out_kernel(const float* __restrict__ Mx, ET* __restrict__ lo, int Clo,
                                                       int Hlo, int Wlo, long T, double* __restrict__ part = nullptr) {
  constexpr int N = WV::N, M = WV::M;
  const int TW = Wlo / M, TH = Hlo / M, Timg = TW * TH;
  int l = blockIdx.y, n = blockIdx.z, tl = blockIdx.x * blockDim.x + threadIdx.x;
  if constexpr (STATS) {
    __shared__ double sm[16];
    double s1 = 0.0, s2 = 0.0;
    if (tl < Timg) {
      const int ty = tl / TW, tx = tl - ty * TW;
      const long xi_stride = (long)Clo * T;
      const float* __restrict__ src = Mx + (long)l * T + (long)n * Timg + tl;
      float m[N][N], y[M][M];
#pragma unroll
      for (int u = 0; u < N; ++u)
#pragma unroll
        for (int v = 0; v < N; ++v) m[u][v] = src[(long)(u * N + v) * xi_stride];
      sandwich<M, N>(m, y, [](int a, int i) { return WV::AT[a][i]; });
      ET* __restrict__ dst = lo + ((long)(n * Clo + l) * Hlo + M * ty) * Wlo + M * tx;
#pragma unroll
      for (int a = 0; a < M; ++a) {
        if constexpr (sizeof(ET) == 2) {   // the sums are those of the rounded values the BatchNorm will read
#pragma unroll
          for (int q = 0; q < M; ++q) y[a][q] = rounded_as(dst, y[a][q]);
        }
        if constexpr (M == 2) {
          st2(dst + (long)a * Wlo, make_float2(y[a][0], y[a][1]));
          s1 += (double)(y[a][0] + y[a][1]);
          s2 += (double)fmaf(y[a][0], y[a][0], y[a][1] * y[a][1]);
        } else {
          st4(dst + (long)a * Wlo, make_float4(y[a][0], y[a][1], y[a][2], y[a][3]));
          s1 += (double)((y[a][0] + y[a][1]) + (y[a][2] + y[a][3]));
          s2 += (double)(fmaf(y[a][0], y[a][0], y[a][1] * y[a][1]) + fmaf(y[a][2], y[a][2], y[a][3] * y[a][3]));
        }
      }
    }
    const double r1 = block_sum(s1, sm);
    const double r2 = block_sum(s2, sm);
    if (threadIdx.x == 0) {
      const long split = (long)n * gridDim.x + blockIdx.x;
      part[(split * Clo + l) * 2 + 0] = r1;
      part[(split * Clo + l) * 2 + 1] = r2;
    }
    return;
  }
  if (!tile_of_thread(Clo, Timg, T, l, n, tl)) return;
  const int ty = tl / TW, tx = tl - ty * TW;
  const long xi_stride = (long)Clo * T;
  const float* __restrict__ src = Mx + (long)l * T + (long)n * Timg + tl;
  float m[N][N], y[M][M];
#pragma unroll
  for (int u = 0; u < N; ++u)
#pragma unroll
    for (int v = 0; v < N; ++v) m[u][v] = src[(long)(u * N + v) * xi_stride];
  sandwich<M, N>(m, y, [](int a, int i) { return WV::AT[a][i]; });
  ET* __restrict__ dst = lo + ((long)(n * Clo + l) * Hlo + M * ty) * Wlo + M * tx;
#pragma unroll
  for (int a = 0; a < M; ++a) {
    if constexpr (M == 2) st2(dst + (long)a * Wlo, make_float2(y[a][0], y[a][1]));
    else st4(dst + (long)a * Wlo, make_float4(y[a][0], y[a][1], y[a][2], y[a][3]));
  }
}

// adjoint of wino_out: lo (N,Clo,Hlo,Wlo) -> Mt[N*N][Clo][T]:  Mt = A Y A^T
template <typename WV, bool SPLIT = false, typename ET = float>
__global__ __launch_bounds__(256) void wino_out_t_kernel(const ET* __restrict__ lo, void* __restrict__ Mt, int Clo,
                                                         int Hlo, int Wlo, long T, int planes = 3) {
  constexpr int N = WV::N, M = WV::M;
  __shared__ __attribute__((aligned(16))) unsigned short tbuf[SPLIT ? 4 * N * N * 64 : 8];   // put_operands: per-wave image
  unsigned short* const tb = tbuf + (SPLIT ? (threadIdx.x >> 6) * (N * N * 64) : 0);
  const int TW = Wlo / M, TH = Hlo / M, Timg = TW * TH;
  int l, n, tl;
  if (!tile_of_thread(Clo, Timg, T, l, n, tl)) return;
  const int ty = tl / TW, tx = tl - ty * TW;
  const ET* __restrict__ src = lo + ((long)(n * Clo + l) * Hlo + M * ty) * Wlo + M * tx;
  float y[M][M], m[N][N];
#pragma unroll
  for (int a = 0; a < M; ++a) {
    if constexpr (M == 2) {
      const float2 v = ld2(src + (long)a * Wlo);
      y[a][0] = v.x; y[a][1] = v.y;
    } else {
      const float4 v = ld4(src + (long)a * Wlo);
      y[a][0] = v.x; y[a][1] = v.y; y[a][2] = v.z; y[a][3] = v.w;
    }
  }
  sandwich<N, M>(y, m, [](int u, int a) { return WV::AT[a][u]; });   // A = (A^T)^T
  const long xi_stride = (long)Clo * T;
  const long dst = (long)l * T + (long)n * Timg + tl;
  float mf[N * N];
#pragma unroll
  for (int u = 0; u < N; ++u)
#pragma unroll
    for (int v = 0; v < N; ++v) mf[u * N + v] = m[u][v];
  put_operands<SPLIT, N * N>(Mt, dst, xi_stride, planes == 3 ? (long)(N * N) * xi_stride : 0, mf, tb);
}

// w (Clo,Chi,4,4) -> U[N*N][Clo][4Chi]:  U = G g G^T per phase, g[a][b] = w[2a+p][2b+q]
// SPLIT: bf16 planes; TRANSPOSED: U^T[N*N][4Chi][Clo] (the A operand of the up GEMM with K = Clo contiguous), consecutive
// threads own consecutive lo channels
template <typename WV, bool SPLIT = false, bool TRANSPOSED = false>
__global__ __launch_bounds__(256) void wino_weights_kernel(const float* __restrict__ w, void* __restrict__ U, int Clo,
                                                           int Chi, int planes = 3) {
  constexpr int N = WV::N;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i0 >= (long)Clo * Chi) return;
  const int h = TRANSPOSED ? (int)(i0 / Clo) : (int)(i0 % Chi), l = TRANSPOSED ? (int)(i0 % Clo) : (int)(i0 / Chi);
  const long i = (long)l * Chi + h;
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
      float gg[2][2], u[N][N];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) gg[a][b] = g[(2 * a + p) * 4 + 2 * b + q];
      sandwich<N, 2>(gg, u, [](int a, int i2) { return WV::G[a][i2]; });
      const long c = (long)(p * 2 + q) * Chi + h;
      const long dst = TRANSPOSED ? c * Clo + l : (long)l * K4 + c;
#pragma unroll
      for (int a = 0; a < N; ++a)
#pragma unroll
        for (int b = 0; b < N; ++b) put_operand<SPLIT>(U, dst + (long)(a * N + b) * xi_stride, planes == 3 ? (long)(N * N) * xi_stride : 0, u[a][b]);
    }
}

// dU[N*N][Clo][4Chi] -> dw (Clo,Chi,4,4) (+= when beta):  dg = G^T dU G per phase
template <typename WV>
__global__ __launch_bounds__(256) void wino_weights_t_kernel(const float* __restrict__ dU, float* __restrict__ dw,
                                                             int Clo, int Chi, int beta) {
  constexpr int N = WV::N;
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
      float m[N][N], dg[2][2];
#pragma unroll
      for (int a = 0; a < N; ++a)
#pragma unroll
        for (int b = 0; b < N; ++b) m[a][b] = src[(long)(a * N + b) * xi_stride];
      sandwich<2, N>(m, dg, [](int a, int u) { return WV::G[u][a]; });   // G^T
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) g[(2 * a + p) * 4 + 2 * b + q] = dg[a][b];
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

inline dim3 tile_grid(int NB, int C, int Hlo, int Wlo, int M) { return dim3(cdiv((long)(Hlo / M) * (Wlo / M), 256), C, NB); }
// the kernels without a block reduction (tile_of_thread): per (channel, image) blocks for images of >= 256 tiles, else one
// thread per (channel, image, tile) with the tiles of all images fastest
inline dim3 flat_grid(int NB, int C, int Hlo, int Wlo, int M) {
  const long Timg = (long)(Hlo / M) * (Wlo / M);
  if (Timg >= 256 || (long)C * NB == 1) return tile_grid(NB, C, Hlo, Wlo, M);
  return dim3(cdiv((long)C * NB * Timg, 256), 1, 1);
}

}  // namespace

namespace wfae {

#define WFAE_WINO_DISPATCH(KERNEL, GRID, ...)                                                     \
  do {                                                                                            \
    if (variant == 0) hipLaunchKernelGGL((KERNEL<W22>), GRID, dim3(256), 0, st, __VA_ARGS__);     \
    else hipLaunchKernelGGL((KERNEL<W42>), GRID, dim3(256), 0, st, __VA_ARGS__);                  \
  } while (0)

int wino_in(int variant, const float* hi, float* V, int NB, int Chi, int Hlo, int Wlo, hipStream_t st) {
  const int M = variant ? 4 : 2;
  const long T = (long)NB * (Hlo / M) * (Wlo / M);
  WFAE_WINO_DISPATCH(wino_in_kernel, flat_grid(NB, Chi, Hlo, Wlo, M), hi, (void*)V, Chi, Hlo, Wlo, T);
  return check_launch("wino_in");
}
// the *_split forms write the three bf16 planes of splitgemm.hip (plane stride = the operand's element count); the tensor
// they read is fp32 or, in the bf16-storage mode, bf16
template <typename T>
static int wino_in_split_t(int variant, const T* hi, unsigned short* V3, int planes, int NB, int Chi, int Hlo, int Wlo, hipStream_t st) {
  const int M = variant ? 4 : 2;
  const long T_ = (long)NB * (Hlo / M) * (Wlo / M);
  const dim3 grid = flat_grid(NB, Chi, Hlo, Wlo, M);
  if (variant == 0) hipLaunchKernelGGL((wino_in_kernel<W22, true, T>), grid, dim3(256), 0, st, hi, (void*)V3, Chi, Hlo, Wlo, T_, planes);
  else hipLaunchKernelGGL((wino_in_kernel<W42, true, T>), grid, dim3(256), 0, st, hi, (void*)V3, Chi, Hlo, Wlo, T_, planes);
  return check_launch("wino_in_split");
}
int wino_in_split(int variant, const float* hi, unsigned short* V3, int planes, int NB, int Chi, int Hlo, int Wlo, hipStream_t st) {
  return wino_in_split_t(variant, hi, V3, planes, NB, Chi, Hlo, Wlo, st);
}
int wino_in_split(int variant, const unsigned short* hi, unsigned short* V3, int planes, int NB, int Chi, int Hlo, int Wlo, hipStream_t st) {
  return wino_in_split_t(variant, hi, V3, planes, NB, Chi, Hlo, Wlo, st);
}
template <typename T>
static int wino_out_t_split_t(int variant, const T* lo, unsigned short* Mt3, int planes, int NB, int Clo, int Hlo, int Wlo, hipStream_t st) {
  const int M = variant ? 4 : 2;
  const long T_ = (long)NB * (Hlo / M) * (Wlo / M);
  const dim3 grid = flat_grid(NB, Clo, Hlo, Wlo, M);
  if (variant == 0) hipLaunchKernelGGL((wino_out_t_kernel<W22, true, T>), grid, dim3(256), 0, st, lo, (void*)Mt3, Clo, Hlo, Wlo, T_, planes);
  else hipLaunchKernelGGL((wino_out_t_kernel<W42, true, T>), grid, dim3(256), 0, st, lo, (void*)Mt3, Clo, Hlo, Wlo, T_, planes);
  return check_launch("wino_out_t_split");
}
int wino_out_t_split(int variant, const float* lo, unsigned short* Mt3, int planes, int NB, int Clo, int Hlo, int Wlo, hipStream_t st) {
  return wino_out_t_split_t(variant, lo, Mt3, planes, NB, Clo, Hlo, Wlo, st);
}
int wino_out_t_split(int variant, const unsigned short* lo, unsigned short* Mt3, int planes, int NB, int Clo, int Hlo, int Wlo, hipStream_t st) {
  return wino_out_t_split_t(variant, lo, Mt3, planes, NB, Clo, Hlo, Wlo, st);
}
// U3 [3][N*N][Clo][4Chi] and its transpose Ut3 [3][N*N][4Chi][Clo]
int wino_weights_split(int variant, const float* w, unsigned short* U3, unsigned short* Ut3, int planes, int Clo, int Chi,
                       hipStream_t st) {
  const dim3 grid(cdiv((long)Clo * Chi, 256));
  if (variant == 0) {
    hipLaunchKernelGGL((wino_weights_kernel<W22, true, false>), grid, dim3(256), 0, st, w, (void*)U3, Clo, Chi, planes);
    hipLaunchKernelGGL((wino_weights_kernel<W22, true, true>), grid, dim3(256), 0, st, w, (void*)Ut3, Clo, Chi, planes);
  } else {
    hipLaunchKernelGGL((wino_weights_kernel<W42, true, false>), grid, dim3(256), 0, st, w, (void*)U3, Clo, Chi, planes);
    hipLaunchKernelGGL((wino_weights_kernel<W42, true, true>), grid, dim3(256), 0, st, w, (void*)Ut3, Clo, Chi, planes);
  }
  return check_launch("wino_weights_split");
}
template <typename T>
static int wino_in_t_t(int variant, const float* dV, T* hi, int NB, int Chi, int Hlo, int Wlo, double* part, hipStream_t st) {
  const int M = variant ? 4 : 2;
  const long T_ = (long)NB * (Hlo / M) * (Wlo / M);
  if (part) {
    const dim3 grid = tile_grid(NB, Chi, Hlo, Wlo, M);
    if (variant == 0) hipLaunchKernelGGL((wino_in_t_kernel<W22, true, T>), grid, dim3(256), 0, st, dV, hi, Chi, Hlo, Wlo, T_, part);
    else hipLaunchKernelGGL((wino_in_t_kernel<W42, true, T>), grid, dim3(256), 0, st, dV, hi, Chi, Hlo, Wlo, T_, part);
    return check_launch("wino_in_t_stats");
  }
  const dim3 grid = flat_grid(NB, Chi, Hlo, Wlo, M);
  if (variant == 0) hipLaunchKernelGGL((wino_in_t_kernel<W22, false, T>), grid, dim3(256), 0, st, dV, hi, Chi, Hlo, Wlo, T_, (double*)nullptr);
  else hipLaunchKernelGGL((wino_in_t_kernel<W42, false, T>), grid, dim3(256), 0, st, dV, hi, Chi, Hlo, Wlo, T_, (double*)nullptr);
  return check_launch("wino_in_t");
}
int wino_in_t(int variant, const float* dV, float* hi, int NB, int Chi, int Hlo, int Wlo, hipStream_t st) {
  return wino_in_t_t(variant, dV, hi, NB, Chi, Hlo, Wlo, nullptr, st);
}
int wino_in_t(int variant, const float* dV, unsigned short* hi, int NB, int Chi, int Hlo, int Wlo, hipStream_t st) {
  return wino_in_t_t(variant, dV, hi, NB, Chi, Hlo, Wlo, nullptr, st);
}
int wino_in_t_stats(int variant, const float* dV, float* hi, int NB, int Chi, int Hlo, int Wlo, double* part, hipStream_t st) {
  return wino_in_t_t(variant, dV, hi, NB, Chi, Hlo, Wlo, part, st);
}
int wino_in_t_stats(int variant, const float* dV, unsigned short* hi, int NB, int Chi, int Hlo, int Wlo, double* part, hipStream_t st) {
  return wino_in_t_t(variant, dV, hi, NB, Chi, Hlo, Wlo, part, st);
}
template <typename T>
static int wino_out_tt(int variant, const float* Mx, T* lo, int NB, int Clo, int Hlo, int Wlo, double* part, hipStream_t st) {
  const int M = variant ? 4 : 2;
  const long T_ = (long)NB * (Hlo / M) * (Wlo / M);
  if (part) {
    const dim3 grid = tile_grid(NB, Clo, Hlo, Wlo, M);
    if (variant == 0) hipLaunchKernelGGL((wino_out_kernel<W22, true, T>), grid, dim3(256), 0, st, Mx, lo, Clo, Hlo, Wlo, T_, part);
    else hipLaunchKernelGGL((wino_out_kernel<W42, true, T>), grid, dim3(256), 0, st, Mx, lo, Clo, Hlo, Wlo, T_, part);
    return check_launch("wino_out_stats");
  }
  const dim3 grid = flat_grid(NB, Clo, Hlo, Wlo, M);
  if (variant == 0) hipLaunchKernelGGL((wino_out_kernel<W22, false, T>), grid, dim3(256), 0, st, Mx, lo, Clo, Hlo, Wlo, T_, (double*)nullptr);
  else hipLaunchKernelGGL((wino_out_kernel<W42, false, T>), grid, dim3(256), 0, st, Mx, lo, Clo, Hlo, Wlo, T_, (double*)nullptr);
  return check_launch("wino_out");
}
int wino_out(int variant, const float* Mx, float* lo, int NB, int Clo, int Hlo, int Wlo, hipStream_t st) {
  return wino_out_tt(variant, Mx, lo, NB, Clo, Hlo, Wlo, nullptr, st);
}
int wino_out(int variant, const float* Mx, unsigned short* lo, int NB, int Clo, int Hlo, int Wlo, hipStream_t st) {
  return wino_out_tt(variant, Mx, lo, NB, Clo, Hlo, Wlo, nullptr, st);
}
int wino_out_stats(int variant, const float* Mx, float* lo, int NB, int Clo, int Hlo, int Wlo, double* part, hipStream_t st) {
  return wino_out_tt(variant, Mx, lo, NB, Clo, Hlo, Wlo, part, st);
}
int wino_out_stats(int variant, const float* Mx, unsigned short* lo, int NB, int Clo, int Hlo, int Wlo, double* part, hipStream_t st) {
  return wino_out_tt(variant, Mx, lo, NB, Clo, Hlo, Wlo, part, st);
}
int wino_out_stat_splits(int variant, int NB, int Hlo, int Wlo) {
  const int M = variant ? 4 : 2;
  return NB * (int)tile_grid(NB, 1, Hlo, Wlo, M).x;
}
int wino_out_t(int variant, const float* lo, float* Mt, int NB, int Clo, int Hlo, int Wlo, hipStream_t st) {
  const int M = variant ? 4 : 2;
  const long T = (long)NB * (Hlo / M) * (Wlo / M);
  WFAE_WINO_DISPATCH(wino_out_t_kernel, flat_grid(NB, Clo, Hlo, Wlo, M), lo, (void*)Mt, Clo, Hlo, Wlo, T);
  return check_launch("wino_out_t");
}
int wino_weights(int variant, const float* w, float* U, int Clo, int Chi, hipStream_t st) {
  WFAE_WINO_DISPATCH(wino_weights_kernel, dim3(cdiv((long)Clo * Chi, 256)), w, (void*)U, Clo, Chi);
  return check_launch("wino_weights");
}
int wino_weights_t(int variant, const float* dU, float* dw, int Clo, int Chi, int beta, hipStream_t st) {
  WFAE_WINO_DISPATCH(wino_weights_t_kernel, dim3(cdiv((long)Clo * Chi, 256)), dU, dw, Clo, Chi, beta);
  return check_launch("wino_weights_t");
}

}  // namespace wfae
