// c1r.hip — the Bottleneck's 1x1 convolutions (pipeline/models/ae_64x8x8_lin.py:15,19) forward and data gradient on fp32
// tensors at the HBM-bound stages (C <= 256: 128 @384x384 and 256 @192x192 hold two thirds of the family's bytes):
//   Y[img][m][p] = sum_k A[m][k] f(X[img][k][p]) (+ res[img][m][p])        m < M, k < K, p < HW
// "REGISTER-DIRECT": the activation never passes through LDS.  v_mfma_f32_16x16x32_bf16 wants, per lane, eight consecutive
// k of ONE column; NCHW has the pixel contiguous.  A lane therefore loads eight rows k0 + 8 (l >> 4) + i (i < 8) as 16-byte
// pieces of FOUR consecutive pixels 4 (l & 15) .. + 3 and feeds component jj of the eight pieces to the jj-th of four MFMAs:
// the column of MFMA jj handled by lane n is pixel 4 n + jj — a relabelling of columns that the result layout undoes for
// free, because a lane's accumulators [jj = 0..3][q] are four consecutive pixels of row 4 (l >> 4) + q: one 16-byte store.
//   * every global access is a 16-byte piece of a 256-byte row segment (4 rows per wave-instruction), loads and stores
//     alike; no LDS transpose, no LDS epilogue, no block barrier after the prologue: the 8 waves of a block run free of
//     each other, so one wave's split / GELU arithmetic and MFMAs sit under the others' memory waits;
//   * fp32 accuracy on the bf16 matrix pipe: every loaded value is split ONCE (each wave owns all M rows of its 64
//     pixels) into three exact bf16 planes in registers (x = h + m + l, split3 of common.h), six products per fp32
//     product, smallest terms first (splitgemm.hip);
//   * the weights (<= 16 K values) are split by the block itself on its way in and stay in LDS as the three-plane row
//     image of splitgemm.hip (off_row16: conflict-free ds_read_b128 fragments) for the whole launch — blocks are
//     persistent, one per CU, each wave strides over 64-pixel tiles (adjacent waves take adjacent tiles: 2 KiB of every
//     row at a time per block);
//   * narrowing products (K = C, M = C/4: the C -> C/4 forward and the C/4 -> C data gradient... of the other conv): the K
//     loop streams 32-row chunks, one chunk of loads in flight ahead of the multiply, all M rows accumulate at once;
//     widening products (K = C/4, M = C): the split operand of the tile stays in registers and M is walked in passes of
//     16 MG rows with the residual rows of the next pass and the first chunk of the next tile in flight;
//   * optional BatchNorm + GELU prologue f = gelu(x * scale[k] + shift[k]) (bn_act_fwd_kernel's arithmetic, as in
//     gemm.hip's PRO loaders), residual add, and the BatchNorm sums of the result: fp32 sums of four, fp64 across the
//     16 lanes of a DPP row, accumulated in fp64 per wave in LDS over all its tiles and written as ONE partial row per
//     wave (the StatRows format of wfae_conv1x1_fwd_stats, finished by wfae_bn_stats_from_rows).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

using namespace wfae;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct C1RP {
  const float* W;            // A[m][k] = W[m * w_sm + k * w_sk]
  long w_sm, w_sk;
  const float* X;            // [NB][K][HW]
  float* Y;                  // [NB][M][HW]
  const float* res;          // [NB][M][HW] or null
  const float* pro_scale;    // PRO: folded BatchNorm scale / shift of the input channels [K]
  const float* pro_shift;
  double* part0;             // STATS: [gridDim.x * 8][M] sums, one row per wave
  double* part1;             // sums of squares
  int HW;                    // % 64 == 0: a 64-pixel tile never leaves its image
  int tpi;                   // tiles per image
  int ntiles;
  const float* bn_x;         // BNR: the BatchNorm input of the layer in FRONT of this convolution, [NB][M][HW] (the shape of Y)
  const float* bn_tab[4];    // BNR: folded scale, shift, batch mean, invstd of that BatchNorm, [M] each
  const float* bn_gamma;     // BNM 3: that BatchNorm's weight [M], the two sums of its backward reduce ([2 c], [2 c + 1]) and 1 / count
  const float* bn_coef;
  float inv_count;
  int training;
  int m_total;               // rows of Y: nslices * M (M = the 16 MT rows one block owns)
  int nslices;               // M-slices: the blocks b, b + 8, .. of one XCD that share a tile sequence split the rows
};


// row image of splitgemm.hip for 16-row ds_read_b128 fragments (lane = row l & 15, 16-byte chunk l >> 4)
__device__ __forceinline__ unsigned swz16(int r) { return (0x78u >> (((r >> 2) & 3) << 1)) & 3u; }

#define C1R_DPP_F64(v, CTRL)                                                                                     \
  __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true),                       \
                   __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true))
__device__ __forceinline__ double row_sum16(double v) {   // lane 15 of every 16-lane DPP row ends up with the row total
  v += C1R_DPP_F64(v, 0x111);
  v += C1R_DPP_F64(v, 0x112);
  v += C1R_DPP_F64(v, 0x114);
  v += C1R_DPP_F64(v, 0x118);
  return v;
}

// two values -> their three exact bf16 planes, packed (a in the low half)
struct Planes3 { unsigned h, m, l; };
__device__ __forceinline__ Planes3 split_pair(float a, float b) {
  Planes3 r;
  r.h = pack_bf16(a, b);
  const float a1 = a - bf16_lo(r.h), b1 = b - bf16_hi(r.h);
  r.m = pack_bf16(a1, b1);
  r.l = pack_bf16(a1 - bf16_lo(r.m), b1 - bf16_hi(r.m));
  return r;
}

// NBUF: chunk buffers of the narrowing ring; RD: residual-row buffers of the widening ring
// RESN: a streaming (one-pass) kernel that carries a residual
// BNR (B-resident kernels, data gradients): the result is dA, the gradient at the OUTPUT of a BatchNorm + GELU; the reductions of
// that layer's backward pass — sum dU and sum dU xhat per channel, dU = dA gelu'(x a + b), xhat = (x - mu) is: phase 1 of
// wfae_bn_act_bwd, bn_act_bwd_reduce_kernel's arithmetic — are taken here while dA is in registers (x travels through the
// residual ring) and leave as the partial rows of STATS; the separate pass over (dA, x) disappears.
// BNM: 0 none; 1 the BNR form above; 2 the same sums WITHOUT storing dA; 3 the second pass of that layer's backward in the
// epilogue: dx = gamma invstd (dU - sum dU / n - xhat sum dU xhat / n) + skip — bn_act_bwd_dx_kernel's arithmetic — with dA
// recomputed from dT (six MFMAs per product on a kernel that waits for HBM) instead of written by pass 1 and read back by
// pass 2: modes 2 + 3 move 4 C-wide tensors per layer where mode 1 + wfae_bn_act_bwd's second pass move 6.
template <int KCH, int MT, int MG, bool PRO, bool STATS, int RWAVES = 8, int NBUF = 2, int RD = 2, bool RESN = false, int BNM = 0,
          int MINB = (RWAVES == 4 ? 1 : 2)>
__global__ __launch_bounds__(64 * RWAVES, MINB) void c1r_kernel(C1RP p) {
  constexpr int RNT = 64 * RWAVES;
  constexpr int K = 32 * KCH, M = 16 * MT, NPASS = MT / MG;
  constexpr bool BRES = NPASS > 1;                 // the split operand of a tile stays in registers, M in passes
  constexpr bool BNR = BNM == 1 || BNM == 2, DXF = BNM == 3;
  static_assert(MT % MG == 0, "whole passes");
  constexpr int PLANE_B = M * 64;                  // one plane of one 32-deep chunk: M rows x 64 bytes
  constexpr int A_B = KCH * 3 * PLANE_B;
  constexpr int PRO_B = PRO ? 2 * K * 4 : 0;
  static_assert(!BNR || (STATS && !PRO && !RESN && MT > MG), "BNR: a B-resident data-gradient kernel; its sums use the STATS rows");
  static_assert(!DXF || (!STATS && !PRO && !RESN && MT > MG), "BNM 3: a B-resident data-gradient kernel");
  constexpr int ST_B = STATS ? RWAVES * 2 * M * 8 : 0;
  constexpr int BNT_B = BNR ? 4 * M * 4 : DXF ? 7 * M * 4 : 0;
  static_assert(A_B + PRO_B + ST_B + BNT_B <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) unsigned char smem[A_B + PRO_B + ST_B + BNT_B];
  float* const lsc = reinterpret_cast<float*>(smem + A_B);
  float* const lbn = reinterpret_cast<float*>(smem + A_B + PRO_B + ST_B);   // BNR: [4][M]; BNM 3: [7][M]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n16 = lane & 15, kg = lane >> 4;
  double* const lst = reinterpret_cast<double*>(smem + A_B + PRO_B) + (STATS ? wave * 2 * M : 0);
  // M-slices (the C >= 512 products, whose weight image does not fit the LDS as a whole): blocks are dealt round-robin to the
  // XCDs, so blocks b and b + 8 share one; the nslices consecutive blocks-of-an-XCD of a GROUP own the row slices [slice M,
  // (slice + 1) M) of the same tiles — the operand tile one of them pulls from HBM is an L2 hit for the others, and each
  // splits it again for its own rows (free while the split stays under two vector instructions per MFMA).  nslices = 1: a
  // group is a block, group = blockIdx.x.
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int slice = jb % p.nslices, group = (jb / p.nslices) * 8 + xcd;
  const int ngroups = gridDim.x / p.nslices;
  const int m_off = slice * M;

  // ---- the weights: split on the way in, three-plane row image [chunk][plane][m][64 B]
  {
    const bool kfast = p.w_sk == 1;
    for (int idx = t; idx < M * K / 8; idx += RNT) {
      int m, ch;
      if (kfast) { ch = idx % (K / 8); m = idx / (K / 8); } else { m = idx % M; ch = idx / M; }
      const float* src = p.W + (long)(m_off + m) * p.w_sm + (long)(8 * ch) * p.w_sk;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = src[(long)e * p.w_sk];
      const Planes3 q0 = split_pair(v[0], v[1]), q1 = split_pair(v[2], v[3]), q2 = split_pair(v[4], v[5]), q3 = split_pair(v[6], v[7]);
      const u32x4 h = {q0.h, q1.h, q2.h, q3.h}, mm = {q0.m, q1.m, q2.m, q3.m}, l = {q0.l, q1.l, q2.l, q3.l};
      const unsigned off = (unsigned)((ch >> 2) * 3 * PLANE_B + m * 64) + ((((unsigned)ch & 3u) ^ swz16(m)) << 4);
      *reinterpret_cast<u32x4*>(smem + off) = h;
      *reinterpret_cast<u32x4*>(smem + off + PLANE_B) = mm;
      *reinterpret_cast<u32x4*>(smem + off + 2 * PLANE_B) = l;
    }
    if constexpr (PRO) {
      for (int i = t; i < K; i += RNT) {
        lsc[i] = p.pro_scale[i];
        lsc[K + i] = p.pro_shift[i];
      }
    }
    if constexpr (STATS) {
      for (int i = lane; i < 2 * M; i += 64) lst[i] = 0.0;
    }
    if constexpr (BNR || DXF) {
      for (int i = t; i < 4 * M; i += RNT) lbn[i] = p.bn_tab[i / M][m_off + i % M];
    }
    if constexpr (DXF) {   // gamma invstd, sum dU / n, sum dU xhat / n (bn_act_bwd_dx_kernel's gi, k1, k2)
      for (int i = t; i < M; i += RNT) {
        lbn[4 * M + i] = p.bn_gamma[m_off + i] * p.bn_tab[3][m_off + i];
        lbn[5 * M + i] = p.training ? p.bn_coef[2 * (m_off + i) + 0] * p.inv_count : 0.f;
        lbn[6 * M + i] = p.training ? p.bn_coef[2 * (m_off + i) + 1] * p.inv_count : 0.f;
      }
    }
  }
  __syncthreads();

  const long rowB = (long)p.HW * 4;                                          // bytes between two channel rows
  const unsigned lane_in = (unsigned)((4 * n16 + 8 * kg * p.HW) * 4);       // this lane's piece of row 8 kg (+ i rows)
  const unsigned lane_out = (unsigned)((4 * n16 + 4 * kg * p.HW) * 4);      // result rows 4 kg + q
  const unsigned a_rd = (unsigned)(n16 * 64) + (((unsigned)kg ^ swz16(n16)) << 4);
  const int tstride = ngroups * RWAVES;
  int tile = group * RWAVES + wave;

  auto tile_off = [&](int tl, int chans) -> long {   // byte offset of a tile's first pixel in a [NB][chans][HW] tensor
    const int img = tl / p.tpi;
    return ((long)img * chans * p.HW + (long)(tl - img * p.tpi) * 64) * 4;
  };
  auto load_chunk = [&](f32x4 (&r)[8], int tl, int c) {
    const char* xb = reinterpret_cast<const char*>(p.X) + tile_off(tl, K) + (long)(32 * c) * rowB;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = *reinterpret_cast<const f32x4*>(xb + i * rowB + lane_in);
  };
  // prologue + split of JW of the four pixel components (jj0 .. jj0 + JW) of one chunk: b[j][plane] = the eight k of this
  // lane, pixel 4 n16 + jj0 + j
  auto split_part = [&](f32x4 (&r)[8], auto& b, int c, auto jj0c, auto jwc) {
    constexpr int JJ0 = decltype(jj0c)::value, JW = decltype(jwc)::value;
    if constexpr (PRO) {
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(lsc + 32 * c + 8 * kg), s1 = *reinterpret_cast<const f32x4*>(lsc + 32 * c + 8 * kg + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(lsc + K + 32 * c + 8 * kg), h1 = *reinterpret_cast<const f32x4*>(lsc + K + 32 * c + 8 * kg + 4);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float s = i < 4 ? s0[i & 3] : s1[i & 3], h = i < 4 ? h0[i & 3] : h1[i & 3];
#pragma unroll
        for (int j = 0; j < JW; ++j) r[i][JJ0 + j] = gelu_f(fmaf(r[i][JJ0 + j], s, h));
      }
    }
#pragma unroll
    for (int j = 0; j < JW; ++j) {
      const int jj = JJ0 + j;
      const Planes3 q0 = split_pair(r[0][jj], r[1][jj]), q1 = split_pair(r[2][jj], r[3][jj]);
      const Planes3 q2 = split_pair(r[4][jj], r[5][jj]), q3 = split_pair(r[6][jj], r[7][jj]);
      b[j][0] = u32x4{q0.h, q1.h, q2.h, q3.h};
      b[j][1] = u32x4{q0.m, q1.m, q2.m, q3.m};
      b[j][2] = u32x4{q0.l, q1.l, q2.l, q3.l};
    }
  };
  // acc[mt][jj0 + j] += A(rows 16 (mt0 + mt) .., chunk c) x b[j]: six products per fp32 product, smallest terms first
  auto multiply_part = [&](f32x4 (&acc)[MG][4], const auto& b, int c, int a_pass_off, auto jj0c, auto jwc) {
    constexpr int JJ0 = decltype(jj0c)::value, JW = decltype(jwc)::value;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    if constexpr (RWAVES == 4 && MG > 1) {
      // one wave per SIMD (512 registers, nobody else to cover an LDS round trip): every weight fragment of the chunk is read
      // before the first MFMA instead of three at a time with a wait in front of each group of twelve MFMAs
      bf16x8 a[MG][3];
#pragma unroll
      for (int mt = 0; mt < MG; ++mt)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          a[mt][pl] = *reinterpret_cast<const bf16x8*>(smem + a_pass_off + (c * 3 + pl) * PLANE_B + mt * (16 * 64) + a_rd);
      __builtin_amdgcn_sched_barrier(0);   // (without it the scheduler sinks each read back in front of its first use)
#pragma unroll
      for (int mt = 0; mt < MG; ++mt)
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
          for (int j = 0; j < JW; ++j)
            acc[mt][JJ0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][PA[pr]], __builtin_bit_cast(bf16x8, b[j][PB[pr]]), acc[mt][JJ0 + j], 0, 0, 0);
    } else {
#pragma unroll
      for (int mt = 0; mt < MG; ++mt) {
        bf16x8 a[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          a[pl] = *reinterpret_cast<const bf16x8*>(smem + a_pass_off + (c * 3 + pl) * PLANE_B + mt * (16 * 64) + a_rd);
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
          for (int j = 0; j < JW; ++j)
            acc[mt][JJ0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[pr]], __builtin_bit_cast(bf16x8, b[j][PB[pr]]), acc[mt][JJ0 + j], 0, 0, 0);
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I2 = std::integral_constant<int, 2>;
  using I4 = std::integral_constant<int, 4>;
  auto split_chunk = [&](f32x4 (&r)[8], u32x4 (&b)[4][3], int c) { split_part(r, b, c, I0{}, I4{}); };
  auto multiply = [&](f32x4 (&acc)[MG][4], const u32x4 (&b)[4][3], int c, int a_pass_off) { multiply_part(acc, b, c, a_pass_off, I0{}, I4{}); };
  auto load_res = [&](f32x4 (&rv)[MG][4], int tl, int mt0) {
    const char* rb = reinterpret_cast<const char*>((BNR || DXF) ? p.bn_x : p.res) + tile_off(tl, p.m_total) + (long)(m_off + 16 * mt0) * rowB;
#pragma unroll
    for (int mt = 0; mt < MG; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) rv[mt][q] = *reinterpret_cast<const f32x4*>(rb + (16 * mt + q) * rowB + lane_out);
  };
  auto load_skip = [&](f32x4 (&sv)[MG][4], int tl, int mt0) {   // BNM 3: the gradient arriving over the skip connection
    const char* rb = reinterpret_cast<const char*>(p.res) + tile_off(tl, p.m_total) + (long)(m_off + 16 * mt0) * rowB;
#pragma unroll
    for (int mt = 0; mt < MG; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) sv[mt][q] = *reinterpret_cast<const f32x4*>(rb + (16 * mt + q) * rowB + lane_out);
  };
  // lane (n16, kg): acc[mt][jj][q] = Y[row 16 (mt0 + mt) + 4 kg + q][pixel 4 n16 + jj]
  auto store_pass = [&](const f32x4 (&acc)[MG][4], const f32x4 (&rv)[MG][4], const f32x4 (&sv)[MG][4], bool with_res, bool with_skip,
                        int tl, int mt0) {
    char* yb = reinterpret_cast<char*>(p.Y) + tile_off(tl, p.m_total) + (long)(m_off + 16 * mt0) * rowB;
#pragma unroll
    for (int mt = 0; mt < MG; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc[mt][0][q], acc[mt][1][q], acc[mt][2][q], acc[mt][3][q]};
        if (!BNR && !DXF && with_res) v += rv[mt][q];
        if constexpr (DXF) {   // bn_act_bwd_dx_kernel's one(): gi (dU - k1 - xhat k2) + skip
          const int ml = 16 * (mt0 + mt) + 4 * kg + q;
          const float a = lbn[ml], b = lbn[M + ml], mu = lbn[2 * M + ml], is = lbn[3 * M + ml];
          const float gi = lbn[4 * M + ml], k1 = lbn[5 * M + ml], k2 = lbn[6 * M + ml];
          const f32x4 xv = rv[mt][q];
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float du = v[e] * gelu_grad_f(fmaf(xv[e], a, b));
            const float xh = (xv[e] - mu) * is;
            o[e] = gi * (du - k1 - xh * k2) + (with_skip ? sv[mt][q][e] : 0.f);
          }
          *reinterpret_cast<f32x4*>(yb + (16 * mt + q) * rowB + lane_out) = o;
        } else if constexpr (BNM != 2) {
          *reinterpret_cast<f32x4*>(yb + (16 * mt + q) * rowB + lane_out) = v;
        }
        if constexpr (BNR) {   // bn_act_bwd_reduce_kernel's quad(): dU = dA gelu'(x a + b), sums of dU and dU xhat
          const int ml = 16 * (mt0 + mt) + 4 * kg + q;
          const float a = lbn[ml], b = lbn[M + ml], mu = lbn[2 * M + ml], is = lbn[3 * M + ml];
          const f32x4 xv = rv[mt][q];
          const float d0 = v.x * gelu_grad_f(fmaf(xv.x, a, b)), d1 = v.y * gelu_grad_f(fmaf(xv.y, a, b));
          const float d2 = v.z * gelu_grad_f(fmaf(xv.z, a, b)), d3 = v.w * gelu_grad_f(fmaf(xv.w, a, b));
          const float h0 = (xv.x - mu) * is, h1 = (xv.y - mu) * is, h2 = (xv.z - mu) * is, h3 = (xv.w - mu) * is;
          const double e1 = row_sum16((double)((d0 + d1) + (d2 + d3)));
          const double e2 = row_sum16((double)(fmaf(d0, h0, d1 * h1) + fmaf(d2, h2, d3 * h3)));
          if (n16 == 15) {
            lst[ml] += e1;
            lst[M + ml] += e2;
          }
        } else if constexpr (STATS) {   // fp32 sums of four enter the fp64 reduction, as in chan_reduce_kernel
          const float s1 = (v.x + v.y) + (v.z + v.w);
          const float s2 = fmaf(v.x, v.x, v.y * v.y) + fmaf(v.z, v.z, v.w * v.w);
          const double d1 = row_sum16((double)s1), d2 = row_sum16((double)s2);
          if (n16 == 15) {
            const int m = 16 * (mt0 + mt) + 4 * kg + q;
            lst[m] += d1;
            lst[M + m] += d2;
          }
        }
      }
  };

  const bool with_res = BNR || DXF || p.res != nullptr;
  const bool with_skip = DXF && p.res != nullptr;
  if (tile < p.ntiles) {
    if constexpr (!BRES) {
      // ---- narrowing: stream the K chunks through a ring of NBUF register buffers, NBUF - 1 chunks of loads in flight ahead of
      // the multiply (ACROSS tiles: the ring positions repeat from tile to tile because NBUF divides KCH); all M rows at once.
      // A lone chunk in flight per wave left the loads of a CU in flight only part of the time (a wave issues its next chunk
      // when it has finished multiplying the previous one): 4.6 - 4.8 TB/s; the same load shape kept in flight without a gap
      // reads 6.0 - 6.5 TB/s (tools/probe/hbm_streams.hip).
      static_assert(KCH % NBUF == 0 && NBUF >= 2, "ring positions must repeat per tile");
      constexpr int D = NBUF - 1;
      f32x4 ring[NBUF][8];
#pragma unroll
      for (int c = 0; c < D; ++c) load_chunk(ring[c], tile, c);
      while (tile < p.ntiles) {
        const int nxt_tile = tile + tstride < p.ntiles ? tile + tstride : tile;   // past the end: a harmless re-read
        f32x4 acc[MG][4];
        f32x4 rv[MG][4];   // residual rows (RESN kernels: the sliced widening products), in flight under the last chunk
#pragma unroll
        for (int mt = 0; mt < MG; ++mt)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) acc[mt][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
          if (c + D < KCH) load_chunk(ring[(c + D) % NBUF], tile, c + D);
          else load_chunk(ring[(c + D) % NBUF], nxt_tile, c + D - KCH);
          if constexpr (RESN) {
            if (c == KCH - 1 && with_res) load_res(rv, tile, 0);
          }
          // fences: without them the scheduler hoists the loads of every chunk of the unrolled loop to the top (spills)
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (MG >= 4) {   // 64 accumulator registers: the chunk in two halves of two pixel components (24 fragment registers)
            u32x4 b2[2][3];
            split_part(ring[c % NBUF], b2, c, I0{}, I2{});
            multiply_part(acc, b2, c, 0, I0{}, I2{});
            __builtin_amdgcn_sched_barrier(0);
            split_part(ring[c % NBUF], b2, c, I2{}, I2{});
            multiply_part(acc, b2, c, 0, I2{}, I2{});
          } else {
            u32x4 b[4][3];
            split_chunk(ring[c % NBUF], b, c);
            multiply(acc, b, c, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        store_pass(acc, rv, rv, RESN && with_res, false, tile, 0);
        tile += tstride;
      }
    } else {
      // ---- widening: the tile's split operand stays in registers, M in passes of 16 MG rows; the residual rows travel
      // RD - 1 passes ahead through a ring of RD register buffers, the next tile's operand under the last passes
      static_assert(NPASS % RD == 0 && RD >= 2 && KCH <= RD, "ring positions must repeat per tile");
      f32x4 raw[KCH][8];
#pragma unroll
      for (int c = 0; c < KCH; ++c) load_chunk(raw[c], tile, c);
      while (tile < p.ntiles) {
        const int nxt_tile = tile + tstride < p.ntiles ? tile + tstride : tile;
        f32x4 rr[RD][MG][4];
        f32x4 rs[DXF ? RD : 1][MG][4];   // BNM 3: the skip gradient's rows travel beside x's
        if (with_res) {
#pragma unroll
          for (int u = 0; u < RD - 1; ++u) {
            load_res(rr[u], tile, u * MG);
            if constexpr (DXF) {
              if (with_skip) load_skip(rs[u], tile, u * MG);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        u32x4 b[KCH][4][3];
#pragma unroll
        for (int c = 0; c < KCH; ++c) split_chunk(raw[c], b[c], c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int pass0 = 0; pass0 < NPASS; pass0 += RD) {
#pragma unroll
          for (int u = 0; u < RD; ++u) {
            const int pass = pass0 + u, mt0 = pass * MG;
            if (with_res && pass + RD - 1 < NPASS) {
              load_res(rr[(u + RD - 1) % RD], tile, (pass + RD - 1) * MG);
              if constexpr (DXF) {
                if (with_skip) load_skip(rs[(u + RD - 1) % RD], tile, (pass + RD - 1) * MG);
              }
            }
            // the last KCH passes of the tile: one chunk each of the wave's next tile (its first passes do not wait)
            if (u >= RD - KCH && pass0 == NPASS - RD) load_chunk(raw[u >= RD - KCH ? u - (RD - KCH) : 0], nxt_tile, u - (RD - KCH));
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[MG][4];
#pragma unroll
            for (int mt = 0; mt < MG; ++mt)
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) acc[mt][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < KCH; ++c) multiply(acc, b[c], c, mt0 * (16 * 64));
            store_pass(acc, rr[u], rs[DXF ? u : 0], with_res, with_skip, tile, mt0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        tile += tstride;
      }
    }
  }
  if constexpr (STATS) {   // one partial row per wave (zeros from a wave that had no tile)
    const long row = (long)group * RWAVES + wave;   // the slices of a group fill disjoint columns of the group's rows
    for (int i = lane; i < M; i += 64) {
      p.part0[row * p.m_total + m_off + i] = lst[i];
      p.part1[row * p.m_total + m_off + i] = lst[M + i];
    }
  }
}

// served (M, K): the Bottleneck products of the C = 128 and C = 256 stages (one block owns all M rows) and, M-sliced, the
// widening products of the C = 512 / 1024 stages (49 / 98 KiB weight-image slices of 64 rows).  The narrowing products of
// those stages stay on gemm.hip: a (M = 128, K = 512) form with four 32-row slices was built and measured equal or slower
// (0.275 - 0.281 vs 0.267 - 0.290 ms, profiles/r04_kbench_c1r_sliced.txt) — every slice splits the whole operand again and
// at 48 MFMAs per chunk that no longer hides; all of these kernels, old and new, sit at 135 - 165 TF fp32-equivalent, half of
// what six bf16 products per fp32 product allow at the clock the chip holds under MFMA load.
struct ShapeInfo { int sid, nslices, waves; };
inline ShapeInfo shape_of(int M, int K) {
  if (M == 32 && K == 128) return {0, 1, 8};
  if (M == 64 && K == 256) return {1, 1, 8};
  if (M == 128 && K == 32) return {2, 1, 4};
  if (M == 256 && K == 64) return {3, 1, 4};
  if (M == 512 && K == 128) return {4, 8, 4};     // slices of 64 rows
  if (M == 1024 && K == 256) return {5, 16, 4};   // slices of 64 rows
  return {-1, 1, 8};
}
inline int shape_id(int M, int K) { return shape_of(M, K).sid; }

inline int num_cus() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();
  return n;
}

// waves per block, ring depths: measured per shape on one box with the variants interleaved (profiles/r04_kbench_c1r_variants.txt).
// Every shape runs one block per CU (the weight image of the C >= 256 shapes fills the LDS).  Narrowing shapes: 8 waves, one
// chunk in flight ahead of the multiply — a ring of four buffers (three chunks in flight) measured the same (0.314 / 0.308 vs
// 0.311 / 0.310 ms at C = 256, 0.598 vs 0.592 at C = 128): these kernels do not wait for their loads.  Widening, C = 256: 8 waves
// and the residual rows one pass ahead (0.49 ms with residual + sums against 0.50 - 0.52 for 4 waves, 0.53 - 0.57 for deeper
// rings); widening, C = 128 (a 24 KiB weight image): 4 waves of 370 registers with the residual rows seven passes ahead (1.02 ms
// against 1.06 for 8 waves three passes ahead and 1.10 one pass ahead; data gradient 0.574 against 0.632 / 0.656).
// grid: one block per CU, a whole number of (XCD x slice) sets
inline int grid_for(long ntiles, const ShapeInfo& si) {
  if (si.nslices > 1) {
    const int unit = 8 * si.nslices;
    const int g = num_cus() / unit * unit;
    return g > 0 ? g : unit;
  }
  const long g = (ntiles + si.waves - 1) / si.waves;
  return (int)(g < num_cus() ? g : num_cus());
}

template <int KCH, int MT, int MG, int NW, int NBUF, int RD, bool RESN = false>
void launch_shape(const C1RP& p, bool pro, bool stats, int grid, hipStream_t st) {
  const dim3 g((unsigned)grid), b(64 * NW);
  if (pro && stats) hipLaunchKernelGGL((c1r_kernel<KCH, MT, MG, true, true, NW, NBUF, RD, RESN>), g, b, 0, st, p);
  else if (pro) hipLaunchKernelGGL((c1r_kernel<KCH, MT, MG, true, false, NW, NBUF, RD, RESN>), g, b, 0, st, p);
  else if (stats) hipLaunchKernelGGL((c1r_kernel<KCH, MT, MG, false, true, NW, NBUF, RD, RESN>), g, b, 0, st, p);
  else hipLaunchKernelGGL((c1r_kernel<KCH, MT, MG, false, false, NW, NBUF, RD, RESN>), g, b, 0, st, p);
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

namespace {
template <int KCH, int MT, int NW, int RD>
void launch_bnr(const C1RP& p, int grid, hipStream_t st) {
  if (p.Y) hipLaunchKernelGGL((c1r_kernel<KCH, MT, 1, false, true, NW, 2, RD, false, 1, 1>), dim3((unsigned)grid), dim3(64 * NW), 0, st, p);
  else hipLaunchKernelGGL((c1r_kernel<KCH, MT, 1, false, true, NW, 2, RD, false, 2, 1>), dim3((unsigned)grid), dim3(64 * NW), 0, st, p);
}
template <int KCH, int MT, int NW, int RD>
void launch_bndx(const C1RP& p, int grid, hipStream_t st) {
  hipLaunchKernelGGL((c1r_kernel<KCH, MT, 1, false, false, NW, 2, RD, false, 3, 1>), dim3((unsigned)grid), dim3(64 * NW), 0, st, p);
}
constexpr int BNR_WAVES = 8;   // the reduce epilogue is vector-instruction work (GELU', fp64 row sums): two waves per SIMD cover it
}  // namespace

extern "C" {

int wfae_c1r_supported(int M, int K, int HW) {
  return (shape_id(M, K) >= 0 && HW > 0 && HW % 64 == 0 && wfae::split_gemm_enabled()) ? 1 : 0;
}

int wfae_c1r_stat_rows(int M, int K, int NB, int HW) {
  const ShapeInfo si = shape_of(M, K);
  if (si.sid < 0 || HW <= 0 || HW % 64 != 0 || NB <= 0) return 0;
  // (the reduce form of wfae_c1r_bnred runs BNR_WAVES waves per block on the shapes it serves: one capacity for both users)
  return grid_for((long)NB * (HW / 64), si) / si.nslices * ((si.sid == 2 || si.sid == 3) ? BNR_WAVES : si.waves);
}

int wfae_c1r_fwd(const float* w, int64_t w_sm, int64_t w_sk, const float* x, const float* pro_scale, const float* pro_shift,
                 const float* res, float* y, int NB, int K, int M, int HW, double* stat_part, int64_t stat_capacity, int* stat_rows,
                 wfae_stream_t stream) {
  WFAE_REQUIRE(w && x && y, WFAE_ERR_NULL_POINTER, "c1r_fwd: null pointer");
  WFAE_REQUIRE(NB > 0 && K > 0 && M > 0 && HW > 0 && (int64_t)NB * HW < (1ll << 31), WFAE_ERR_BAD_SHAPE, "c1r_fwd: bad shape");
  const ShapeInfo si = shape_of(M, K);
  const int sid = si.sid;
  WFAE_REQUIRE(sid >= 0 && HW % 64 == 0, WFAE_ERR_UNSUPPORTED,
               "c1r_fwd: serves (M, K) = (32, 128), (64, 256), (128, 32), (256, 64), (512, 128), (1024, 256) with "
               "HW %% 64 == 0 (M %d, K %d, HW %d)", M, K, HW);
  WFAE_REQUIRE(wfae::split_gemm_enabled(), WFAE_ERR_UNSUPPORTED, "c1r_fwd: needs fp32 precision with the split GEMMs on");
  WFAE_REQUIRE((w_sm == K && w_sk == 1) || (w_sm == 1 && w_sk == M), WFAE_ERR_BAD_SHAPE,
               "c1r_fwd: the weight is (M, K) row-major (strides K, 1) or its transpose (strides 1, M)");
  WFAE_REQUIRE(al16(x) && al16(y) && (!res || al16(res)), WFAE_ERR_UNSUPPORTED, "c1r_fwd: tensors must be 16-byte aligned");
  WFAE_REQUIRE(!res || M > K, WFAE_ERR_UNSUPPORTED, "c1r_fwd: the residual add belongs to the widening products (M > K)");
  WFAE_REQUIRE((pro_scale != nullptr) == (pro_shift != nullptr) && (stat_part != nullptr) == (stat_rows != nullptr),
               WFAE_ERR_NULL_POINTER, "c1r_fwd: scale / shift and stat_part / stat_rows go together");
  C1RP p = {};
  p.W = w; p.w_sm = w_sm; p.w_sk = w_sk;
  p.X = x; p.Y = y; p.res = res;
  p.pro_scale = pro_scale; p.pro_shift = pro_shift;
  p.HW = HW; p.tpi = HW / 64; p.ntiles = NB * p.tpi;
  p.m_total = M; p.nslices = si.nslices;
  const int grid = grid_for(p.ntiles, si);
  if (stat_part) {
    const int rows = grid / si.nslices * si.waves;
    WFAE_REQUIRE(stat_capacity >= 2 * (int64_t)rows * M, WFAE_ERR_WORKSPACE, "c1r_fwd: stat_part holds %lld doubles, needs %lld",
                 (long long)stat_capacity, (long long)(2 * (int64_t)rows * M));
    *stat_rows = rows;
    p.part0 = stat_part;
    p.part1 = stat_part + (long)rows * M;
  }
  hipStream_t st = (hipStream_t)stream;
  const bool pro = pro_scale != nullptr, stats = stat_part != nullptr;
  switch (sid) {
    case 0: launch_shape<4, 2, 2, 8, 2, 2>(p, pro, stats, grid, st); break;
    case 1: launch_shape<8, 4, 4, 8, 2, 2>(p, pro, stats, grid, st); break;
    case 2: launch_shape<1, 8, 1, 4, 2, 8>(p, pro, stats, grid, st); break;
    case 3: launch_shape<2, 16, 1, 4, 2, 2>(p, pro, stats, grid, st); break;
    case 4: launch_shape<4, 4, 4, 4, 4, 2, true>(p, pro, stats, grid, st); break;
    default: launch_shape<8, 4, 4, 4, 4, 2, true>(p, pro, stats, grid, st); break;
  }
  return check_launch("c1r_fwd");
}

int wfae_c1r_bnred_supported(int M, int K, int HW) {
  const int sid = shape_of(M, K).sid;
  return ((sid == 2 || sid == 3) && HW > 0 && HW % 64 == 0 && wfae::split_gemm_enabled()) ? 1 : 0;
}

int wfae_c1r_bnred(const float* w, int64_t w_sm, int64_t w_sk, const float* dt, const float* x, const float* bn_scale,
                   const float* bn_shift, const float* save_mean, const float* save_invstd, float* da, int NB, int K, int M, int HW,
                   double* part, int64_t part_capacity, int* part_rows, wfae_stream_t stream) {
  WFAE_REQUIRE(w && dt && x && bn_scale && bn_shift && save_mean && save_invstd && part && part_rows, WFAE_ERR_NULL_POINTER,
               "c1r_bnred: null pointer");   // da may be null: the sums alone (wfae_c1r_bndx recomputes dA)
  WFAE_REQUIRE(NB > 0 && K > 0 && M > 0 && HW > 0 && (int64_t)NB * HW < (1ll << 31), WFAE_ERR_BAD_SHAPE, "c1r_bnred: bad shape");
  const ShapeInfo si = shape_of(M, K);
  WFAE_REQUIRE(wfae_c1r_bnred_supported(M, K, HW), WFAE_ERR_UNSUPPORTED,
               "c1r_bnred: serves the widening data gradients (M, K) = (128, 32), (256, 64) with HW %% 64 == 0 at fp32 precision with "
               "the split GEMMs on (M %d, K %d, HW %d)", M, K, HW);
  WFAE_REQUIRE((w_sm == K && w_sk == 1) || (w_sm == 1 && w_sk == M), WFAE_ERR_BAD_SHAPE,
               "c1r_bnred: the weight is (M, K) row-major (strides K, 1) or its transpose (strides 1, M)");
  WFAE_REQUIRE(al16(dt) && al16(x) && (!da || al16(da)), WFAE_ERR_UNSUPPORTED, "c1r_bnred: tensors must be 16-byte aligned");
  C1RP p = {};
  p.W = w; p.w_sm = w_sm; p.w_sk = w_sk;
  p.X = dt; p.Y = da; p.bn_x = x;
  p.bn_tab[0] = bn_scale; p.bn_tab[1] = bn_shift; p.bn_tab[2] = save_mean; p.bn_tab[3] = save_invstd;
  p.HW = HW; p.tpi = HW / 64; p.ntiles = NB * p.tpi;
  p.m_total = M; p.nslices = 1;
  const int grid = grid_for(p.ntiles, si);
  const int rows = grid * BNR_WAVES;
  WFAE_REQUIRE(part_capacity >= 2 * (int64_t)rows * M, WFAE_ERR_WORKSPACE, "c1r_bnred: part holds %lld doubles, needs %lld",
               (long long)part_capacity, (long long)(2 * (int64_t)rows * M));
  *part_rows = rows;
  p.part0 = part;
  p.part1 = part + (long)rows * M;
  hipStream_t st = (hipStream_t)stream;
  // 8 waves (256 registers each, x four / one pass ahead) against 4 waves with the plain product's rings: sums alone 0.98 ->
  // 0.73 ms at C = 128, 0.60 -> 0.48 ms at C = 256 (profiles/r04_kbench_bn1_backward_recompute.txt)
  if (si.sid == 2) launch_bnr<1, 8, BNR_WAVES, 4>(p, grid, st);
  else launch_bnr<2, 16, BNR_WAVES, 2>(p, grid, st);
  return check_launch("c1r_bnred");
}

int wfae_c1r_bndx(const float* w, int64_t w_sm, int64_t w_sk, const float* dt, const float* x, const float* gamma, const float* bn_scale,
                  const float* bn_shift, const float* save_mean, const float* save_invstd, const float* coef, const float* res,
                  float* dx, int NB, int K, int M, int HW, int training, wfae_stream_t stream) {
  WFAE_REQUIRE(w && dt && x && gamma && bn_scale && bn_shift && save_mean && save_invstd && coef && dx, WFAE_ERR_NULL_POINTER,
               "c1r_bndx: null pointer");
  WFAE_REQUIRE(NB > 0 && K > 0 && M > 0 && HW > 0 && (int64_t)NB * HW < (1ll << 31), WFAE_ERR_BAD_SHAPE, "c1r_bndx: bad shape");
  const ShapeInfo si = shape_of(M, K);
  WFAE_REQUIRE(wfae_c1r_bnred_supported(M, K, HW), WFAE_ERR_UNSUPPORTED,
               "c1r_bndx: serves the widening data gradients (M, K) = (128, 32), (256, 64) with HW %% 64 == 0 at fp32 precision with "
               "the split GEMMs on (M %d, K %d, HW %d)", M, K, HW);
  WFAE_REQUIRE((w_sm == K && w_sk == 1) || (w_sm == 1 && w_sk == M), WFAE_ERR_BAD_SHAPE,
               "c1r_bndx: the weight is (M, K) row-major (strides K, 1) or its transpose (strides 1, M)");
  WFAE_REQUIRE(al16(dt) && al16(x) && al16(dx) && (!res || al16(res)), WFAE_ERR_UNSUPPORTED, "c1r_bndx: tensors must be 16-byte aligned");
  C1RP p = {};
  p.W = w; p.w_sm = w_sm; p.w_sk = w_sk;
  p.X = dt; p.Y = dx; p.bn_x = x; p.res = res;
  p.bn_tab[0] = bn_scale; p.bn_tab[1] = bn_shift; p.bn_tab[2] = save_mean; p.bn_tab[3] = save_invstd;
  p.bn_gamma = gamma; p.bn_coef = coef; p.inv_count = 1.0f / (float)((double)NB * HW); p.training = training;
  p.HW = HW; p.tpi = HW / 64; p.ntiles = NB * p.tpi;
  p.m_total = M; p.nslices = 1;
  const int grid = grid_for(p.ntiles, si);
  hipStream_t st = (hipStream_t)stream;
  // two rings (x and the skip gradient): four passes deep at C = 128 where the reduce form keeps one ring of eight
  // (8 waves with rings two deep: 1.50 against 1.47 ms at C = 128, and 162 spilled registers at C = 256)
  if (si.sid == 2) launch_bndx<1, 8, 4, 4>(p, grid, st);
  else launch_bndx<2, 16, 4, 2>(p, grid, st);
  return check_launch("c1r_bndx");
}

}  // extern "C"
