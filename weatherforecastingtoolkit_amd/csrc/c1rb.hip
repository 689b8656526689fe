// c1rb.hip — csrc/c1r.hip's register-direct 1x1 convolution on bf16-STORED tensors ('medium' precision, BASELINE config 5):
//   Y[img][m][p] (bf16) = sum_k A[m][k] f(X[img][k][p]) (+ res[img][m][p])      fp32 accumulation, one rounding of the result
// The activation goes HBM -> registers -> matrix core with no LDS staging and NO arithmetic on the way: a lane loads rows
// k0 + 8 (l >> 4) + i (i < 8) of a 32-row chunk as 16-byte pieces of EIGHT consecutive pixels 8 (l & 15) .. + 7; the B fragment
// of MFMA jj (jj = 0..7) is the jj-th bf16 of the eight pieces — four v_perm_b32 per fragment — and the column of MFMA jj that
// lane n handles is pixel 8 n + jj: the relabelling the accumulator layout undoes (a lane's accumulators [jj = 0..7][q] are eight
// consecutive pixels of row 4 (l >> 4) + q: one 16-byte bf16 store).  One v_mfma_f32_16x16x32_bf16 product, the weights as one
// bf16 plane (rounded by the block on the way into its LDS row image), 128-pixel wave tiles, everything else as c1r.hip:
// persistent one-block-per-CU grid, free-running waves, streaming ("narrowing") kernels with a register ring of chunks and
// B-resident ("widening") kernels with a ring of residual rows, M-slices per XCD where the weight image of a product exceeds
// the LDS, BatchNorm + GELU prologue (c1b.hip's arithmetic on bf16 pairs), residual, BatchNorm sums of the ROUNDED result.
// c1b.hip (LDS-tiled, ds_read_b64_tr_b16) holds 2.4 - 4.7 TB/s on these tensors (profiles/r03_v10_medium_*); it stays the
// kernel of the shapes this one does not serve (HW % 128 != 0: the 24 x 24 stage).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

using namespace wfae;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct C1RBP {
  const float* W;            // A[m][k] = W[m * w_sm + k * w_sk] (fp32 parameter)
  long w_sm, w_sk;
  const bf16_t* X;           // [NB][K][HW]
  bf16_t* Y;                 // [NB][m_total][HW]
  const bf16_t* res;         // [NB][m_total][HW] or null
  const float* pro_scale;    // PRO: folded BatchNorm scale / shift of the input channels [K]
  const float* pro_shift;
  double* part0;             // STATS: [groups * waves][m_total] sums, one row per wave
  double* part1;
  int HW;                    // % 128 == 0
  int tpi;                   // tiles per image
  int ntiles;
  int m_total, nslices;
};

__device__ __forceinline__ unsigned swz16(int r) { return (0x78u >> (((r >> 2) & 3) << 1)) & 3u; }

#define C1RB_DPP_F64(v, CTRL)                                                                                    \
  __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true),                       \
                   __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true))
__device__ __forceinline__ double row_sum16(double v) {
  v += C1RB_DPP_F64(v, 0x111);
  v += C1RB_DPP_F64(v, 0x112);
  v += C1RB_DPP_F64(v, 0x114);
  v += C1RB_DPP_F64(v, 0x118);
  return v;
}

// NBUF: chunk buffers of the streaming ring; RD: residual-row buffers of the B-resident ring; RESN: streaming kernel with a residual
template <int KCH, int MT, int MG, bool PRO, bool STATS, int RWAVES, int NBUF, int RD, bool RESN>
__global__ __launch_bounds__(64 * RWAVES, (RWAVES == 4 ? 1 : 2)) void c1rb_kernel(C1RBP p) {
  constexpr int RNT = 64 * RWAVES;
  constexpr int K = 32 * KCH, M = 16 * MT, NPASS = MT / MG;
  constexpr bool BRES = NPASS > 1;
  static_assert(MT % MG == 0, "whole passes");
  constexpr int PLANE_B = M * 64;                  // one 32-deep chunk: M rows x 64 bytes
  constexpr int A_B = KCH * PLANE_B;
  constexpr int PRO_B = PRO ? 2 * K * 4 : 0;
  constexpr int ST_B = STATS ? RWAVES * 2 * M * 8 : 0;
  // prologue forms (4-wave blocks, see wfae_c1rb_fwd): padded past half of the CU's LDS, so that a second block can never
  // share the CU — and with it the SIMDs — whatever the grid and the register count
  constexpr int PAD_B = (PRO && A_B + PRO_B + ST_B < 84 * 1024) ? 84 * 1024 - (A_B + PRO_B + ST_B) : 0;
  static_assert(!PRO || RWAVES == 4, "prologue forms: one wave per SIMD");
  static_assert(A_B + PRO_B + ST_B + PAD_B <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) unsigned char smem[A_B + PRO_B + ST_B + PAD_B];
  float* const lsc = reinterpret_cast<float*>(smem + A_B);
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n16 = lane & 15, kg = lane >> 4;
  double* const lst = reinterpret_cast<double*>(smem + A_B + PRO_B) + (STATS ? wave * 2 * M : 0);
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;      // M-slices: see c1r.hip
  const int slice = jb % p.nslices, group = (jb / p.nslices) * 8 + xcd;
  const int ngroups = gridDim.x / p.nslices;
  const int m_off = slice * M;

  // ---- the weights: rounded to bf16 on the way in, row image [chunk][m][64 B] (off_row16 of splitgemm.hip)
  {
    const bool kfast = p.w_sk == 1;
    for (int idx = t; idx < M * K / 8; idx += RNT) {
      int m, ch;
      if (kfast) { ch = idx % (K / 8); m = idx / (K / 8); } else { m = idx % M; ch = idx / M; }
      const float* src = p.W + (long)(m_off + m) * p.w_sm + (long)(8 * ch) * p.w_sk;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = src[(long)e * p.w_sk];
      const u32x4 h = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
      const unsigned off = (unsigned)((ch >> 2) * PLANE_B + m * 64) + ((((unsigned)ch & 3u) ^ swz16(m)) << 4);
      *reinterpret_cast<u32x4*>(smem + off) = h;
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
  }
  __syncthreads();

  const long rowB = (long)p.HW * 2;                                          // bytes between two channel rows
  const unsigned lane_in = (unsigned)((8 * n16 + 8 * kg * p.HW) * 2);       // this lane's piece of row 8 kg (+ i rows)
  const unsigned lane_out = (unsigned)((8 * n16 + 4 * kg * p.HW) * 2);      // result rows 4 kg + q
  const unsigned a_rd = (unsigned)(n16 * 64) + (((unsigned)kg ^ swz16(n16)) << 4);
  const int tstride = ngroups * RWAVES;
  int tile = group * RWAVES + wave;

  auto tile_off = [&](int tl, int chans) -> long {   // byte offset of a tile's first pixel in a [NB][chans][HW] bf16 tensor
    const int img = tl / p.tpi;
    return ((long)img * chans * p.HW + (long)(tl - img * p.tpi) * 128) * 2;
  };
  auto load_chunk = [&](u32x4 (&r)[8], int tl, int c) {
    const char* xb = reinterpret_cast<const char*>(p.X) + tile_off(tl, K) + (long)(32 * c) * rowB;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = *reinterpret_cast<const u32x4*>(xb + i * rowB + lane_in);
  };
  auto act2 = [&](unsigned w, float s, float h) -> unsigned {   // bn_act_fwd_kernel<GELU>'s arithmetic on a bf16 pair (c1b.hip)
    return pack_bf16(gelu_f(fmaf(bf16_lo(w), s, h)), gelu_f(fmaf(bf16_hi(w), s, h)));
  };
  auto pro_chunk = [&](u32x4 (&r)[8], int c) {
    if constexpr (PRO) {
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(lsc + 32 * c + 8 * kg), s1 = *reinterpret_cast<const f32x4*>(lsc + 32 * c + 8 * kg + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(lsc + K + 32 * c + 8 * kg), h1 = *reinterpret_cast<const f32x4*>(lsc + K + 32 * c + 8 * kg + 4);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float s = i < 4 ? s0[i & 3] : s1[i & 3], h = i < 4 ? h0[i & 3] : h1[i & 3];
        const unsigned w0 = r[i].x, w1 = r[i].y, w2 = r[i].z, w3 = r[i].w;
        r[i] = u32x4{act2(w0, s, h), act2(w1, s, h), act2(w2, s, h), act2(w3, s, h)};
      }
    }
  };
  // B fragments of pixel components jj0 .. jj0 + JW: b[j] = the eight k of this lane at pixel 8 n16 + jj0 + j
  auto frag_part = [&](const u32x4 (&r)[8], auto& b, auto jj0c, auto jwc) {
    constexpr int JJ0 = decltype(jj0c)::value, JW = decltype(jwc)::value;
#pragma unroll
    for (int j = 0; j < JW; ++j) {
      const int jj = JJ0 + j, d = jj >> 1;
      const unsigned sel = (jj & 1) ? 0x07060302u : 0x05040100u;   // {S1 half, S0 half}: low = row 2 ip, high = row 2 ip + 1
      unsigned o[4];
#pragma unroll
      for (int ip = 0; ip < 4; ++ip) {
        const unsigned lo = r[2 * ip][d], hi = r[2 * ip + 1][d];
        o[ip] = __builtin_amdgcn_perm(hi, lo, sel);
      }
      b[j] = u32x4{o[0], o[1], o[2], o[3]};
    }
  };
  auto multiply_part = [&](f32x4 (&acc)[MG][8], const auto& b, int c, int a_pass_off, auto jj0c, auto jwc) {
    constexpr int JJ0 = decltype(jj0c)::value, JW = decltype(jwc)::value;
    bf16x8 a[MG];
#pragma unroll
    for (int mt = 0; mt < MG; ++mt)
      a[mt] = *reinterpret_cast<const bf16x8*>(smem + a_pass_off + c * PLANE_B + mt * (16 * 64) + a_rd);
#pragma unroll
    for (int mt = 0; mt < MG; ++mt)
#pragma unroll
      for (int j = 0; j < JW; ++j)
        acc[mt][JJ0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], __builtin_bit_cast(bf16x8, b[j]), acc[mt][JJ0 + j], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I4 = std::integral_constant<int, 4>;
  using I8 = std::integral_constant<int, 8>;
  auto load_res = [&](u32x4 (&rv)[MG][4], int tl, int mt0) {
    const char* rb = reinterpret_cast<const char*>(p.res) + tile_off(tl, p.m_total) + (long)(m_off + 16 * mt0) * rowB;
#pragma unroll
    for (int mt = 0; mt < MG; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) rv[mt][q] = *reinterpret_cast<const u32x4*>(rb + (16 * mt + q) * rowB + lane_out);
  };
  // lane (n16, kg): acc[mt][jj][q] = Y[row 16 (mt0 + mt) + 4 kg + q][pixel 8 n16 + jj]
  auto store_pass = [&](const f32x4 (&acc)[MG][8], const u32x4 (&rv)[MG][4], bool with_res, int tl, int mt0) {
    char* yb = reinterpret_cast<char*>(p.Y) + tile_off(tl, p.m_total) + (long)(m_off + 16 * mt0) * rowB;
#pragma unroll
    for (int mt = 0; mt < MG; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v[jj] = acc[mt][jj][q];
        if (with_res) {
          const unsigned r0 = rv[mt][q].x, r1 = rv[mt][q].y, r2 = rv[mt][q].z, r3 = rv[mt][q].w;
          v[0] += bf16_lo(r0); v[1] += bf16_hi(r0); v[2] += bf16_lo(r1); v[3] += bf16_hi(r1);
          v[4] += bf16_lo(r2); v[5] += bf16_hi(r2); v[6] += bf16_lo(r3); v[7] += bf16_hi(r3);
        }
        const u32x4 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
        *reinterpret_cast<u32x4*>(yb + (16 * mt + q) * rowB + lane_out) = o;
        if constexpr (STATS) {   // sums of the ROUNDED values: fp32 sums of four, fp64 across (c1b.hip / chan_reduce_kernel)
          const unsigned o0 = o.x, o1 = o.y, o2 = o.z, o3 = o.w;
          const float a0 = bf16_lo(o0), a1 = bf16_hi(o0), a2 = bf16_lo(o1), a3 = bf16_hi(o1);
          const float b0 = bf16_lo(o2), b1 = bf16_hi(o2), b2 = bf16_lo(o3), b3 = bf16_hi(o3);
          const double s1 = (double)((a0 + a1) + (a2 + a3)) + (double)((b0 + b1) + (b2 + b3));
          const double s2 = (double)(fmaf(a0, a0, a1 * a1) + fmaf(a2, a2, a3 * a3)) +
                            (double)(fmaf(b0, b0, b1 * b1) + fmaf(b2, b2, b3 * b3));
          const double d1 = row_sum16(s1), d2 = row_sum16(s2);
          if (n16 == 15) {
            const int m = 16 * (mt0 + mt) + 4 * kg + q;
            lst[m] += d1;
            lst[M + m] += d2;
          }
        }
      }
  };

  const bool with_res = p.res != nullptr;
  if (tile < p.ntiles) {
    if constexpr (!BRES) {
      // ---- streaming: K chunks through a ring of NBUF register buffers, all M rows (of the slice) at once
      static_assert(KCH % NBUF == 0 && NBUF >= 2, "ring positions must repeat per tile");
      constexpr int D = NBUF - 1;
      u32x4 ring[NBUF][8];
#pragma unroll
      for (int c = 0; c < D; ++c) load_chunk(ring[c], tile, c);
      while (tile < p.ntiles) {
        const int nxt_tile = tile + tstride < p.ntiles ? tile + tstride : tile;   // past the end: a harmless re-read
        f32x4 acc[MG][8];
        u32x4 rv[MG][4];
#pragma unroll
        for (int mt = 0; mt < MG; ++mt)
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) acc[mt][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
          if (c + D < KCH) load_chunk(ring[(c + D) % NBUF], tile, c + D);
          else load_chunk(ring[(c + D) % NBUF], nxt_tile, c + D - KCH);
          if constexpr (RESN) {
            if (c == KCH - 1 && with_res) load_res(rv, tile, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          pro_chunk(ring[c % NBUF], c);
          if constexpr (MG >= 4) {   // 128 accumulator registers: the chunk in two halves of four pixel components
            u32x4 b4[4];
            frag_part(ring[c % NBUF], b4, I0{}, I4{});
            multiply_part(acc, b4, c, 0, I0{}, I4{});
            frag_part(ring[c % NBUF], b4, I4{}, I4{});
            multiply_part(acc, b4, c, 0, I4{}, I4{});
          } else {
            u32x4 b[8];
            frag_part(ring[c % NBUF], b, I0{}, I8{});
            multiply_part(acc, b, c, 0, I0{}, I8{});
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        store_pass(acc, rv, RESN && with_res, tile, 0);
        tile += tstride;
      }
    } else {
      // ---- B-resident: the tile's fragments stay in registers, M in passes of 16 MG rows, residual rows RD - 1 passes ahead
      static_assert(NPASS % RD == 0 && RD >= 2 && KCH <= RD, "ring positions must repeat per tile");
      u32x4 raw[KCH][8];
#pragma unroll
      for (int c = 0; c < KCH; ++c) load_chunk(raw[c], tile, c);
      while (tile < p.ntiles) {
        const int nxt_tile = tile + tstride < p.ntiles ? tile + tstride : tile;
        u32x4 rr[RD][MG][4];
        if (with_res) {
#pragma unroll
          for (int u = 0; u < RD - 1; ++u) load_res(rr[u], tile, u * MG);
        }
        __builtin_amdgcn_sched_barrier(0);
        u32x4 b[KCH][8];
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
          pro_chunk(raw[c], c);
          frag_part(raw[c], b[c], I0{}, I8{});
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int pass0 = 0; pass0 < NPASS; pass0 += RD) {
#pragma unroll
          for (int u = 0; u < RD; ++u) {
            const int pass = pass0 + u, mt0 = pass * MG;
            if (with_res && pass + RD - 1 < NPASS) load_res(rr[(u + RD - 1) % RD], tile, (pass + RD - 1) * MG);
            // the last KCH passes of the tile: one chunk each of the wave's next tile
            if (u >= RD - KCH && pass0 == NPASS - RD) load_chunk(raw[u >= RD - KCH ? u - (RD - KCH) : 0], nxt_tile, u - (RD - KCH));
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[MG][8];
#pragma unroll
            for (int mt = 0; mt < MG; ++mt)
#pragma unroll
              for (int jj = 0; jj < 8; ++jj) acc[mt][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < KCH; ++c) multiply_part(acc, b[c], c, mt0 * (16 * 64), I0{}, I8{});
            store_pass(acc, rr[u], with_res, tile, mt0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        tile += tstride;
      }
    }
  }
  if constexpr (STATS) {
    const long row = (long)group * RWAVES + wave;
    for (int i = lane; i < M; i += 64) {
      p.part0[row * p.m_total + m_off + i] = lst[i];
      p.part1[row * p.m_total + m_off + i] = lst[M + i];
    }
  }
}

// served (M, K) -> kernel shape.  One block owns all rows where the one-plane weight image fits the LDS (<= 128 KiB);
// otherwise the blocks of one XCD split the rows in slices (c1r.hip).
struct ShapeInfo { int sid, nslices, waves; };
inline ShapeInfo shape_of(int M, int K) {
  if (M == 32 && K == 128) return {0, 1, 8};
  if (M == 64 && K == 256) return {1, 1, 8};
  if (M == 128 && K == 32) return {2, 1, 8};
  if (M == 256 && K == 64) return {3, 1, 4};
  if (M == 128 && K == 512) return {4, 2, 8};      // streaming, 2 slices of 64 rows
  if (M == 512 && K == 128) return {5, 2, 4};      // B-resident (128 fragment registers), 2 slices of 256 rows (64 KiB each)
  if (M == 256 && K == 1024) return {6, 4, 8};     // streaming, 4 slices of 64 rows (128 KiB each)
  if (M == 1024 && K == 256) return {7, 16, 4};    // streaming with residual, 16 slices of 64 rows
  return {-1, 1, 8};
}

inline int num_cus() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();
  return n;
}

inline int grid_for(long ntiles, const ShapeInfo& si) {
  if (si.nslices > 1) {
    const int unit = 8 * si.nslices;
    const int g = num_cus() / unit * unit;
    return g > 0 ? g : unit;
  }
  const long g = (ntiles + si.waves - 1) / si.waves;
  return (int)(g < num_cus() ? g : num_cus());
}

// (the prologue forms exist for 4-wave blocks only: see wfae_c1rb_fwd)
template <int KCH, int MT, int MG, int NW, int NBUF, int RD, bool RESN = false>
void launch_shape(const C1RBP& p, bool pro, bool stats, int grid, hipStream_t st) {
  const dim3 g((unsigned)grid), b(64 * NW);
  if constexpr (NW == 4) {
    if (pro && stats) { hipLaunchKernelGGL((c1rb_kernel<KCH, MT, MG, true, true, NW, NBUF, RD, RESN>), g, b, 0, st, p); return; }
    if (pro) { hipLaunchKernelGGL((c1rb_kernel<KCH, MT, MG, true, false, NW, NBUF, RD, RESN>), g, b, 0, st, p); return; }
  }
  if (stats) hipLaunchKernelGGL((c1rb_kernel<KCH, MT, MG, false, true, NW, NBUF, RD, RESN>), g, b, 0, st, p);
  else hipLaunchKernelGGL((c1rb_kernel<KCH, MT, MG, false, false, NW, NBUF, RD, RESN>), g, b, 0, st, p);
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

extern "C" {

int wfae_c1rb_supported(int M, int K, int HW) {
  return (shape_of(M, K).sid >= 0 && HW > 0 && HW % 128 == 0 && wfae::matmul_precision() == WFAE_PRECISION_BF16) ? 1 : 0;
}

int wfae_c1rb_stat_rows(int M, int K, int NB, int HW) {
  const ShapeInfo si = shape_of(M, K);
  if (si.sid < 0 || HW <= 0 || HW % 128 != 0 || NB <= 0) return 0;
  return grid_for((long)NB * (HW / 128), si) / si.nslices * si.waves;
}

int wfae_c1rb_fwd(const float* w, int64_t w_sm, int64_t w_sk, const uint16_t* x, const float* pro_scale, const float* pro_shift,
                  const uint16_t* res, uint16_t* y, int NB, int K, int M, int HW, double* stat_part, int64_t stat_capacity,
                  int* stat_rows, wfae_stream_t stream) {
  WFAE_REQUIRE(w && x && y, WFAE_ERR_NULL_POINTER, "c1rb_fwd: null pointer");
  WFAE_REQUIRE(NB > 0 && K > 0 && M > 0 && HW > 0 && (int64_t)NB * HW < (1ll << 31), WFAE_ERR_BAD_SHAPE, "c1rb_fwd: bad shape");
  const ShapeInfo si = shape_of(M, K);
  WFAE_REQUIRE(si.sid >= 0 && HW % 128 == 0, WFAE_ERR_UNSUPPORTED,
               "c1rb_fwd: serves the Bottleneck products (M, K) = (C/4, C), (C, C/4), C = 128 .. 1024, with HW %% 128 == 0 "
               "(M %d, K %d, HW %d)", M, K, HW);
  WFAE_REQUIRE(wfae::matmul_precision() == WFAE_PRECISION_BF16, WFAE_ERR_UNSUPPORTED, "c1rb_fwd: needs WFAE_PRECISION_BF16");
  WFAE_REQUIRE((w_sm == K && w_sk == 1) || (w_sm == 1 && w_sk == M), WFAE_ERR_BAD_SHAPE,
               "c1rb_fwd: the weight is (M, K) row-major (strides K, 1) or its transpose (strides 1, M)");
  WFAE_REQUIRE(al16(x) && al16(y) && (!res || al16(res)), WFAE_ERR_UNSUPPORTED, "c1rb_fwd: tensors must be 16-byte aligned");
  WFAE_REQUIRE(!res || M > K, WFAE_ERR_UNSUPPORTED, "c1rb_fwd: the residual add belongs to the widening products (M > K)");
  WFAE_REQUIRE((pro_scale != nullptr) == (pro_shift != nullptr) && (stat_part != nullptr) == (stat_rows != nullptr),
               WFAE_ERR_NULL_POINTER, "c1rb_fwd: scale / shift and stat_part / stat_rows go together");
  C1RBP p = {};
  p.W = w; p.w_sm = w_sm; p.w_sk = w_sk;
  p.X = x; p.Y = y; p.res = res;
  p.pro_scale = pro_scale; p.pro_shift = pro_shift;
  p.HW = HW; p.tpi = HW / 128; p.ntiles = NB * p.tpi;
  p.m_total = M; p.nslices = si.nslices;
  const bool pro = pro_scale != nullptr, stats = stat_part != nullptr;
  // The prologue forms run ONE wave per SIMD.  With two (the 8-wave blocks of the shapes below) identical launches returned
  // different results on gfx950, about one launch in ten with operands in L2 and one in two with cold ones: in one tile, the
  // LOW bf16 of one dword of the activated operand wrong in most lanes (tools/debug_c1rb_repeat.py; csrc/c1b.hip, which
  // evaluates the same prologue, and the fp32 kernels of csrc/c1r.hip stayed bit-stable under the same stress).  Taking
  // v_cmp / v_cndmask out of gelu_f (common.h) removed the warm failures, a full s_waitcnt before the prologue, s_nops
  // around the MFMAs or in front of v_cvt_pk_bf16_f32 removed nothing; with 4 waves per block: 0 of 320 cold launches.
  // Cause not established — the evidence says an interaction of two waves on one SIMD in this instruction mix.
  const bool pro4 = pro && si.waves == 8;
  ShapeInfo sl = si;
  if (pro4) sl.waves = 4;
  const int grid = grid_for(p.ntiles, sl);
  if (stat_part) {
    const int rows = grid / si.nslices * sl.waves;
    WFAE_REQUIRE(stat_capacity >= 2 * (int64_t)rows * M, WFAE_ERR_WORKSPACE, "c1rb_fwd: stat_part holds %lld doubles, needs %lld",
                 (long long)stat_capacity, (long long)(2 * (int64_t)rows * M));
    *stat_rows = rows;
    p.part0 = stat_part;
    p.part1 = stat_part + (long)rows * M;
  }
  hipStream_t st = (hipStream_t)stream;
  if (pro4) {   // (the streaming shapes with four chunk buffers instead of two: the loads of a CU stay in flight with half the waves)
    switch (si.sid) {
      case 0: launch_shape<4, 2, 2, 4, 4, 2>(p, pro, stats, grid, st); break;
      case 1: launch_shape<8, 4, 4, 4, 4, 2>(p, pro, stats, grid, st); break;
      case 2: launch_shape<1, 8, 1, 4, 2, 4>(p, pro, stats, grid, st); break;
      case 4: launch_shape<16, 4, 4, 4, 4, 2>(p, pro, stats, grid, st); break;
      default: launch_shape<32, 4, 4, 4, 4, 2>(p, pro, stats, grid, st); break;
    }
    return check_launch("c1rb_fwd");
  }
  switch (si.sid) {
    case 0: launch_shape<4, 2, 2, 8, 2, 2>(p, pro, stats, grid, st); break;
    case 1: launch_shape<8, 4, 4, 8, 2, 2>(p, pro, stats, grid, st); break;
    case 2: launch_shape<1, 8, 1, 8, 2, 4>(p, pro, stats, grid, st); break;
    case 3: launch_shape<2, 16, 1, 4, 2, 4>(p, pro, stats, grid, st); break;
    case 4: launch_shape<16, 4, 4, 8, 2, 2>(p, pro, stats, grid, st); break;
    case 5: launch_shape<4, 16, 1, 4, 2, 4>(p, pro, stats, grid, st); break;
    case 6: launch_shape<32, 4, 4, 8, 2, 2>(p, pro, stats, grid, st); break;
    default: launch_shape<8, 4, 4, 4, 4, 2, true>(p, pro, stats, grid, st); break;
  }
  return check_launch("c1rb_fwd");
}

}  // extern "C"
