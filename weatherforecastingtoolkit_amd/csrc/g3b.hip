// g3b.hip — the Bottleneck's grouped 3x3 'same' convolution (pipeline/models/ae_64x8x8_lin.py:17), forward and data
// gradient, in the bf16-STORAGE mode ('medium' precision, BASELINE config 5) as an implicit GEMM on the bf16 matrix pipe.
//
// dconv.hip serves these shapes through fp32 LDS patches: a VALU kernel at 4 / 8 channels per group (36 / 72 FMAs per output
// element) and v_mfma_f32_*_f32 tiles at 16 / 32 — with bf16 tensors both stay bound by fp32 arithmetic and by a patch staging
// that moves one 2-byte element per lane (1.0 - 1.4 TB/s, profiles/r03_v6_medium_*).  Here a bf16 value never changes format
// between HBM and the matrix core:
//   y[oc][p] = sum_{kx} P_kx[oc][p + kx - 1],     P_kx[oc][q] = sum_{ky, ci} w[oc][ci][ky][kx] x[ci][q + (ky - 1) W]
//   * the x shift of a tap is taken on the OUTPUT side: each P_kx is accumulated over 16-pixel blocks q that are ALIGNED in
//     the input row, so the B operand (k = (ky, ci), n = 16 pixels) is read from an LDS image that is laid out exactly like
//     the tensor ([channel][row][x], 16-byte global loads stored as they are) through ds_read_b64_tr_b16, whose 4 x 16
//     blocks must start on a multiple of four columns; the ky shift is a whole LDS row.  The three partial tiles are combined
//     with two DPP row shifts (the MFMA's C layout puts the 16 pixels of a tile in the 16 lanes of a DPP row); the lane that
//     falls off a tile takes the neighbour tile's value, which the same wave computes next (or, at the two ends of a wave's
//     run of tiles, with the one partial product it needs of the neighbour tile);
//   * a block owns a slab of 16 (32 at 32 channels per group) channels of one image over the FULL image width — no column
//     halo — and walks down a strip of rows with a ring of rows in LDS: every input row is loaded once per strip, row
//     r + 2 travels HBM -> registers while row r is multiplied, one barrier per step;
//   * A operand: the weights of the slab as ready-made MFMA fragments (bf16, block-diagonal over the groups that share a
//     16-channel tile: 4 / 8 channels per group use a quarter / half of it, which is still far from binding — the kernel is
//     HBM-bound), written once per call by g3b_pack_kernel, held in registers for the whole strip;
//   * results leave through wave-private LDS slices as 16-byte pieces of 8 bf16.
// MFMA k order inside a 32-deep step (it only has to be the same for both operands): k = 8 g + 4 h + q with g the lane
// group, h the transposed read, q the row of its 4 x 16 block  ->  channel 8 h + 4 (g & 1) + q of tap row 2 s + (g >> 1)
// (16-channel slabs: two tap rows per step, the fourth is a zero weight) or channel 16 (g >> 1) + 8 h + 4 (g & 1) + q of tap
// row s (32-channel slabs): the two lane groups of a 32-lane LDS access then read 8 CONSECUTIVE channels, and a channel stride
// of an odd multiple of 32 bytes spreads them over all 64 banks.
#include "common.h"

using namespace wfae;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// eight consecutive pixels of one channel row in registers: bf16 tensors one 16-byte piece, fp32 tensors two
template <typename T> struct G3Piece;
template <> struct G3Piece<bf16_t> { u32x4 v; };
template <> struct G3Piece<float> { u32x4 lo, hi; };
__device__ __forceinline__ void g3_load(G3Piece<bf16_t>& r, const bf16_t* p, bool ok) {
  const u32x4 v = *reinterpret_cast<const u32x4*>(p);
  r.v = ok ? v : u32x4{0u, 0u, 0u, 0u};
}
__device__ __forceinline__ void g3_load(G3Piece<float>& r, const float* p, bool ok) {
  const u32x4 a = *reinterpret_cast<const u32x4*>(p), b = *reinterpret_cast<const u32x4*>(p + 4);
  r.lo = ok ? a : u32x4{0u, 0u, 0u, 0u};
  r.hi = ok ? b : u32x4{0u, 0u, 0u, 0u};
}
// planes of the piece as MFMA-ready 16-byte rows: bf16 tensors the value itself, fp32 tensors h + m + l == x (common.h split3)
__device__ __forceinline__ void g3_planes(const G3Piece<bf16_t>& r, u32x4 (&pl)[1]) { pl[0] = r.v; }
__device__ __forceinline__ void g3_planes(const G3Piece<float>& r, u32x4 (&pl)[3]) {
  unsigned short h[8], m[8], l[8];
  const unsigned u[8] = {r.lo[0], r.lo[1], r.lo[2], r.lo[3], r.hi[0], r.hi[1], r.hi[2], r.hi[3]};
#pragma unroll
  for (int i = 0; i < 8; ++i) split3(__uint_as_float(u[i]), h[i], m[i], l[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pl[0][i] = (unsigned)h[2 * i] | ((unsigned)h[2 * i + 1] << 16);
    pl[1][i] = (unsigned)m[2 * i] | ((unsigned)m[2 * i + 1] << 16);
    pl[2][i] = (unsigned)l[2 * i] | ((unsigned)l[2 * i + 1] << 16);
  }
}


template <int CPG, int TPR, int W_, int NP, int TPW, typename T>
struct G3B {
  static constexpr int SC = CPG == 32 ? 32 : 16;   // channels of a slab
  static constexpr int MT = SC / 16;               // 16-row tiles of output channels
  static constexpr int KS = CPG == 32 ? 3 : 2;     // 32-deep K-steps per kx
  static constexpr int RI = 4 * TPW / TPR;         // rows per step: TPW tiles of 16 pixels per wave, four waves
  static constexpr int P = 16 * TPR;               // LDS row pitch in pixels (>= W_; the rest stays zero)
  // ring rows: bf16 tensors keep the incoming rows in their own slots (one barrier per step); fp32 tensors (three planes:
  // 3x the bytes) overwrite the rows that just died (two barriers per step)
  static constexpr bool TWO_BAR = NP != 1;
  static constexpr int R = TWO_BAR ? RI + 2 : 2 * RI + 2;
  static constexpr int RS = P * 2;                 // bytes
  static constexpr int CS0 = R * RS;
  static constexpr int CS = (CS0 / 32) % 2 ? CS0 : CS0 + 32;   // odd multiple of 32 bytes
  static constexpr int PLANE_B = SC * CS;
  static constexpr int RING_B = NP * PLANE_B;
  static constexpr int ES = sizeof(T);
  static constexpr int OB_RS = TPW * 16 * ES + 16; // + 16: the four lane groups write different bank windows
  static constexpr int OB_B = MT * 16 * OB_RS;     // per wave
  static constexpr int PPT = ES;                   // 16-byte pieces of a tile row: 2 (bf16) / 4 (fp32)
  static constexpr int PPR = TPW * PPT;            // ... of a wave's run
  static constexpr int XC = W_ / 8;                // 8-pixel pieces per channel row
  static constexpr int CH_ROW = SC * XC;
  static constexpr int NLD = (RI * CH_ROW + 255) / 256;
  static constexpr int SEG = TPR < TPW ? TPR : TPW;   // tiles of one row in a wave's run
  static constexpr int NSEG = TPW / SEG;
  static_assert((4 * TPW) % TPR == 0 && RI >= 1 && W_ % 8 == 0 && W_ <= P && P - W_ < 16, "geometry");
  static_assert(RING_B + 4 * OB_B <= 160 * 1024, "LDS");
};

__device__ __forceinline__ float dpp_shr1(float cur, float prev) {   // lane i <- cur[i - 1], lane 0 <- prev[15] (16-lane rows)
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(prev), 0x121, 0xf, 0xf, false);          // row_ror:1
  return __int_as_float(__builtin_amdgcn_update_dpp(t, __float_as_int(cur), 0x111, 0xf, 0xf, false));  // row_shr:1
}
__device__ __forceinline__ float dpp_shl1(float cur, float next) {   // lane i <- cur[i + 1], lane 15 <- next[0]
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(next), 0x12f, 0xf, 0xf, false);          // row_ror:15
  return __int_as_float(__builtin_amdgcn_update_dpp(t, __float_as_int(cur), 0x101, 0xf, 0xf, false));  // row_shl:1
}
__device__ __forceinline__ void g3_put(bf16_t* p, float v) { *p = (bf16_t)(pack_bf16(v, 0.f) & 0xffffu); }
__device__ __forceinline__ void g3_put(float* p, float v) { *p = v; }

// T = bf16_t, NP = 1: bf16 tensors.  T = float, NP = 3: fp32 tensors, every value split exactly into three bf16 planes at the
// LDS store and the weights likewise, six plane products per fp32 product (the arithmetic of splitgemm.hip: fp32 accuracy)
template <int CPG, int TPR, int W_, int NP, int TPW, typename T>
__global__ __launch_bounds__(256) void g3b_kernel(const T* __restrict__ x, const bf16x8* __restrict__ wpk,
                                                  T* __restrict__ y, int C, int H, int RH) {
  using G = G3B<CPG, TPR, W_, NP, TPW, T>;
  constexpr int SC = G::SC, MT = G::MT, KS = G::KS, RI = G::RI, R = G::R, RS = G::RS, CS = G::CS;
  __shared__ __attribute__((aligned(16))) unsigned char ring[G::RING_B];
  __shared__ __attribute__((aligned(16))) unsigned char obuf[4 * G::OB_B];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int strip = blockIdx.x, vg = blockIdx.y, n = blockIdx.z;
  const int y0 = strip * RH;
  const long HW = (long)H * W_;
  const T* __restrict__ xg = x + ((long)n * C + (long)vg * SC) * HW;
  T* __restrict__ yg = y + ((long)n * C + (long)vg * SC) * HW;

  if constexpr (G::P != W_) {   // pad columns are never written again
    for (int i = t * 16; i < G::RING_B; i += 256 * 16) *reinterpret_cast<u32x4*>(ring + i) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
  }

  // ---- weights: MFMA A fragments of this slab, in registers for the whole strip
  bf16x8 afr[3][KS][MT][NP];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
          afr[kx][s][mt][pl] = wpk[(((((long)vg * 3 + kx) * KS + s) * MT + mt) * NP + pl) * 64 + lane];

  // ---- staging: rows [first, first + nrows) of the slab, 8-pixel pieces, as they lie in memory; rows outside the image are zero
  G3Piece<T> rg[G::NLD];
  auto load_rows = [&](int first, int nrows) {
#pragma unroll
    for (int j = 0; j < G::NLD; ++j) {
      const int id = t + 256 * j;
      const int ri = id / G::CH_ROW, rem = id - ri * G::CH_ROW;
      const int ci = rem / G::XC, xc = rem - ci * G::XC;
      const int row = first + ri;
      const bool ok = id < nrows * G::CH_ROW && row >= 0 && row < H;
      g3_load(rg[j], xg + (ok ? (long)ci * HW + (long)row * W_ + xc * 8 : 0), ok);
    }
  };
  auto store_rows = [&](int first, int nrows) {
#pragma unroll
    for (int j = 0; j < G::NLD; ++j) {
      const int id = t + 256 * j;
      const int ri = id / G::CH_ROW, rem = id - ri * G::CH_ROW;
      const int ci = rem / G::XC, xc = rem - ci * G::XC;
      const int slot = (first + ri - y0 + 1) % R;
      if (id < nrows * G::CH_ROW) {
        u32x4 pl[NP];
        g3_planes(rg[j], pl);
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x4*>(ring + q * G::PLANE_B + ci * CS + slot * RS + xc * 16) = pl[q];
      }
    }
  };

  // ---- per-lane pieces of the transposed-read addresses
  const int g4 = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
  const int ci0 = (CPG == 32 ? 16 * (g4 >> 1) : 0) + 4 * (g4 & 1) + tq;   // h = 0; h = 1 is 8 channels on
  const unsigned lane_off = (unsigned)(ci0 * CS + 8 * tp);
  const bool ky_hi = (g4 >> 1) != 0;   // 16-channel slabs: this lane group takes the second tap row of a step

  // B fragments of the tile (row index `ridx` into the ring = row - y0 + 1 of tap row 0, column block xb)
  auto read_b = [&](bf16x8 (&b)[KS][NP], int ridx, int xb) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      int ky_lo = CPG == 32 ? s : 2 * s, ky_up = CPG == 32 ? s : (2 * s + 1 > 2 ? 2 : 2 * s + 1);
      const unsigned rlo = (unsigned)(((ridx + ky_lo) % R) * RS), rup = (unsigned)(((ridx + ky_up) % R) * RS);
      const unsigned roff = CPG == 32 ? rlo : (ky_hi ? rup : rlo);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
        const unsigned char* a = ring + pl * G::PLANE_B + lane_off + roff + xb * 32;
        const s16x4 p0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a));
        const s16x4 p1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a + 8 * CS));
        b[s][pl] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(p0, p1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    }
  };
  auto partial = [&](f32x4 (&acc)[MT], const bf16x8 (&b)[KS][NP], int kx) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if constexpr (NP == 3) {   // smallest terms first
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kx][s][mt][2], b[s][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kx][s][mt][0], b[s][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kx][s][mt][1], b[s][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kx][s][mt][1], b[s][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kx][s][mt][0], b[s][1], c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[kx][s][mt][0], b[s][0], c, 0, 0, 0);
      }
      acc[mt] = c;
    }
  };

  // ---- prologue: rows y0 - 1 .. y0 + RI
  for (int r = y0 - 1; r <= y0 + RI; r += RI) {
    const int nr = min(RI, y0 + RI - r + 1);
    load_rows(r, nr);
    store_rows(r, nr);
  }
  __syncthreads();

  unsigned char* ob = obuf + wave * G::OB_B;
  const int iters = RH / RI;
  for (int it = 0; it < iters; ++it) {
    const int r0 = y0 + it * RI;
    const bool more = it + 1 < iters;
    if (more) load_rows(r0 + RI + 1, RI);

#pragma unroll
    for (int sg = 0; sg < G::NSEG; ++sg) {
      const int idx0 = TPW * wave + sg * G::SEG;
      const int rowi = idx0 / TPR, xa = idx0 - rowi * TPR;   // row inside the step, first column block
      const int ridx = it * RI + rowi;                        // ring index of tap row 0 (input row r0 + rowi - 1)
      f32x4 prev0[MT], pend[MT], pend2[MT];
      bf16x8 b[KS][NP];
      // the P_0 of the block left of the run feeds pixel 0 of the first tile (zero at the image border)
      if (xa > 0) {
        read_b(b, ridx, xa - 1);
        partial(prev0, b, 0);
      } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) prev0[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i <= G::SEG; ++i) {
        f32x4 p0[MT], p1[MT], p2[MT];
        const bool inside = i < G::SEG;
        if (inside || xa + G::SEG < TPR) {
          read_b(b, ridx, xa + i);
          partial(p2, b, 2);
          if (inside) {
            partial(p0, b, 0);
            partial(p1, b, 1);
          }
        } else {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) p2[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (i > 0) {   // tile i - 1 is complete: its P_2 shifted left takes pixel 0 of this tile's P_2
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float v = pend[mt][q] + dpp_shl1(pend2[mt][q], p2[mt][q]);
              const int oc = mt * 16 + 4 * g4 + q;
              g3_put(reinterpret_cast<T*>(ob + oc * G::OB_RS) + (sg * G::SEG + i - 1) * 16 + i16, v);
            }
        }
        if (inside) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) pend[mt][q] = p1[mt][q] + dpp_shr1(p0[mt][q], prev0[mt][q]);
            prev0[mt] = p0[mt];
            pend2[mt] = p2[mt];
          }
        }
      }
    }

    // ---- the wave's tiles leave as 16-byte pieces: MT x 16 channel rows x PPR pieces
#pragma unroll
    for (int j = 0; j < (MT * 16 * G::PPR + 63) / 64; ++j) {
      const int id = lane + 64 * j;
      const int orow = id / G::PPR, ch = id - orow * G::PPR;
      const int idx = TPW * wave + ch / G::PPT;
      const int rowi = idx / TPR, xb = idx - rowi * TPR;
      const int px = xb * 16 + (16 / G::ES) * (ch % G::PPT);
      if (id < MT * 16 * G::PPR) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(ob + orow * G::OB_RS + ch * 16);
        if (px < W_) *reinterpret_cast<u32x4*>(yg + (long)orow * HW + (long)(r0 + rowi) * W_ + px) = v;
      }
    }

    if constexpr (G::TWO_BAR) __syncthreads();   // every wave is done with the rows that the incoming ones replace
    if (more) store_rows(r0 + RI + 1, RI);
    __syncthreads();
  }
}

// wpk[(((((vg * 3 + kx) * KS + s) * MT + mt) * NP + plane) * 64 + lane) * 8 + j]: A[m = lane & 15][k = 8 (lane >> 4) + j]
//   forward:    out channel co, in channel ci:  w[co][ci - group base][ky][kx]
//   transposed: (data gradient) the kernel's "out" channel is the convolution's input channel: w[ci][co - group base][2 - ky][2 - kx]
// NP = 1: the weight rounded to bf16 ('medium' operands); NP = 3: its exact three-plane split
template <int CPG, int NP>
__global__ void g3b_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ wpk, int C, int transposed) {
  constexpr int SC = CPG == 32 ? 32 : 16, MT = SC / 16, KS = CPG == 32 ? 3 : 2;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)(C / SC) * 3 * KS * MT * 64 * 8;
  if (i0 >= total) return;
  const int j = (int)(i0 & 7), lane = (int)((i0 >> 3) & 63);
  long r = i0 >> 9;
  const long frag = r;   // ((vg * 3 + kx) * KS + s) * MT + mt
  const int mt = (int)(r % MT); r /= MT;
  const int s = (int)(r % KS); r /= KS;
  const int kx = (int)(r % 3);
  const int vg = (int)(r / 3);
  const int m = lane & 15, g4 = lane >> 4, h = j >> 2, q = j & 3;
  const int oc = mt * 16 + m;
  const int ci = CPG == 32 ? 16 * (g4 >> 1) + 8 * h + 4 * (g4 & 1) + q : 8 * h + 4 * (g4 & 1) + q;
  const int ky = CPG == 32 ? s : 2 * s + (g4 >> 1);
  float v = 0.f;
  if (ky <= 2 && oc / CPG == ci / CPG) {
    const int co = vg * SC + oc, cin = vg * SC + ci, gb = (co / CPG) * CPG;
    v = transposed ? w[((long)cin * CPG + (co - gb)) * 9 + (2 - ky) * 3 + (2 - kx)]
                   : w[((long)co * CPG + (cin - gb)) * 9 + ky * 3 + kx];
  }
  bf16_t pl[3];
  if constexpr (NP == 1) pl[0] = (bf16_t)(pack_bf16(v, 0.f) & 0xffffu);
  else split3(v, pl[0], pl[1], pl[2]);
#pragma unroll
  for (int p = 0; p < NP; ++p) wpk[((frag * NP + p) * 64 + lane) * 8 + j] = pl[p];
}

template <int CPG, int TPR, int W_, int NP, int TPW, typename T>
int g3b_launch(const T* x, const float* w, T* y, int NB, int C, int H, int RH, int transposed, void* ws, size_t ws_bytes,
               hipStream_t st) {
  using G = G3B<CPG, TPR, W_, NP, TPW, T>;
  const int slabs = C / G::SC;
  const long frag_elems = (long)slabs * 3 * G::KS * G::MT * 64 * 8;
  WFAE_REQUIRE(ws && ws_bytes >= (size_t)frag_elems * 2 * NP, WFAE_ERR_WORKSPACE, "g3b_fwd: workspace %zu < %zu", ws_bytes,
               (size_t)frag_elems * 2 * NP);
  hipLaunchKernelGGL((g3b_pack_kernel<CPG, NP>), dim3((unsigned)cdiv(frag_elems, 256)), dim3(256), 0, st, w, (bf16_t*)ws, C,
                     transposed);
  int rc = check_launch("g3b_pack");
  if (rc) return rc;
  hipLaunchKernelGGL((g3b_kernel<CPG, TPR, W_, NP, TPW, T>), dim3(H / RH, slabs, NB), dim3(256), 0, st, x, (const bf16x8*)ws, y,
                     C, H, RH);
  return check_launch("g3b");
}

// the (channels per group, width) pairs of the model; rows per strip: a multiple of the rows per step that divides H
int g3b_strip(int cpg, int H, int W) {
  int ri;
  if (cpg == 4 && W == 384) ri = 1;
  else if (cpg == 8 && W == 192) ri = 2;
  else if (cpg == 16 && W == 96) ri = 4;
  else if (cpg == 32 && W == 48) ri = 8;
  else if (cpg == 32 && W == 24) ri = 12;
  else return 0;
  for (int rh = 24; rh >= ri; rh -= ri)
    if (rh % ri == 0 && H % rh == 0) return rh;
  return 0;
}


// =====================================================================================================================
// Weight gradient of the same convolution on bf16 tensors:
//   dW[oc][ci][ky][kx] = sum_q dy[oc][q - (kx - 1)] x[ci][q + (ky - 1) P]        (q over the padded pixel slots of a strip)
// The contraction runs over pixels, which are the contiguous index of BOTH tensors: both are staged as they lie in memory
// (16-byte pieces) into chunk-major LDS images  [32-pixel chunk][channel][64 bytes]  — per chunk exactly the A / B row image
// of splitgemm.hip (16-byte unit XOR g(channel >> 2): conflict-free ds_read_b128 fragments) — with a row pitch P that is a
// multiple of 32 pixels and >= W + 8, the pad columns zero in both images:
//   * the ky shift of a tap is then a whole number of chunks of the x image;
//   * the kx shift is taken on the dy fragment in registers: one 16-byte read plus the two neighbouring dwords, two funnel
//     shifts of four dwords (v_alignbit) give dy[q + 1 ..] and dy[q - 1 ..] — one fragment read serves all nine taps;
//     the zero pad between two rows is what the shifted fragment reads at a row end (the 'same' padding of the convolution).
// One v_mfma_f32_16x16x32_bf16 per tap and 32 pixels.  A block owns a 16- (32-) channel slab of one image and a strip of
// rows, rows travel through LDS once (x with one halo row either side), next rows HBM -> registers during the products.
// 4 / 8 / 16 channels per group: the four waves split the chunks and their nine accumulator tiles are added through LDS at
// the end (fixed order); 32 per group: wave = (oc tile, ci tile).  The block's partial goes to a slab, slab_reduce adds the
// slabs in fixed order (deterministic).
// =====================================================================================================================
template <int CPG, int W_, int NP>
struct G3W {
  static constexpr int SC = CPG == 32 ? 32 : 16;
  static constexpr int P = (W_ + 8 + 31) / 32 * 32;
  static constexpr int CPRW = P / 32;                                   // chunks per row
  // rows per step: bf16 tensors 12 - 16 chunks per step; fp32 tensors (three planes: 3x the LDS) as few rows as give the
  // four waves a chunk each
  static constexpr int RI = NP == 1 ? (W_ >= 384 ? 1 : W_ >= 192 ? 2 : W_ >= 96 ? 4 : W_ >= 48 ? 8 : 12)
                                    : (W_ >= 192 ? 1 : W_ >= 96 ? 2 : W_ >= 48 ? 1 : 2);
  static constexpr int RX = RI + 2, RD = RI;
  static constexpr int CHB = SC * 64;                                   // bytes of a chunk
  static constexpr int XB = RX * CPRW * CHB, DB = (RD * CPRW + 2) * CHB;  // per plane; dy image with a zero guard chunk at either end
  static constexpr int XC = W_ / 8;
  static constexpr int NLD = (RI * SC * XC + 255) / 256;
  static constexpr int NCH = RI * CPRW;                                 // dy chunks per step
  static_assert(NP * XB >= 2 * 9 * 256 * 4, "the x image doubles as the cross-wave reduction buffer");
  static_assert(NP * (XB + DB) <= 160 * 1024, "LDS");
};

__device__ __forceinline__ unsigned g3w_sw(int ch) { return (unsigned)((0x78 >> (((ch >> 2) & 3) << 1)) & 3); }

template <int CPG, int W_, int NP, typename T>
__global__ __launch_bounds__(256) void g3bw_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                   float* __restrict__ part, int C, int H, int RH, long out_elems) {
  using G = G3W<CPG, W_, NP>;
  constexpr int SC = G::SC, RI = G::RI, RX = G::RX, CPRW = G::CPRW, CHB = G::CHB, XB = G::XB, DB = G::DB;
  __shared__ __attribute__((aligned(16))) unsigned char xl[NP * XB];
  __shared__ __attribute__((aligned(16))) unsigned char dl[NP * DB];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int strip = blockIdx.x, vg = blockIdx.y, n = blockIdx.z;
  const int y0 = strip * RH;
  const long HW = (long)H * W_;
  const T* __restrict__ xg = x + ((long)n * C + (long)vg * SC) * HW;
  const T* __restrict__ dg = dy + ((long)n * C + (long)vg * SC) * HW;

  for (int i = t * 16; i < NP * XB; i += 256 * 16) *reinterpret_cast<u32x4*>(xl + i) = u32x4{0u, 0u, 0u, 0u};
  for (int i = t * 16; i < NP * DB; i += 256 * 16) *reinterpret_cast<u32x4*>(dl + i) = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();

  G3Piece<T> rgx[G::NLD], rgd[G::NLD];
  auto load_rows = [&](G3Piece<T> (&rg)[G::NLD], const T* __restrict__ src, int first, int nrows) {
#pragma unroll
    for (int j = 0; j < G::NLD; ++j) {
      const int id = t + 256 * j;
      const int ri = id / (SC * G::XC), rem = id - ri * (SC * G::XC);
      const int ch = rem / G::XC, xc = rem - ch * G::XC;
      const int row = first + ri;
      const bool ok = id < nrows * SC * G::XC && row >= 0 && row < H;
      g3_load(rg[j], src + (ok ? (long)ch * HW + (long)row * W_ + xc * 8 : 0), ok);
    }
  };
  // slot0: ring index of row `first` (x: (row - y0 + 1) % RX, dy: (row - y0) % RD); img: plane 0 of the image (past its guard
  // chunk), pstride: bytes between planes
  auto store_rows = [&](const G3Piece<T> (&rg)[G::NLD], unsigned char* img, int pstride, int slot0, int nslots, int nrows) {
#pragma unroll
    for (int j = 0; j < G::NLD; ++j) {
      const int id = t + 256 * j;
      const int ri = id / (SC * G::XC), rem = id - ri * (SC * G::XC);
      const int ch = rem / G::XC, xc = rem - ch * G::XC;
      const int slot = (slot0 + ri) % nslots;
      if (id < nrows * SC * G::XC) {
        u32x4 pl[NP];
        g3_planes(rg[j], pl);
        unsigned char* dst = img + (slot * CPRW + (xc >> 2)) * CHB + ch * 64 + ((((unsigned)xc & 3u) ^ g3w_sw(ch)) << 4);
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x4*>(dst + q * pstride) = pl[q];
      }
    }
  };
  unsigned char* dimg = dl + CHB;

  // ---- fragment addresses inside a chunk
  const int g4 = lane >> 4, c16 = lane & 15;
  const int cha = (CPG == 32 ? 16 * (wave >> 1) : 0) + c16;   // dy row (oc) of this lane
  const int chb = (CPG == 32 ? 16 * (wave & 1) : 0) + c16;    // x row (ci)
  const unsigned swa = g3w_sw(cha), swb = g3w_sw(chb);
  const int a_cur = cha * 64 + (int)(((unsigned)g4 ^ swa) << 4);
  const int a_prev = (g4 > 0 ? cha * 64 + (int)(((unsigned)(g4 - 1) ^ swa) << 4) : -CHB + cha * 64 + (int)((3u ^ swa) << 4)) + 12;
  const int a_next = g4 < 3 ? cha * 64 + (int)(((unsigned)(g4 + 1) ^ swa) << 4) : CHB + cha * 64 + (int)((0u ^ swa) << 4);
  const int b_cur = chb * 64 + (int)(((unsigned)g4 ^ swb) << 4);

  f32x4 acc[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: x rows y0 - 1 .. y0 + RI, dy rows y0 .. y0 + RI - 1
  for (int r = y0 - 1; r <= y0 + RI; r += RI) {
    const int nr = min(RI, y0 + RI - r + 1);
    load_rows(rgx, xg, r, nr);
    store_rows(rgx, xl, XB, r - y0 + 1, RX, nr);
  }
  load_rows(rgd, dg, y0, RI);
  store_rows(rgd, dimg, DB, 0, G::RD, RI);
  __syncthreads();

  const int iters = RH / RI;
  for (int it = 0; it < iters; ++it) {
    const int r0 = y0 + it * RI;
    const bool more = it + 1 < iters;
    if (more) {
      load_rows(rgx, xg, r0 + RI + 1, RI);
      load_rows(rgd, dg, r0 + RI, RI);
    }
    for (int c = (CPG == 32 ? 0 : wave); c < G::NCH; c += (CPG == 32 ? 1 : 4)) {
      const int rowi = c / CPRW, cr = c - rowi * CPRW;
      bf16x8 af[NP][3];
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const unsigned char* dc = dimg + q * DB + c * CHB;
        const u32x4 d = *reinterpret_cast<const u32x4*>(dc + a_cur);
        const unsigned dm = *reinterpret_cast<const unsigned*>(dc + a_prev);
        const unsigned dn = *reinterpret_cast<const unsigned*>(dc + a_next);
        af[q][1] = __builtin_bit_cast(bf16x8, d);
        // kx = 0 multiplies dy[q + 1], kx = 2 dy[q - 1]
        af[q][0] = __builtin_bit_cast(bf16x8, u32x4{__builtin_amdgcn_alignbit(d[1], d[0], 16), __builtin_amdgcn_alignbit(d[2], d[1], 16),
                                                    __builtin_amdgcn_alignbit(d[3], d[2], 16), __builtin_amdgcn_alignbit(dn, d[3], 16)});
        af[q][2] = __builtin_bit_cast(bf16x8, u32x4{__builtin_amdgcn_alignbit(d[0], dm, 16), __builtin_amdgcn_alignbit(d[1], d[0], 16),
                                                    __builtin_amdgcn_alignbit(d[2], d[1], 16), __builtin_amdgcn_alignbit(d[3], d[2], 16)});
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int sx = (it * RI + rowi + ky) % RX;
        bf16x8 bfr[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) bfr[q] = *reinterpret_cast<const bf16x8*>(xl + q * XB + (sx * CPRW + cr) * CHB + b_cur);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          f32x4 cc = acc[ky][kx];
          if constexpr (NP == 3) {   // the six plane products of the exact split, smallest terms first (splitgemm.hip)
            cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2][kx], bfr[0], cc, 0, 0, 0);
            cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][kx], bfr[2], cc, 0, 0, 0);
            cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][kx], bfr[1], cc, 0, 0, 0);
            cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][kx], bfr[0], cc, 0, 0, 0);
            cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][kx], bfr[1], cc, 0, 0, 0);
          }
          acc[ky][kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][kx], bfr[0], cc, 0, 0, 0);
        }
      }
    }
    __syncthreads();   // every wave is done with the rows that the incoming ones replace
    if (more) {
      store_rows(rgx, xl, XB, (r0 + RI + 1 - y0 + 1) % RX, RX, RI);
      store_rows(rgd, dimg, DB, 0, G::RD, RI);
    }
    __syncthreads();
  }

  // ---- 16-channel slabs: add the four waves' tiles (3 + 2 -> 1 + 0, then 1 -> 0), fixed order
  if constexpr (CPG != 32) {
    float* red = reinterpret_cast<float*>(xl);
    auto put = [&](int slotw) {
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) *reinterpret_cast<f32x4*>(red + ((slotw * 9 + a * 3 + b) * 64 + lane) * 4) = acc[a][b];
    };
    auto add = [&](int slotw) {
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[a][b] += *reinterpret_cast<const f32x4*>(red + ((slotw * 9 + a * 3 + b) * 64 + lane) * 4);
    };
    if (wave >= 2) put(wave - 2);
    __syncthreads();
    if (wave < 2) add(wave);
    __syncthreads();
    if (wave == 1) put(0);
    __syncthreads();
    if (wave == 0) add(0);
  }
  // ---- D[oc = 4 g4 + q][ci = c16] of tap (ky, kx): only the products inside a group are weights
  if (CPG == 32 || wave == 0) {
    float* __restrict__ dst = part + ((long)n * gridDim.x + strip) * out_elems;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int oc = (CPG == 32 ? 16 * (wave >> 1) : 0) + 4 * g4 + q, ci = chb;
      if (oc / CPG == ci / CPG) {
        const long co = (long)vg * SC + oc;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) dst[(co * CPG + (ci % CPG)) * 9 + ky * 3 + kx] = acc[ky][kx][q];
      }
    }
  }
}

template <int CPG, int W_, int NP, typename T>
int g3bw_launch(const T* dy, const T* x, float* dw, int NB, int C, int H, int RH, int accumulate, void* ws,
                size_t ws_bytes, hipStream_t st) {
  using G = G3W<CPG, W_, NP>;
  const int slabs = C / G::SC, strips = H / RH;
  const long out_elems = (long)C * CPG * 9, parts = (long)NB * strips;
  WFAE_REQUIRE(ws && ws_bytes >= (size_t)parts * out_elems * sizeof(float), WFAE_ERR_WORKSPACE,
               "g3b_bwd_weight: workspace %zu < %zu", ws_bytes, (size_t)parts * out_elems * sizeof(float));
  hipLaunchKernelGGL((g3bw_kernel<CPG, W_, NP, T>), dim3(strips, slabs, NB), dim3(256), 0, st, dy, x, (float*)ws, C, H, RH,
                     out_elems);
  int rc = check_launch("g3bw");
  if (rc) return rc;
  return slab_reduce((const float*)ws, dw, nullptr, out_elems, 1, (int)parts, accumulate, st);
}

}  // namespace

extern "C" {

int wfae_g3b_supported(int C, int H, int W, int groups) {
  if (C <= 0 || groups <= 0 || C % groups != 0 || H <= 0) return 0;
  const int cpg = C / groups;
  if (C % (cpg == 32 ? 32 : 16) != 0 || C / (cpg == 32 ? 32 : 16) > 65535) return 0;
  return g3b_strip(cpg, H, W) > 0 ? 1 : 0;
}

int wfae_g3b_fwd_bf16(const uint16_t* x, const float* w, uint16_t* y, int NB, int C, int H, int W, int groups, int transposed,
                      void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(x && w && y, WFAE_ERR_NULL_POINTER, "g3b_fwd: null pointer");
  WFAE_REQUIRE(NB > 0 && NB <= 65535 && C > 0 && H > 0 && W > 0 && groups > 0 && C % groups == 0, WFAE_ERR_BAD_SHAPE,
               "g3b_fwd: bad shape");
  WFAE_REQUIRE(wfae_g3b_supported(C, H, W, groups), WFAE_ERR_UNSUPPORTED,
               "g3b_fwd: (channels per group, width) must be (4,384), (8,192), (16,96), (32,48) or (32,24) with H a multiple of "
               "the strip height (ask wfae_g3b_supported; wfae_gconv3x3_fwd_bf16 serves every shape)");
  WFAE_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(ws)) & 15) == 0,
               WFAE_ERR_BAD_SHAPE, "g3b_fwd: tensors and workspace must be 16-byte aligned");
  WFAE_REQUIRE(wfae::matmul_precision() == WFAE_PRECISION_BF16, WFAE_ERR_UNSUPPORTED,
               "g3b_fwd: bf16 activation storage needs wfae_set_matmul_precision(WFAE_PRECISION_BF16)");
  const int cpg = C / groups, rh = g3b_strip(cpg, H, W);
  hipStream_t st = (hipStream_t)stream;
  if (cpg == 4) return g3b_launch<4, 24, 384, 1, 6, bf16_t>(x, w, y, NB, C, H, rh, transposed, ws, ws_bytes, st);
  if (cpg == 8) return g3b_launch<8, 12, 192, 1, 6, bf16_t>(x, w, y, NB, C, H, rh, transposed, ws, ws_bytes, st);
  if (cpg == 16) return g3b_launch<16, 6, 96, 1, 6, bf16_t>(x, w, y, NB, C, H, rh, transposed, ws, ws_bytes, st);
  if (W == 48) return g3b_launch<32, 3, 48, 1, 6, bf16_t>(x, w, y, NB, C, H, rh, transposed, ws, ws_bytes, st);
  return g3b_launch<32, 2, 24, 1, 6, bf16_t>(x, w, y, NB, C, H, rh, transposed, ws, ws_bytes, st);
}

int wfae_g3b_bwd_weight_bf16(const uint16_t* dy, const uint16_t* x, float* dw, int NB, int C, int H, int W, int groups,
                             int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(dy && x && dw, WFAE_ERR_NULL_POINTER, "g3b_bwd_weight: null pointer");
  WFAE_REQUIRE(NB > 0 && NB <= 65535 && C > 0 && H > 0 && W > 0 && groups > 0 && C % groups == 0, WFAE_ERR_BAD_SHAPE,
               "g3b_bwd_weight: bad shape");
  WFAE_REQUIRE(wfae_g3b_supported(C, H, W, groups), WFAE_ERR_UNSUPPORTED,
               "g3b_bwd_weight: shape not served (ask wfae_g3b_supported; wfae_gconv3x3_bwd_weight_bf16 serves every shape)");
  WFAE_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0, WFAE_ERR_BAD_SHAPE,
               "g3b_bwd_weight: tensors must be 16-byte aligned");
  WFAE_REQUIRE(wfae::matmul_precision() == WFAE_PRECISION_BF16, WFAE_ERR_UNSUPPORTED,
               "g3b_bwd_weight: bf16 activation storage needs wfae_set_matmul_precision(WFAE_PRECISION_BF16)");
  const int cpg = C / groups, rh = g3b_strip(cpg, H, W);
  hipStream_t st = (hipStream_t)stream;
  if (cpg == 4) return g3bw_launch<4, 384, 1, bf16_t>(dy, x, dw, NB, C, H, rh, accumulate, ws, ws_bytes, st);
  if (cpg == 8) return g3bw_launch<8, 192, 1, bf16_t>(dy, x, dw, NB, C, H, rh, accumulate, ws, ws_bytes, st);
  if (cpg == 16) return g3bw_launch<16, 96, 1, bf16_t>(dy, x, dw, NB, C, H, rh, accumulate, ws, ws_bytes, st);
  if (W == 48) return g3bw_launch<32, 48, 1, bf16_t>(dy, x, dw, NB, C, H, rh, accumulate, ws, ws_bytes, st);
  return g3bw_launch<32, 24, 1, bf16_t>(dy, x, dw, NB, C, H, rh, accumulate, ws, ws_bytes, st);
}

/* fp32 tensors: the same kernel with the exact three-plane split done once per value at the LDS store (six bf16 products
 * per fp32 product, fp32 accuracy — csrc/splitgemm.hip).  Served for W <= 96 (16 / 32 channels per group in the model): three
 * planes of full-width rows do not fit the LDS at W = 384, and at W = 192 they leave room for one block per CU, which measured
 * slower than the fp32-MFMA kernel of dconv.hip (0.335 vs 0.305 ms at 64 channels; W <= 96: 0.155 -> 0.122, 0.179 -> 0.096,
 * 0.077 -> 0.049 ms) */
int wfae_g3b_f32_supported(int C, int H, int W, int groups, int wgrad) {
  if (!wfae_g3b_supported(C, H, W, groups)) return 0;
  if (wfae::matmul_precision() != WFAE_PRECISION_FP32 || !wfae::split_gemm_enabled()) return 0;
  if (wgrad) return W <= 96 ? 1 : 0;
  // forward / data gradient: 16 channels per group @96 (three tiles per wave, two blocks per CU: 0.155 - 0.172 -> 0.119 - 0.129
  // ms).  8 @192 measured SLOWER than the VALU kernel of dconv.hip (0.228 vs 0.218 ms: the split of every staged value,
  // three times the transposed reads and 36 MFMAs per tile at two waves per SIMD) and was removed in round 4;
  // 4 @384 fits one block per CU only; at 32 per group the three-plane weight fragments of a wave exceed the register file
  const int cpg = C / groups;
  return cpg == 16 && W == 96 ? 1 : 0;
}

int wfae_g3b_fwd(const float* x, const float* w, float* y, int NB, int C, int H, int W, int groups, int transposed, void* ws,
                 size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(x && w && y, WFAE_ERR_NULL_POINTER, "g3b_fwd: null pointer");
  WFAE_REQUIRE(NB > 0 && NB <= 65535 && C > 0 && H > 0 && W > 0 && groups > 0 && C % groups == 0, WFAE_ERR_BAD_SHAPE,
               "g3b_fwd: bad shape");
  WFAE_REQUIRE(wfae_g3b_f32_supported(C, H, W, groups, 0), WFAE_ERR_UNSUPPORTED,
               "g3b_fwd: (channels per group, W) must be (16, 96) at fp32 precision with the split switch on (wfae_gconv3x3_fwd "
               "serves every shape)");
  WFAE_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(ws)) & 15) == 0,
               WFAE_ERR_BAD_SHAPE, "g3b_fwd: tensors and workspace must be 16-byte aligned");
  const int cpg = C / groups, rh = g3b_strip(cpg, H, W);
  hipStream_t st = (hipStream_t)stream;
  return g3b_launch<16, 6, 96, 3, 3, float>(x, w, y, NB, C, H, rh, transposed, ws, ws_bytes, st);
}

int wfae_g3b_bwd_weight(const float* dy, const float* x, float* dw, int NB, int C, int H, int W, int groups, int accumulate,
                        void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(dy && x && dw, WFAE_ERR_NULL_POINTER, "g3b_bwd_weight: null pointer");
  WFAE_REQUIRE(NB > 0 && NB <= 65535 && C > 0 && H > 0 && W > 0 && groups > 0 && C % groups == 0, WFAE_ERR_BAD_SHAPE,
               "g3b_bwd_weight: bad shape");
  WFAE_REQUIRE(wfae_g3b_f32_supported(C, H, W, groups, 1), WFAE_ERR_UNSUPPORTED,
               "g3b_bwd_weight: shape / mode not served (ask wfae_g3b_f32_supported; wfae_gconv3x3_bwd_weight serves every shape)");
  WFAE_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0, WFAE_ERR_BAD_SHAPE,
               "g3b_bwd_weight: tensors must be 16-byte aligned");
  const int cpg = C / groups, rh = g3b_strip(cpg, H, W);
  hipStream_t st = (hipStream_t)stream;
  if (cpg == 16) return g3bw_launch<16, 96, 3, float>(dy, x, dw, NB, C, H, rh, accumulate, ws, ws_bytes, st);
  if (W == 48) return g3bw_launch<32, 48, 3, float>(dy, x, dw, NB, C, H, rh, accumulate, ws, ws_bytes, st);
  return g3bw_launch<32, 24, 3, float>(dy, x, dw, NB, C, H, rh, accumulate, ws, ws_bytes, st);
}

}  // extern "C"
