// dconv.hip — direct (im2col-free) VALU convolutions on NCHW tiles staged in LDS.
// Used where the channel count per group is too small for a 32x32 MFMA tile:
//   * grouped 3x3 pad-1 conv of Bottleneck (ae_64x8x8_lin.py:17): 4..32 ch/group
//   * 128->1 3x3 output conv (:84) and its data gradient (1->128)
//   * 1->256 4x4 stride-2 input conv (:31 with in_ch = 1)
// Each block stages an input patch (+halo) with coalesced row reads, weights
// are wave-uniform (scalar loads), every thread owns one output pixel and OCB
// output channels in registers.
#include <stdlib.h>
#include "common.h"
#include <type_traits>

using namespace wfae;

namespace {

struct DConvP {
  const void* x;   // element type XT of the kernel (float, or bf16_t in the bf16-storage mode)
  const float* w;
  const float* bias;
  void* y;         // element type YT
  int Cin, Cout, H, W, Ho, Wo, pad, groups;
  int tiles_x;
};

// TRANSPOSED: weights are read as the data-gradient operator of a stride-1
// conv: w'(o,i,tap) = w[(g*IG + i)][o][KK-1-tap] with the original tensor laid
// out [groups*IG][OG][KS][KS].
template <int KS, int S, int OCB, bool TRANSPOSED, typename XT = float, typename YT = float>
__global__ __launch_bounds__(256) void dconv_fwd_kernel(DConvP p) {
  constexpr int KK = KS * KS;
  // 8 x 32 output tiles: a tile row is one whole 128-byte line of y (measured a few % better than 16 x 16 on the
  // HBM-bound first-layer / output-convolution shapes)
  constexpr int TY = 8, TX = 32;
  constexpr int TIH = (TY - 1) * S + KS;
  constexpr int TIW = (TX - 1) * S + KS;
  constexpr int CIB = (S == 1) ? 8 : 2;
  __shared__ float xs[CIB][TIH][TIW + 1];

  const int t = threadIdx.x;
  const int tx = t & (TX - 1), ty = t / TX;
  const int tile = blockIdx.x;
  const int oy0 = (tile / p.tiles_x) * TY, ox0 = (tile % p.tiles_x) * TX;
  const int n = blockIdx.z;
  const int IG = p.Cin / p.groups, OG = p.Cout / p.groups;
  const int o0 = blockIdx.y * OCB;
  const int g = o0 / OG;
  const int ol0 = o0 - g * OG;

  float acc[OCB];
#pragma unroll
  for (int o = 0; o < OCB; ++o) acc[o] = 0.f;

  const XT* xg = reinterpret_cast<const XT*>(p.x) + ((long)n * p.Cin + (long)g * IG) * p.H * p.W;
  const int iy_base = oy0 * S - p.pad, ix_base = ox0 * S - p.pad;

  for (int ci0 = 0; ci0 < IG; ci0 += CIB) {
    const int cn = min(CIB, IG - ci0);
    __syncthreads();
    if (cn == CIB) {
      // full chunk: all loads of the patch are issued before the first LDS store (the rolled loop below waits for
      // every element: load -> store -> load ...), out-of-image elements read a clamped address and are zeroed
      constexpr int NSLOT = (CIB * TIH * TIW + 255) / 256;
      float rv[NSLOT];
#pragma unroll
      for (int j = 0; j < NSLOT; ++j) {
        const int idx = t + j * 256;
        const int c = idx / (TIH * TIW);
        const int r = idx - c * (TIH * TIW);
        const int ry = r / TIW, rx = r - ry * TIW;
        const int iy = iy_base + ry, ix = ix_base + rx;
        const bool ok = idx < CIB * TIH * TIW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const float v = ld1(xg + (ok ? ((long)(ci0 + c) * p.H + iy) * p.W + ix : 0));
        rv[j] = ok ? v : 0.f;
      }
#pragma unroll
      for (int j = 0; j < NSLOT; ++j) {
        const int idx = t + j * 256;
        const int c = idx / (TIH * TIW);
        const int r = idx - c * (TIH * TIW);
        const int ry = r / TIW, rx = r - ry * TIW;
        if (idx < CIB * TIH * TIW) xs[c][ry][rx] = rv[j];
      }
    } else {
      for (int idx = t; idx < cn * TIH * TIW; idx += 256) {
        const int c = idx / (TIH * TIW);
        const int r = idx - c * (TIH * TIW);
        const int ry = r / TIW, rx = r - ry * TIW;
        const int iy = iy_base + ry, ix = ix_base + rx;
        float v = 0.f;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) v = ld1(xg + ((long)(ci0 + c) * p.H + iy) * p.W + ix);
        xs[c][ry][rx] = v;
      }
    }
    __syncthreads();
    for (int c = 0; c < cn; ++c) {
      const int ci = ci0 + c;
#pragma unroll
      for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const float v = xs[c][ty * S + ky][tx * S + kx];
#pragma unroll
          for (int o = 0; o < OCB; ++o) {
            long wi;
            if constexpr (TRANSPOSED)
              wi = (((long)g * IG + ci) * OG + (ol0 + o)) * KK + (KK - 1 - (ky * KS + kx));
            else
              wi = ((long)(o0 + o) * IG + ci) * KK + ky * KS + kx;
            acc[o] = fmaf(v, p.w[wi], acc[o]);
          }
        }
    }
  }
  const int oy = oy0 + ty, ox = ox0 + tx;
  if (oy < p.Ho && ox < p.Wo) {
#pragma unroll
    for (int o = 0; o < OCB; ++o) {
      float v = acc[o];
      if (p.bias) v += p.bias[o0 + o];
      st1(reinterpret_cast<YT*>(p.y) + (((long)n * p.Cout + o0 + o) * p.Ho + oy) * p.Wo + ox, v);
    }
  }
}

// -------------------------------------------------------------------------------------
// Convolutions whose INPUT has one channel: the encoder's / PatchGAN's first layer Conv2d(1, C, 4, 2, 1) and the data
// gradient of the output convolution Conv2d(C, 1, 3, 1, 1) (a 3x3 stride-1 convolution of the one-channel dy with
// the flipped taps).  y[c][p] = sum_tap w[c][tap] * s[S p - pad + tap]: every output element is KS^2 FMAs on a
// neighbourhood that is the same for all C channels, so a thread keeps the neighbourhood of its 4 consecutive output
// pixels in registers, walks the channels with wave-uniform (scalar) weight loads and streams one float4 per channel
// to HBM — no LDS, whole 1 KB row segments per wave store.  HBM-bound on the write of y.
// -------------------------------------------------------------------------------------
template <int KS, int S, bool FLIP, typename YT = float>
__global__ __launch_bounds__(256) void c1in_conv_kernel(const float* __restrict__ s_, const float* __restrict__ w,
                                                        const float* __restrict__ bias, YT* __restrict__ y, int C,
                                                        int Hs, int Ws, int Ho, int Wo, int pad, int cpb) {
  constexpr int KK = KS * KS, NW = 3 * S + KS;   // neighbourhood: KS rows x NW columns
  const int WQ = Wo >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)Ho * WQ) return;
  const int oy = (int)(idx / WQ), ox = (int)(idx - (long)oy * WQ) * 4;
  const int n = blockIdx.z;
  const float* __restrict__ sp = s_ + (long)n * Hs * Ws;
  float nb[KS][NW];
#pragma unroll
  for (int r = 0; r < KS; ++r) {
    const int iy = S * oy - pad + r;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const int ix = S * ox - pad + q;
      const bool ok = iy >= 0 && iy < Hs && ix >= 0 && ix < Ws;
      const float v = sp[ok ? (long)iy * Ws + ix : 0];
      nb[r][q] = ok ? v : 0.f;
    }
  }
  const int c0 = blockIdx.y * cpb;
  const int c1 = min(C, c0 + cpb);
  YT* __restrict__ yp = y + ((long)n * C + c0) * Ho * Wo + (long)oy * Wo + ox;
  const long plane = (long)Ho * Wo;
#pragma unroll 2
  for (int c = c0; c < c1; ++c) {
    const float* __restrict__ wc = w + (long)c * KK;   // wave-uniform: scalar loads
    const float b = bias ? bias[c] : 0.f;
    float o0 = b, o1 = b, o2 = b, o3 = b;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const float wv = FLIP ? wc[KK - 1 - (ky * KS + kx)] : wc[ky * KS + kx];
        o0 = fmaf(nb[ky][kx], wv, o0);
        o1 = fmaf(nb[ky][S + kx], wv, o1);
        o2 = fmaf(nb[ky][2 * S + kx], wv, o2);
        o3 = fmaf(nb[ky][3 * S + kx], wv, o3);
      }
    if constexpr (sizeof(YT) == 4) {
      typedef float vf4s __attribute__((ext_vector_type(4)));
      const vf4s o = {o0, o1, o2, o3};
      __builtin_nontemporal_store(o, reinterpret_cast<vf4s*>(yp));
    } else {
      typedef unsigned vu2s __attribute__((ext_vector_type(2)));
      const vu2s o = {pack_bf16(o0, o1), pack_bf16(o2, o3)};
      __builtin_nontemporal_store(o, reinterpret_cast<vu2s*>(yp));
    }
    yp += plane;
  }
}

// s: (N,1,Hs,Ws); y: (N,C,Ho,Wo); w: C x KS x KS taps (flip: used mirrored).  WFAE_ERR_UNSUPPORTED when Wo % 4 != 0
// or y is not 16-byte aligned (the generic kernels then run).
template <int KS, int S, bool FLIP, typename YT = float>
int launch_c1in_conv(const float* s_, const float* w, const float* bias, YT* y, int NB, int C, int Hs, int Ws, int Ho,
                     int Wo, int pad, hipStream_t st) {
  if (Wo % 4 != 0 || (reinterpret_cast<uintptr_t>(y) & 15) != 0 || NB > 65535) return WFAE_ERR_UNSUPPORTED;
  const long threads = (long)Ho * (Wo / 4);
  const int bx = cdiv(threads, 256);
  // channel chunks so that the grid has a few thousand blocks; every chunk re-reads the (tiny) one-channel input
  int chunks = cdiv(4096, (long)bx * NB);
  if (chunks < 1) chunks = 1;
  if (chunks > C) chunks = C;
  const int cpb = cdiv(C, chunks);
  hipLaunchKernelGGL((c1in_conv_kernel<KS, S, FLIP, YT>), dim3(bx, cdiv(C, cpb), NB), dim3(256), 0, st, s_, w, bias, y, C, Hs,
                     Ws, Ho, Wo, pad, cpb);
  return check_launch("c1in_conv");
}

template <int KS, int S, bool TR, typename XT = float, typename YT = float>
int launch_dconv(const DConvP& p, int NB, hipStream_t st) {
  const int OG = p.Cout / p.groups;
  const int tiles = p.tiles_x * cdiv(p.Ho, 8);
  dim3 block(256);
  if (OG % 8 == 0) {
    hipLaunchKernelGGL((dconv_fwd_kernel<KS, S, 8, TR, XT, YT>), dim3(tiles, p.Cout / 8, NB), block, 0, st, p);
  } else if (OG % 4 == 0) {
    hipLaunchKernelGGL((dconv_fwd_kernel<KS, S, 4, TR, XT, YT>), dim3(tiles, p.Cout / 4, NB), block, 0, st, p);
  } else {
    hipLaunchKernelGGL((dconv_fwd_kernel<KS, S, 1, TR, XT, YT>), dim3(tiles, p.Cout, NB), block, 0, st, p);
  }
  return check_launch("dconv");
}

// ------------------------------------------------------------------ wgrad
struct DWgradP {
  const float* dy;
  const float* x;
  float* part;  // [parts][Cout*IG*KK]
  int NB, Cin, Cout, H, W, Ho, Wo, pad, groups;
  int GB, OGc, IGc, PS;  // groups per block, out/in channels per group per block, pixel slices
  int n_ochunk, n_ichunk, n_gset;
  int tiles_x, tiles_y, parts;
};

constexpr int WG_TH = 8, WG_TW = 32, WG_TP = WG_TH * WG_TW;

template <int KS, int S, int OB>
__global__ __launch_bounds__(256) void dconv_wgrad_kernel(DWgradP p) {
  constexpr int KK = KS * KS;
  constexpr int HH = (WG_TH - 1) * S + KS;
  constexpr int HWD = (WG_TW - 1) * S + KS;
  constexpr int HALO = HH * HWD + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dys = smem;                              // [GB*OGc][WG_TP]
  float* xsm = smem + (long)p.GB * p.OGc * WG_TP;  // [GB*IGc][HALO]

  const int t = threadIdx.x;
  const int IG = p.Cin / p.groups, OG = p.Cout / p.groups;
  // blockIdx.y -> (gset, ochunk, ichunk)
  int by = blockIdx.y;
  const int ichunk = by % p.n_ichunk;
  by /= p.n_ichunk;
  const int ochunk = by % p.n_ochunk;
  const int gset = by / p.n_ochunk;
  const int g0 = gset * p.GB;
  const int oc0 = ochunk * p.OGc, ic0 = ichunk * p.IGc;

  // thread -> (gb, oq, i, ps)
  const int noq = p.OGc / OB;
  const int ps = t % p.PS;
  int r = t / p.PS;
  const int il = r % p.IGc;
  r /= p.IGc;
  const int oq = r % noq;
  const int gb = r / noq;
  const bool active = gb < p.GB && (g0 + gb) < p.groups && (ic0 + il) < IG && (oc0 + oq * OB) < OG;

  float acc[OB][KK];
#pragma unroll
  for (int o = 0; o < OB; ++o)
#pragma unroll
    for (int k = 0; k < KK; ++k) acc[o][k] = 0.f;

  const int total_tiles = p.NB * p.tiles_x * p.tiles_y;
  for (int tile = blockIdx.x; tile < total_tiles; tile += p.parts) {
    const int n = tile / (p.tiles_x * p.tiles_y);
    const int tr = tile - n * (p.tiles_x * p.tiles_y);
    const int oy0 = (tr / p.tiles_x) * WG_TH, ox0 = (tr % p.tiles_x) * WG_TW;
    __syncthreads();
    // stage dy tile
    const int ndy = p.GB * p.OGc * WG_TP;
    for (int idx = t; idx < ndy; idx += 256) {
      const int ch = idx / WG_TP, pp = idx - ch * WG_TP;
      const int gbb = ch / p.OGc, ol = ch - gbb * p.OGc;
      const int oy = oy0 + pp / WG_TW, ox = ox0 + pp % WG_TW;
      float v = 0.f;
      if (g0 + gbb < p.groups && oc0 + ol < OG && oy < p.Ho && ox < p.Wo)
        v = p.dy[(((long)n * p.Cout + (long)(g0 + gbb) * OG + oc0 + ol) * p.Ho + oy) * p.Wo + ox];
      dys[idx] = v;
    }
    // stage x halo tile
    const int nx = p.GB * p.IGc * (HH * HWD);
    const int iyb = oy0 * S - p.pad, ixb = ox0 * S - p.pad;
    for (int idx = t; idx < nx; idx += 256) {
      const int ch = idx / (HH * HWD), rr = idx - ch * (HH * HWD);
      const int gbb = ch / p.IGc, ii = ch - gbb * p.IGc;
      const int iy = iyb + rr / HWD, ix = ixb + rr % HWD;
      float v = 0.f;
      if (g0 + gbb < p.groups && ic0 + ii < IG && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
        v = p.x[(((long)n * p.Cin + (long)(g0 + gbb) * IG + ic0 + ii) * p.H + iy) * p.W + ix];
      xsm[ch * HALO + rr] = v;
    }
    __syncthreads();
    if (active) {
      const float* dyr = dys + (long)(gb * p.OGc + oq * OB) * WG_TP;
      const float* xr = xsm + (long)(gb * p.IGc + il) * HALO;
      for (int pp = ps; pp < WG_TP; pp += p.PS) {
        const int py = pp / WG_TW, px = pp % WG_TW;
        float d[OB];
#pragma unroll
        for (int o = 0; o < OB; ++o) d[o] = dyr[o * WG_TP + pp];
        const float* xp = xr + (py * S) * HWD + px * S;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            const float xv = xp[ky * HWD + kx];
#pragma unroll
            for (int o = 0; o < OB; ++o) acc[o][ky * KS + kx] = fmaf(d[o], xv, acc[o][ky * KS + kx]);
          }
      }
    }
  }
  // reduce over the PS consecutive lanes that share (gb, oq, i); PS is a power of two <= 64
#pragma unroll
  for (int o = 0; o < OB; ++o)
#pragma unroll
    for (int k = 0; k < KK; ++k) {
      float v = acc[o][k];
      for (int off = 1; off < p.PS; off <<= 1) v += __shfl_xor(v, off, 64);
      acc[o][k] = v;
    }
  if (active && ps == 0) {
    float* dst = p.part + (long)blockIdx.x * p.Cout * IG * KK;
#pragma unroll
    for (int o = 0; o < OB; ++o) {
      const int oc = oc0 + oq * OB + o;
      if (oc < OG) {
        const long base = (((long)(g0 + gb) * OG + oc) * IG + ic0 + il) * KK;
#pragma unroll
        for (int k = 0; k < KK; ++k) dst[base + k] = acc[o][k];
      }
    }
  }
}

inline int pow2_floor(int v) {
  int r = 1;
  while (r * 2 <= v) r *= 2;
  return r;
}

}  // namespace

extern "C" {

int wfae_dconv_fwd(const float* x, const float* w, const float* bias, float* y, int NB, int Cin,
                   int Cout, int H, int W, int KS, int stride, int pad, int groups,
                   wfae_stream_t stream) {
  WFAE_REQUIRE(x && w && y, WFAE_ERR_NULL_POINTER, "dconv_fwd: null pointer");
  WFAE_REQUIRE(NB > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && groups > 0 && Cin % groups == 0 &&
                   Cout % groups == 0,
               WFAE_ERR_BAD_SHAPE, "dconv_fwd: bad shape");
  WFAE_REQUIRE(NB <= 65535, WFAE_ERR_BAD_SHAPE, "dconv_fwd: batch > 65535");
  if (Cin == 1 && groups == 1 && KS == 4 && stride == 2 && pad == 1 && !(H & 1) && !(W & 1)) {
    const int rc = launch_c1in_conv<4, 2, false, float>(x, w, bias, y, NB, Cout, H, W, H / 2, W / 2, 1, (hipStream_t)stream);
    if (rc != WFAE_ERR_UNSUPPORTED) return rc;
  }
  DConvP p = {};
  p.x = x; p.w = w; p.bias = bias; p.y = y;
  p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W; p.pad = pad; p.groups = groups;
  p.Ho = (H + 2 * pad - KS) / stride + 1;
  p.Wo = (W + 2 * pad - KS) / stride + 1;
  p.tiles_x = cdiv(p.Wo, 32);
  hipStream_t st = (hipStream_t)stream;
  if (KS == 3 && stride == 1) return launch_dconv<3, 1, false>(p, NB, st);
  if (KS == 4 && stride == 2) return launch_dconv<4, 2, false>(p, NB, st);
  if (KS == 4 && stride == 1) return launch_dconv<4, 1, false>(p, NB, st);  // PatchGAN layer 4 (losses/model.py:137)
  return fail(WFAE_ERR_UNSUPPORTED, "dconv_fwd: KS=%d stride=%d unsupported", KS, stride);
}

int wfae_dconv_bwd_data(const float* dy, const float* w, float* dx, int NB, int Cin, int Cout,
                        int H, int W, int KS, int pad, int groups, wfae_stream_t stream) {
  WFAE_REQUIRE(dy && w && dx, WFAE_ERR_NULL_POINTER, "dconv_bwd_data: null pointer");
  WFAE_REQUIRE(NB > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && groups > 0 && Cin % groups == 0 &&
                   Cout % groups == 0,
               WFAE_ERR_BAD_SHAPE, "dconv_bwd_data: bad shape");
  WFAE_REQUIRE((KS == 3 || KS == 4) && pad <= KS - 1, WFAE_ERR_UNSUPPORTED,
               "dconv_bwd_data: stride-1 3x3 / 4x4 convolutions");
  WFAE_REQUIRE(NB <= 65535, WFAE_ERR_BAD_SHAPE, "dconv_bwd_data: batch > 65535");
  if (Cout == 1 && groups == 1 && KS == 3 && pad == 1) {
    // one-channel dy: dx[c][p] = sum_tap w[0][c][tap] dy[p + 1 - tap] = a 3x3 convolution of dy with the flipped taps
    const int rc = launch_c1in_conv<3, 1, true, float>(dy, w, nullptr, dx, NB, Cin, H, W, H, W, 1, (hipStream_t)stream);
    if (rc != WFAE_ERR_UNSUPPORTED) return rc;
  }
  // data gradient = stride-1 convolution of dy (Cout channels, (H+2pad-KS+1) x (W+2pad-KS+1)) producing the
  // Cin x H x W input gradient with transposed + spatially flipped weights and pad' = KS-1-pad.
  DConvP p = {};
  p.x = dy; p.w = w; p.bias = nullptr; p.y = dx;
  p.Cin = Cout; p.Cout = Cin;
  p.H = H + 2 * pad - KS + 1; p.W = W + 2 * pad - KS + 1;
  p.Ho = H; p.Wo = W;
  p.pad = KS - 1 - pad; p.groups = groups;
  p.tiles_x = cdiv(p.Wo, 32);
  if (KS == 3) return launch_dconv<3, 1, true>(p, NB, (hipStream_t)stream);
  return launch_dconv<4, 1, true>(p, NB, (hipStream_t)stream);
}

int wfae_dconv_bwd_weight(const float* dy, const float* x, float* dw, int NB, int Cin, int Cout,
                          int H, int W, int KS, int stride, int pad, int groups, int accumulate,
                          void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(dy && x && dw, WFAE_ERR_NULL_POINTER, "dconv_bwd_weight: null pointer");
  WFAE_REQUIRE(NB > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && groups > 0 && Cin % groups == 0 &&
                   Cout % groups == 0,
               WFAE_ERR_BAD_SHAPE, "dconv_bwd_weight: bad shape");
  if (KS == 3 && stride == 1 && pad == 1 && groups == 1 && Cout == 1) {
    const int rc = c1_wgrad_mfma(1, x, dy, dw, NB, Cin, H, W, accumulate, ws, ws_bytes, (hipStream_t)stream);
    if (rc != WFAE_ERR_UNSUPPORTED) return rc;
  }
  if (KS == 3 && stride == 1 && pad == 1 && groups == 1 && Cout == 1 && Cin >= 16 && NB <= 65535) {
    const int rc = c1conv3_wgrad(dy, x, dw, NB, Cin, H, W, accumulate, ws, ws_bytes, (hipStream_t)stream);
    if (rc != WFAE_ERR_WORKSPACE) return rc;      // partials did not fit: generic kernel below
  }
  if (KS == 4 && stride == 2 && pad == 1 && Cin == 1 && groups == 1) {
    const int rc = c1_wgrad_mfma(0, dy, x, dw, NB, Cout, H, W, accumulate, ws, ws_bytes, (hipStream_t)stream);
    if (rc != WFAE_ERR_UNSUPPORTED) return rc;   // odd shapes: generic kernel below
  }
  const int IG = Cin / groups, OG = Cout / groups;
  const int KK = KS * KS;
  DWgradP p = {};
  p.dy = dy; p.x = x; p.NB = NB; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W;
  p.pad = pad; p.groups = groups;
  p.Ho = (H + 2 * pad - KS) / stride + 1;
  p.Wo = (W + 2 * pad - KS) / stride + 1;
  const int OB = (OG % 4 == 0) ? 4 : 1;
  p.IGc = IG < 32 ? IG : 32;
  p.OGc = OG < 64 ? OG : 64;
  if (OB == 1 && p.OGc > 16) p.OGc = 16;
  const int units = (p.OGc / OB) * p.IGc;
  int GB = groups;
  if (GB > 32 / p.IGc) GB = 32 / p.IGc;
  if (GB > 64 / p.OGc) GB = 64 / p.OGc;
  if (GB > 256 / units) GB = 256 / units;
  if (GB < 1) GB = 1;
  p.GB = GB;
  WFAE_REQUIRE(GB * units <= 256, WFAE_ERR_UNSUPPORTED, "dconv_bwd_weight: channel blocking");
  int PS = pow2_floor(256 / (GB * units));
  if (PS > 64) PS = 64;
  p.PS = PS;
  p.n_ochunk = cdiv(OG, p.OGc);
  p.n_ichunk = cdiv(IG, p.IGc);
  p.n_gset = cdiv(groups, GB);
  p.tiles_x = cdiv(p.Wo, WG_TW);
  p.tiles_y = cdiv(p.Ho, WG_TH);
  const int gy = p.n_gset * p.n_ochunk * p.n_ichunk;
  const long total_tiles = (long)NB * p.tiles_x * p.tiles_y;
  const size_t out_elems = (size_t)Cout * IG * KK;
  long parts = 1024 / gy;
  if (parts < 1) parts = 1;
  if (parts > total_tiles) parts = total_tiles;
  while (parts > 1 && (size_t)parts * out_elems * sizeof(float) > ws_bytes) --parts;
  WFAE_REQUIRE(ws && (size_t)parts * out_elems * sizeof(float) <= ws_bytes, WFAE_ERR_WORKSPACE,
               "dconv_bwd_weight: workspace %zu too small", ws_bytes);
  p.parts = (int)parts;
  p.part = (float*)ws;
  const int HH = (WG_TH - 1) * stride + KS, HWD = (WG_TW - 1) * stride + KS;
  const size_t lds = ((size_t)GB * p.OGc * WG_TP + (size_t)GB * p.IGc * (HH * HWD + 1)) * sizeof(float);
  WFAE_REQUIRE(lds <= 160 * 1024, WFAE_ERR_UNSUPPORTED, "dconv_bwd_weight: LDS %zu", lds);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)parts, gy, 1), block(256);
#define WFAE_WG(KS_, S_, OB_)                                                                     \
  do {                                                                                            \
    (void)hipFuncSetAttribute((const void*)dconv_wgrad_kernel<KS_, S_, OB_>,                            \
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
    hipLaunchKernelGGL((dconv_wgrad_kernel<KS_, S_, OB_>), grid, block, lds, st, p);              \
  } while (0)
  if (KS == 3 && stride == 1) {
    if (OB == 4) WFAE_WG(3, 1, 4); else WFAE_WG(3, 1, 1);
  } else if (KS == 4 && stride == 2) {
    if (OB == 4) WFAE_WG(4, 2, 4); else WFAE_WG(4, 2, 1);
  } else if (KS == 4 && stride == 1) {
    if (OB == 4) WFAE_WG(4, 1, 4); else WFAE_WG(4, 1, 1);
  } else {
    return fail(WFAE_ERR_UNSUPPORTED, "dconv_bwd_weight: KS=%d stride=%d unsupported", KS, stride);
  }
#undef WFAE_WG
  int rc = check_launch("dconv_wgrad");
  if (rc) return rc;
  return slab_reduce((const float*)ws, dw, nullptr, (long)out_elems, 1, (int)parts, accumulate, st);
}

// bf16 storage: the two full-resolution convolutions with ONE channel on one side keep that side fp32 —
//   dconv_fwd_bf16out:      Conv2d(1, C, 4, 2, 1) forward (the encoder's first layer, ae_64x8x8_lin.py:31): x fp32 -> y bf16
//   dconv_fwd_bf16in:       Conv2d(C, 1, 3, 1, 1) forward (the output convolution, :84): x bf16 -> y fp32
//   dconv_bwd_data_bf16out: its data gradient, dy fp32 (one channel) -> dx bf16
//   c1_wgrad_bf16:          their weight gradients: flip 0: big = dy bf16 (N,C,H/2,W/2), small = x fp32 (N,1,H,W);
//                           flip 1: big = x bf16 (N,C,H,W), small = dy fp32 (N,1,H,W)
int wfae_dconv_fwd_bf16out(const float* x, const float* w, const float* bias, uint16_t* y, int NB, int Cout, int H, int W,
                           wfae_stream_t stream) {
  WFAE_REQUIRE(x && w && y, WFAE_ERR_NULL_POINTER, "dconv_fwd_bf16out: null pointer");
  WFAE_REQUIRE(NB > 0 && Cout > 0 && H > 0 && W > 0 && !(H & 1) && !(W & 1), WFAE_ERR_BAD_SHAPE, "dconv_fwd_bf16out: bad shape");
  const int rc = launch_c1in_conv<4, 2, false, bf16_t>(x, w, bias, y, NB, Cout, H, W, H / 2, W / 2, 1, (hipStream_t)stream);
  if (rc == WFAE_ERR_UNSUPPORTED) return fail(rc, "dconv_fwd_bf16out: needs W %% 8 == 0 and a 16-byte aligned result");
  return rc;
}
int wfae_dconv_fwd_bf16in(const uint16_t* x, const float* w, const float* bias, float* y, int NB, int Cin, int Cout, int H,
                          int W, wfae_stream_t stream) {
  WFAE_REQUIRE(x && w && y, WFAE_ERR_NULL_POINTER, "dconv_fwd_bf16in: null pointer");
  WFAE_REQUIRE(NB > 0 && NB <= 65535 && Cin > 0 && Cout > 0 && H > 0 && W > 0, WFAE_ERR_BAD_SHAPE, "dconv_fwd_bf16in: bad shape");
  DConvP p = {};
  p.x = x; p.w = w; p.bias = bias; p.y = y;
  p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W; p.pad = 1; p.groups = 1;
  p.Ho = H; p.Wo = W;
  p.tiles_x = cdiv(p.Wo, 32);
  return launch_dconv<3, 1, false, bf16_t, float>(p, NB, (hipStream_t)stream);
}
int wfae_dconv_bwd_data_bf16out(const float* dy, const float* w, uint16_t* dx, int NB, int Cin, int H, int W,
                                wfae_stream_t stream) {
  WFAE_REQUIRE(dy && w && dx, WFAE_ERR_NULL_POINTER, "dconv_bwd_data_bf16out: null pointer");
  WFAE_REQUIRE(NB > 0 && Cin > 0 && H > 0 && W > 0, WFAE_ERR_BAD_SHAPE, "dconv_bwd_data_bf16out: bad shape");
  const int rc = launch_c1in_conv<3, 1, true, bf16_t>(dy, w, nullptr, dx, NB, Cin, H, W, H, W, 1, (hipStream_t)stream);
  if (rc == WFAE_ERR_UNSUPPORTED) return fail(rc, "dconv_bwd_data_bf16out: needs W %% 4 == 0 and a 16-byte aligned result");
  return rc;
}
int wfae_c1_wgrad_bf16(int flip, const uint16_t* big, const float* small, float* dw, int NB, int C, int H, int W, int accumulate,
                       void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(big && small && dw, WFAE_ERR_NULL_POINTER, "c1_wgrad_bf16: null pointer");
  WFAE_REQUIRE(NB > 0 && C > 0 && H > 0 && W > 0, WFAE_ERR_BAD_SHAPE, "c1_wgrad_bf16: bad shape");
  const int rc = c1_wgrad_mfma(flip, big, small, dw, NB, C, H, W, accumulate, ws, ws_bytes, (hipStream_t)stream);
  if (rc == WFAE_ERR_UNSUPPORTED)
    return fail(rc, "c1_wgrad_bf16: needs C %% 64 == 0, even sizes, W %% 4 == 0 of the big tensor, aligned tensors, workspace");
  return rc;
}

}  // extern "C"

// =====================================================================================
// Grouped 3x3 'same' convolution of Bottleneck (ae_64x8x8_lin.py:17), register-blocked:
// each thread owns PY x 2 output pixels and ALL CPG output channels of its group, the
// input patch (+halo) of 8 channels at a time is staged in LDS with coalesced row reads,
// weights are pre-packed [g][ci][tap][oc] so the CPG weights of one (ci,tap) are one wide
// scalar load.  The same kernel computes the data gradient from transposed/flipped packing.
// =====================================================================================
namespace {

// XCD-aware block placement for the (tile, group, image) grids below.  Workgroups are dealt round-robin over the 8 XCDs in
// launch order, each XCD with its own L2, so spatially adjacent tiles — which share their halo rows / columns and, for
// 16- or 32-pixel-wide tiles, whole 128-byte lines — landed under eight different L2s and every shared line came from HBM
// once per neighbour (PMC round 2: 2.1x the algorithmic bytes for the 4- and 8-channel groups).  The linear block index is
// remapped so that each XCD works through ONE contiguous run of (tile, group, image) triples.
__device__ __forceinline__ void xcd_block(int& bx, int& by, int& bz) {
  const unsigned nx = gridDim.x, ny = gridDim.y;
  const unsigned total = nx * ny * gridDim.z;
  unsigned L = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
  const unsigned full = total & ~7u;
  if (L < full) L = (L & 7u) * (full >> 3) + (L >> 3);
  bx = (int)(L % nx);
  const unsigned r = L / nx;
  by = (int)(r % ny);
  bz = (int)(r / ny);
}

template <int CPG, int PY, typename T = float>
__global__ __launch_bounds__(256) void gconv3_kernel(const T* __restrict__ x, const float* __restrict__ wp,
                                                     T* __restrict__ y, int C, int H, int W, int tiles_x) {
  constexpr int TH = 16 * PY, TW = 32;
  constexpr int CIB = CPG < 8 ? CPG : 8;
  constexpr int IH = TH + 2, IWU = TW + 2, IW = 36;
  __shared__ __attribute__((aligned(16))) float xs[CIB][IH][IW];
  const int t = threadIdx.x;
  const int tx = t & 15, ty = t >> 4;
  int bx, g, n;
  xcd_block(bx, g, n);
  const int oy0 = (bx / tiles_x) * TH, ox0 = (bx % tiles_x) * TW;
  const T* xg = x + ((long)n * C + (long)g * CPG) * H * W;

  float acc[PY][2][CPG];
#pragma unroll
  for (int a = 0; a < PY; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int o = 0; o < CPG; ++o) acc[a][b][o] = 0.f;

  // Staging: every load of the patch is issued before the first LDS store (branch-free: out-of-range elements read a
  // clamped address and are zeroed afterwards).  The loop form "load, store, next element" made each of the ~10 (20)
  // elements a thread stages a separate round trip to memory, and with 36 (72) rounds of resident blocks per launch those
  // round trips, not bandwidth or FMA rate, set the time of the 4 / 8 channels-per-group layers.  (A row-wise float4
  // variant of the staging measured no faster at 8 channels per group and 10 % slower at 4.)
  constexpr int NEL = CIB * IH * IWU;
  constexpr int NIT = (NEL + 255) / 256;
  for (int ci0 = 0; ci0 < CPG; ci0 += CIB) {
    float sv[NIT];
    int so[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int idx = t + i * 256;
      const int c = idx / (IH * IWU);
      const int r = idx - c * (IH * IWU);
      const int ry = r / IWU, rx = r - ry * IWU;
      const int iy = oy0 - 1 + ry, ix = ox0 - 1 + rx;
      const bool ok = idx < NEL && iy >= 0 && iy < H && ix >= 0 && ix < W;
      sv[i] = ld1(xg + (ok ? ((long)(ci0 + c) * H + iy) * W + ix : 0));
      so[i] = idx < NEL ? (ok ? (c * IH + ry) * IW + rx : -1 - ((c * IH + ry) * IW + rx)) : (1 << 30);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      float* xf = &xs[0][0][0];
      if (so[i] != (1 << 30)) xf[so[i] >= 0 ? so[i] : -1 - so[i]] = so[i] >= 0 ? sv[i] : 0.f;
    }
    __syncthreads();
#pragma unroll 1  // measured: 2 and 4 are slower at 4 / 8 channels per group (SGPR pressure of the weight loads)
    for (int c = 0; c < CIB; ++c) {
      float in[PY + 2][4];
#pragma unroll
      for (int r = 0; r < PY + 2; ++r) {
        const float2 a = *reinterpret_cast<const float2*>(&xs[c][ty * PY + r][2 * tx]);
        const float2 b = *reinterpret_cast<const float2*>(&xs[c][ty * PY + r][2 * tx + 2]);
        in[r][0] = a.x; in[r][1] = a.y; in[r][2] = b.x; in[r][3] = b.y;
      }
      const float* __restrict__ wc = wp + ((long)(g * CPG + ci0 + c) * 9) * CPG;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int o = 0; o < CPG; ++o) {
            const float w = wc[(ky * 3 + kx) * CPG + o];
#pragma unroll
            for (int a = 0; a < PY; ++a)
#pragma unroll
              for (int b = 0; b < 2; ++b) acc[a][b][o] = fmaf(in[a + ky][b + kx], w, acc[a][b][o]);
          }
    }
  }
  const int ox = ox0 + 2 * tx;
#pragma unroll
  for (int a = 0; a < PY; ++a) {
    const int oy = oy0 + ty * PY + a;
    if (oy < H && ox < W) {
      T* yp = y + (((long)n * C + (long)g * CPG) * H + oy) * W + ox;
#pragma unroll
      for (int o = 0; o < CPG; ++o) {
        if (ox + 1 < W && (W & 1) == 0) {
          if constexpr (sizeof(T) == 4)
            *reinterpret_cast<float2*>(yp + (long)o * H * W) = make_float2(acc[a][0][o], acc[a][1][o]);
          else
            *reinterpret_cast<unsigned*>(yp + (long)o * H * W) = pack_bf16(acc[a][0][o], acc[a][1][o]);
        } else {
          st1(yp + (long)o * H * W, acc[a][0][o]);
          if (ox + 1 < W) st1(yp + (long)o * H * W + 1, acc[a][1][o]);
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------
// The same convolution for 4 / 8 channels per group (the 128@384 and 256@192 stages), fp32 tensors, as a persistent
// kernel: a block walks tiles of 16 x 64 pixels, FOUR pixels of a row per lane; the patch of its next tile travels HBM -> registers
// (16-byte row pieces + the two halo columns) while the current one is multiplied out of LDS, and the results of a tile are stored
// one tile late (16-byte stores), so the wait at the top of the loop finds both long landed.  The products are packed FMAs over
// pairs of output channels (see the loop).  What round 4 measured on this kernel's first form — 16 x 32 tiles, two pixels per lane —
// (256@192 / 128@384, B = 32; profiles/r04_gconv3p_*.txt):
//  - the one-tile-per-block kernel above already moved exactly the algorithmic bytes (PMC 1.00x) — in 0.199 / 0.320 ms;
//  - with loads and stores removed the products alone take 0.145 / 0.176 ms: 75 / 62 TF against 137 TF that the same
//    instruction (v_pk_fma_f32 with a scalar-register pair) reaches in a bare loop at 4 waves per SIMD
//    (tools/probe/valu_fma.hip; plain v_fma_f32 tops out at 77 TF) — the nine taps of one input channel are 72 FMAs per
//    scalar-load + LDS-read wait;
//  - loads alone or stores alone add ~0.02 ms to that, both ~0.04 - 0.13: a wave is either issuing memory operations
//    (38 % of its cycles by s_memtime stamps, the queue ahead of it full) or multiplying (49 %), and 4 - 6 waves per SIMD
//    overlap the two only partly.  Result 0.18 / 0.305 ms.  Not the DRAM pattern: 128-byte row pieces stream at 6.2 TB/s
//    read / 5.7 TB/s copy in tools/probe/hbm_streams.hip.
//  - four pixels per lane (this form): half the weight fetches, LDS reads and memory instructions per FMA and byte:
//    0.167 - 0.177 / 0.263 ms; eight (PX = 8, 16 x 128 tiles) measured the same at 4 channels per group (0.260 ms) and is not
//    instantiated.
// -------------------------------------------------------------------------------------
typedef float g3_f32x2 __attribute__((ext_vector_type(2)));

template <int CPG, int PX = 4>
__global__ __launch_bounds__(256) void gconv3p_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                      float* __restrict__ y, int NB, int C, int H, int W, int tiles_x,
                                                      int ntile, int total) {
  static_assert(CPG == 4 || CPG == 8, "whole group staged at once");
  static_assert(PX == 4 || PX == 8, "whole 16-byte pieces per lane");
  constexpr int TH = 16, TW = 16 * PX;      // PX pixels of a row per lane
  constexpr int IH = TH + 2, IW = TW + 4;   // LDS row: [left halo][TW body columns][right halo][2 pad]
  constexpr int F4R = TW / 4, RPP = 256 / F4R;   // 16-byte pieces per body row, rows per pass of the block
  constexpr int NROW = CPG * IH;            // patch rows (channel, ry)
  constexpr int NB4 = (NROW + RPP - 1) / RPP;
  constexpr int NHL = (2 * NROW + 255) / 256;   // halo columns: 2 elements per row
  __shared__ __attribute__((aligned(16))) float xs[NROW * IW];
  const int t = threadIdx.x;
  const int tx = t & 15, ty = t >> 4;
  const int HW = H * W;

  // tile order: (group, image, tile), group slowest.  Tiles are dealt in chunks of gridDim.x / 8: chunk q goes to XCD q % 8
  // (workgroup b runs on XCD b % 8), tile j of the chunk to that XCD's j-th block — neighbouring tiles are in flight
  // together under one L2 (shared halo lines), and the whole chip is inside a window of gridDim.x tiles at any time:
  // one or two groups, whose weights stay in the scalar caches.
  const int pb = (int)(gridDim.x >> 3);
  const int first = (int)(blockIdx.x & 7u) * pb + (int)(blockIdx.x >> 3), step = 8 * pb;
  if (first >= total) return;

  const int q4 = (t % F4R) * 4, r0 = t / F4R;
  float4 bv[NB4];
  float hv[NHL];
  auto fetch = [&](int L) {
    const int bx = L % ntile;
    const int r = L / ntile;
    const int n = r % NB, g = r / NB;
    const int py = (bx / tiles_x) * TH - 1, ox0 = (bx % tiles_x) * TW;
    const float* __restrict__ xb = x + ((long)n * C + (long)g * CPG) * HW;   // uniform base + 32-bit lane offsets
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
      const int R = r0 + RPP * i;
      const int c = R / IH, ry = R - c * IH;
      const int iy = py + ry, ix = ox0 + q4;
      bv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (R < NROW && (unsigned)iy < (unsigned)H && ix < W)
        bv[i] = *reinterpret_cast<const float4*>(xb + (unsigned)(__umul24(c, HW) + __umul24(iy, W) + ix));
    }
#pragma unroll
    for (int i = 0; i < NHL; ++i) {
      const int j = t + 256 * i;
      const int R = j >> 1;
      const int c = R / IH, ry = R - c * IH;
      const int iy = py + ry, ix = (j & 1) ? ox0 + TW : ox0 - 1;
      hv[i] = 0.f;
      if (R < NROW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
        hv[i] = xb[(unsigned)(__umul24(c, HW) + __umul24(iy, W) + ix)];
    }
  };

  // results leave one tile late: the stores of tile i - 1 are issued before the products of tile i, so that the wait at
  // the top of the loop (which covers every outstanding memory operation of the wave) finds loads and stores long landed
  float out[PX][CPG];
  int out_L = -1;
  auto put = [&]() {
    const int bx = out_L % ntile;
    const int r = out_L / ntile;
    const int n = r % NB, g = r / NB;
    const int ox = (bx % tiles_x) * TW + PX * tx, oy = (bx / tiles_x) * TH + ty;
    if (oy < H) {
      float* yp = y + (((long)n * C + (long)g * CPG) * H + oy) * W + ox;
#pragma unroll
      for (int o = 0; o < CPG; ++o)
#pragma unroll
        for (int v = 0; v < PX; v += 4)   // W % 4 == 0: a 16-byte piece is inside the row or outside it
          if (ox + v < W) *reinterpret_cast<float4*>(yp + (long)o * HW + v) = make_float4(out[v][o], out[v + 1][o], out[v + 2][o], out[v + 3][o]);
    }
  };

  fetch(first);
  for (int L = first; L < total; L += step) {
    __syncthreads();   // every wave is done reading the previous tile's patch
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
      const int R = r0 + RPP * i;
      if (R < NROW) {
        float* d = &xs[R * IW + 1 + q4];
        d[0] = bv[i].x; d[1] = bv[i].y; d[2] = bv[i].z; d[3] = bv[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < NHL; ++i) {
      const int j = t + 256 * i;
      if (j < 2 * NROW) xs[(j >> 1) * IW + ((j & 1) ? TW + 1 : 0)] = hv[i];
    }
    __syncthreads();
    if (L + step < total) fetch(L + step);
    if (out_L >= 0) put();

    const int g = L / ntile / NB;
    // packed FMAs over PAIRS OF OUTPUT CHANNELS: the weight pair (oc, oc + 1) of one (ci, tap) is an aligned SGPR pair as
    // the scalar load delivers it and the input value is broadcast by the instruction's operand select — pairing the two
    // pixels instead (what the compiler picks on its own) needs a register copy for every odd-aligned operand: 544 scalar
    // and 60 vector moves per tile and wave beside 576 packed FMAs (PMC round 4)
    g3_f32x2 acc[PX][CPG / 2];
#pragma unroll
    for (int b = 0; b < PX; ++b)
#pragma unroll
      for (int o = 0; o < CPG / 2; ++o) acc[b][o] = (g3_f32x2){0.f, 0.f};
    const float* __restrict__ wg = wp + (long)g * (CPG * 9 * CPG);
#pragma unroll(CPG == 4 && PX == 4 ? 2 : 1)   // <= 72 weights (scalar registers) and <= 144 packed FMAs per trip
    for (int c = 0; c < CPG; ++c) {
      float in[3][PX + 2];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float* xr = &xs[(c * IH + ty + r) * IW + PX * tx];
#pragma unroll
        for (int v = 0; v < PX; v += 4) {
          const float4 a = *reinterpret_cast<const float4*>(xr + v);
          in[r][v] = a.x; in[r][v + 1] = a.y; in[r][v + 2] = a.z; in[r][v + 3] = a.w;
        }
        const float2 b = *reinterpret_cast<const float2*>(xr + PX);
        in[r][PX] = b.x; in[r][PX + 1] = b.y;
      }
      const g3_f32x2* __restrict__ wc = reinterpret_cast<const g3_f32x2*>(wg + c * (9 * CPG));
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int o = 0; o < CPG / 2; ++o) {
            const g3_f32x2 w = wc[(ky * 3 + kx) * (CPG / 2) + o];
#pragma unroll
            for (int b = 0; b < PX; ++b)
              acc[b][o] = __builtin_elementwise_fma((g3_f32x2){in[ky][b + kx], in[ky][b + kx]}, w, acc[b][o]);
          }
    }
#pragma unroll
    for (int b = 0; b < PX; ++b)
#pragma unroll
      for (int o = 0; o < CPG / 2; ++o) { out[b][2 * o] = acc[b][o].x; out[b][2 * o + 1] = acc[b][o].y; }
    out_L = L;
  }
  put();
}

// -------------------------------------------------------------------------------------
// MFMA form of the same convolution for 16 / 32 channels per group (the 512@96, 1024@48 and 1024@24 stages):
// per group it is a GEMM  Y[oc][pixel] = sum_k W[oc][k] X[k][pixel],  k = (ci, tap), with M = CPG rows — one
// v_mfma_f32_32x32x2_f32 (CPG = 32) or v_mfma_f32_16x16x4_f32 (CPG = 16) tile.  A block owns a 16 x 16 pixel
// tile of one (image, group); the input patch (+halo) and the packed weights of 8 input channels at a time are
// staged in LDS (double-buffered through registers).  The k lanes of an MFMA (lane / CPG) take input channels
// CSTEP apart with the SAME tap, so both operand fragments are one ds_read_b32 at a per-lane base plus a
// compile-time immediate: no address arithmetic in the fully unrolled k loop.  A wave computes 4 tile rows
// (2 / 4 N tiles); the VALU kernel above reaches ~44 TF on these shapes.
// -------------------------------------------------------------------------------------
typedef float g3_f32x16 __attribute__((ext_vector_type(16)));
typedef float g3_f32x4 __attribute__((ext_vector_type(4)));

template <int CPG, typename T = float>
__global__ __launch_bounds__(256) void gconv3_mfma_kernel(const T* __restrict__ x, const float* __restrict__ wp,
                                                          T* __restrict__ y, int C, int H, int W, int tiles_x) {
  static_assert(CPG == 32 || CPG == 16, "one MFMA tile of output channels");
  constexpr int MF = CPG;             // MFMA tile edge
  constexpr int KL = 64 / MF;         // k lane groups of the MFMA: 2 (32x32x2) / 4 (16x16x4)
  constexpr int CSTEP = 8 / KL;       // input channels one lane group walks inside an 8-channel chunk
  constexpr int RPT = MF / 16;        // tile rows per N tile
  constexpr int NTW = 4 / RPT;        // N tiles per wave (4 tile rows per wave)
  constexpr int ROWS = 20, PLANE = 18 * ROWS;
  constexpr int XS = 8 * PLANE, WS = 8 * 9 * CPG;
  constexpr int NSLOT = (8 * 18 * 18 + 255) / 256;  // 11 patch elements per thread and chunk
  constexpr int WSLOT = (WS / 4 + 255) / 256;       // float4 weight loads per thread and chunk
  constexpr int NCH = CPG / 8;
  using acc_t = typename std::conditional<MF == 32, g3_f32x16, g3_f32x4>::type;
  constexpr int NACC = MF == 32 ? 16 : 4;
  __shared__ __attribute__((aligned(16))) float xs[2][XS];
  __shared__ __attribute__((aligned(16))) float wsm[2][WS];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int bx, g, n;
  xcd_block(bx, g, n);
  const int oy0 = (bx / tiles_x) * 16, ox0 = (bx % tiles_x) * 16;
  const long HW = (long)H * W;
  const T* __restrict__ xg = x + ((long)n * C + (long)g * CPG) * HW;
  const float* __restrict__ wg = wp + (long)g * CPG * 9 * CPG;

  int goff[NSLOT], loff[NSLOT];
  unsigned ok = 0;
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) {
    const int idx = t + j * 256;
    const int c = idx / 324, r = idx - c * 324;
    const int ry = r / 18, rx = r - ry * 18;
    const int iy = oy0 - 1 + ry, ix = ox0 - 1 + rx;
    const bool v = idx < 8 * 324 && iy >= 0 && iy < H && ix >= 0 && ix < W;
    ok |= (unsigned)v << j;
    goff[j] = v ? (int)(c * HW) + iy * W + ix : 0;
    loff[j] = idx < 8 * 324 ? c * PLANE + ry * ROWS + rx : -1;
  }
  float rx_[NSLOT];
  float4 rw_[WSLOT];
  acc_t acc[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int r = 0; r < NACC; ++r) acc[i][r] = 0.f;

  const int kk = lane / MF, lm = lane % MF;
  const int rit = lm >> 4, xl = lm & 15;  // pixel of this lane inside an N tile
  const int lane_a = (CSTEP * kk) * 9 * CPG + lm;
  const int lane_b = (CSTEP * kk) * PLANE + (wave * 4 + rit) * ROWS + xl;

#define WFAE_G3_LOAD(CH)                                                                     \
  {                                                                                          \
    const T* __restrict__ xc = xg + (long)(CH) * 8 * HW;                                     \
    _Pragma("unroll") for (int j = 0; j < NSLOT; ++j) rx_[j] = ld1(xc + goff[j]);            \
    const float4* __restrict__ wc = reinterpret_cast<const float4*>(wg + (long)(CH) * WS);   \
    _Pragma("unroll") for (int j = 0; j < WSLOT; ++j) {                                      \
      const int idx = t + j * 256;                                                           \
      rw_[j] = wc[idx < WS / 4 ? idx : 0];                                                   \
    }                                                                                        \
  }
#define WFAE_G3_STORE(BUF)                                                                   \
  {                                                                                          \
    float* __restrict__ xd = xs[BUF];                                                        \
    _Pragma("unroll") for (int j = 0; j < NSLOT; ++j)                                        \
      if (loff[j] >= 0) xd[loff[j]] = ((ok >> j) & 1u) ? rx_[j] : 0.f;                       \
    float4* __restrict__ wd = reinterpret_cast<float4*>(wsm[BUF]);                           \
    _Pragma("unroll") for (int j = 0; j < WSLOT; ++j) {                                      \
      const int idx = t + j * 256;                                                           \
      if (idx < WS / 4) wd[idx] = rw_[j];                                                    \
    }                                                                                        \
  }
#define WFAE_G3_COMPUTE(BUF)                                                                 \
  {                                                                                          \
    const float* __restrict__ wb = wsm[BUF] + lane_a;                                        \
    const float* __restrict__ xb = xs[BUF] + lane_b;                                         \
    _Pragma("unroll") for (int ci = 0; ci < CSTEP; ++ci)                                     \
      _Pragma("unroll") for (int ky = 0; ky < 3; ++ky)                                       \
        _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {                                   \
          const float a = wb[ci * 9 * CPG + (ky * 3 + kx) * CPG];                            \
          _Pragma("unroll") for (int i = 0; i < NTW; ++i) {                                  \
            const float b = xb[ci * PLANE + (i * RPT + ky) * ROWS + kx];                     \
            if constexpr (MF == 32)                                                          \
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);          \
            else                                                                             \
              acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);          \
          }                                                                                  \
        }                                                                                    \
  }
  // chunk ch + 1 travels HBM -> registers while chunk ch is multiplied; the last chunk is peeled
  WFAE_G3_LOAD(0)
  WFAE_G3_STORE(0)
  __syncthreads();
#pragma unroll 1
  for (int ch = 0; ch + 1 < NCH; ++ch) {
    const int buf = ch & 1;
    WFAE_G3_LOAD(ch + 1)
    WFAE_G3_COMPUTE(buf)
    WFAE_G3_STORE(buf ^ 1)
    __syncthreads();
  }
  WFAE_G3_COMPUTE((NCH - 1) & 1)
#undef WFAE_G3_LOAD
#undef WFAE_G3_STORE
#undef WFAE_G3_COMPUTE

  // C/D layout: column = lane % MF (the pixel), row = output channel:
  //   32x32: (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5);   16x16: 4 (lane >> 4) + reg
  T* __restrict__ yg = y + ((long)n * C + (long)g * CPG) * HW + (long)oy0 * W + ox0 + xl;
  const bool xin = ox0 + xl < W;  // partial tiles at the right / bottom edge
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int row = wave * 4 + i * RPT + rit;
    if (!xin || oy0 + row >= H) continue;
#pragma unroll
    for (int r = 0; r < NACC; ++r) {
      const int m = MF == 32 ? (r & 3) + 8 * (r >> 2) + 4 * kk : 4 * kk + r;
      st1(yg + (long)m * HW + (long)row * W, acc[i][r]);
    }
  }
}

// wp[((g*CPG + i)*9 + tap)*CPG + o]:  forward  i = ci, o = oc : w[g*CPG+o][i][tap]
//                                     transposed (dgrad) i = oc_orig, o = ci_orig : w[g*CPG+i][o][8-tap]
__global__ void gconv3_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int C, int CPG,
                                   int transposed) {
  const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
  if (i0 >= C * CPG * 9) return;
  const int o = i0 % CPG;
  int r = i0 / CPG;
  const int tap = r % 9;
  r /= 9;
  const int i = r % CPG, g = r / CPG;
  wp[i0] = transposed ? w[((long)(g * CPG + i) * CPG + o) * 9 + (8 - tap)]
                      : w[((long)(g * CPG + o) * CPG + i) * 9 + tap];
}

template <int CPG, typename T>
int launch_gconv3_mfma(const T* x, const float* wp, T* y, int NB, int C, int H, int W, hipStream_t st) {
  const int tiles_x = cdiv(W, 16), tiles_y = cdiv(H, 16);
  hipLaunchKernelGGL((gconv3_mfma_kernel<CPG, T>), dim3(tiles_x * tiles_y, C / CPG, NB), dim3(256), 0, st, x, wp, y, C, H,
                     W, tiles_x);
  return check_launch("gconv3_mfma");
}

template <int CPG, int PY, typename T>
int launch_gconv3(const T* x, const float* wp, T* y, int NB, int C, int H, int W, hipStream_t st) {
  const int tiles_x = cdiv(W, 32), tiles_y = cdiv(H, 16 * PY);
  hipLaunchKernelGGL((gconv3_kernel<CPG, PY, T>), dim3(tiles_x * tiles_y, C / CPG, NB), dim3(256), 0, st, x, wp, y, C, H,
                     W, tiles_x);
  return check_launch("gconv3");
}

inline int g3_num_cus() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();
  return n;
}

template <int CPG, int PX = 4>
int launch_gconv3p(const float* x, const float* wp, float* y, int NB, int C, int H, int W, hipStream_t st) {
  const int tiles_x = cdiv(W, 16 * PX), tiles_y = cdiv(H, 16);
  const long total = (long)tiles_x * tiles_y * (C / CPG) * NB;
  // a multiple of 8 blocks: 2 per CU at 4 channels per group (0.263 ms; 0.274 at 3, 0.282 at 4 and 6), 6 at 8 (0.167 - 0.177 ms;
  // 0.171 - 0.185 at 3: three fit by registers, the rest queue behind them)
  const int blocks = (int)std::min<long>((total + 7) / 8 * 8, (long)g3_num_cus() / 8 * 8 * (CPG == 4 ? 2 : 6));
  hipLaunchKernelGGL((gconv3p_kernel<CPG, PX>), dim3(blocks), dim3(256), 0, st, x, wp, y, NB, C, H, W, tiles_x, tiles_x * tiles_y,
                     (int)total);
  return check_launch("gconv3p");
}

template <typename T>
int gconv3x3_fwd_impl(const T* x, const float* w, T* y, int NB, int C, int H, int W, int groups, int transposed, void* ws,
                      size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(x && w && y, WFAE_ERR_NULL_POINTER, "gconv3x3_fwd: null pointer");
  WFAE_REQUIRE(NB > 0 && NB <= 65535 && C > 0 && H > 0 && W > 0 && groups > 0 && C % groups == 0, WFAE_ERR_BAD_SHAPE,
               "gconv3x3_fwd: bad shape");
  const int cpg = C / groups;
  WFAE_REQUIRE(cpg == 4 || cpg == 8 || cpg == 16 || cpg == 32, WFAE_ERR_UNSUPPORTED,
               "gconv3x3_fwd: %d channels per group (4/8/16/32 built; use wfae_dconv_fwd otherwise)", cpg);
  const size_t need = (size_t)C * cpg * 9 * sizeof(float);
  WFAE_REQUIRE(ws && ws_bytes >= need, WFAE_ERR_WORKSPACE, "gconv3x3_fwd: workspace %zu < %zu", ws_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gconv3_pack_kernel, dim3(cdiv((long)C * cpg * 9, 256)), dim3(256), 0, st, w, (float*)ws, C, cpg,
                     transposed);
  int rc = check_launch("gconv3_pack");
  if (rc) return rc;
  const float* wp = (const float*)ws;
  // 16 / 32 channels per group: the MFMA form on 16 x 16 pixel tiles (partial at the edges)
  if (cpg >= 16 && (long)cpg * H * W < (1l << 28)) {
    if (cpg == 32) return launch_gconv3_mfma<32>(x, wp, y, NB, C, H, W, st);
    return launch_gconv3_mfma<16>(x, wp, y, NB, C, H, W, st);
  }
  // one row per thread: measured 1 < 2 < 4 in time (the kernel is latency-bound: smaller LDS tiles, more blocks per CU)
  // 4 / 8 channels per group, fp32 tensors with 16-byte rows: the software-pipelined kernel (24-bit lane offsets in a group)
  if constexpr (sizeof(T) == 4) {
    if (cpg <= 8 && W % 4 == 0 && (long)cdiv(W, 64) * cdiv(H, 16) * groups * NB < (1l << 31) && (long)cpg * H * W < (1l << 24)) {
      if (cpg == 4) return launch_gconv3p<4>(x, wp, y, NB, C, H, W, st);
      return launch_gconv3p<8>(x, wp, y, NB, C, H, W, st);
    }
  }
  switch (cpg) {
    case 4: return launch_gconv3<4, 1>(x, wp, y, NB, C, H, W, st);
    case 8: return launch_gconv3<8, 1>(x, wp, y, NB, C, H, W, st);
    case 16: return launch_gconv3<16, 1>(x, wp, y, NB, C, H, W, st);
    default: return launch_gconv3<32, 1>(x, wp, y, NB, C, H, W, st);
  }
}

}  // namespace

extern "C" int wfae_gconv3x3_fwd(const float* x, const float* w, float* y, int NB, int C, int H, int W, int groups,
                                 int transposed, void* ws, size_t ws_bytes, wfae_stream_t stream) {
  return gconv3x3_fwd_impl(x, w, y, NB, C, H, W, groups, transposed, ws, ws_bytes, stream);
}
extern "C" int wfae_gconv3x3_fwd_bf16(const uint16_t* x, const float* w, uint16_t* y, int NB, int C, int H, int W, int groups,
                                      int transposed, void* ws, size_t ws_bytes, wfae_stream_t stream) {
  return gconv3x3_fwd_impl(x, w, y, NB, C, H, W, groups, transposed, ws, ws_bytes, stream);
}

// =====================================================================================
// Grouped 3x3 weight gradient for 4 / 8 / 16 channels per group: pixel-parallel.
// Every lane owns one pixel column of the tile and accumulates the full OCW x CPG x 9 = 144
// partial products of its wave's (group, output-channel chunk) in registers while the wave walks
// the rows of the tile (x patch + halo in LDS, dy read coalesced from HBM); the 64 lanes are
// combined once per block with wavefront shuffles and written as one partial slab row.
// =====================================================================================
namespace {

template <int CPG, int OCW>
__global__ __launch_bounds__(256, (OCW * CPG * 9 <= 72 ? 4 : 2)) void gconv3_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              float* __restrict__ part, int NB, int C, int H, int W,
                                                              int tiles_x, int tiles_y, int parts) {
  static_assert(OCW * CPG * 9 <= 144, "at most 144 accumulators per lane");
  constexpr int TH = 8, TW = 64, IH = TH + 2, IWU = TW + 2, IW = 68;
  constexpr int Q = CPG / OCW;                 // work units per group
  constexpr int GPB = Q >= 4 ? 1 : 4 / Q;      // groups per block (4 waves = 4 units)
  constexpr int XCH = GPB * CPG;               // input channels staged per block
  __shared__ float xs[XCH][IH][IW];
  // 72-accumulator variants park the dy values of a tile in LDS up front; with 144 accumulators the extra live
  // registers cost more than the per-row load stall (measured: 4 ch/group 0.95 -> 0.75 ms, 8 ch/group 0.63 -> 0.77 ms)
  constexpr bool HOIST = OCW * CPG * 9 <= 72;
  __shared__ float dys[HOIST ? 4 : 1][TH][OCW][64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int u = blockIdx.y * 4 + wave;         // (group, oc chunk)
  const int g = u / Q, q = u - g * Q;
  const int g0 = (blockIdx.y * 4) / Q;         // first group of this block
  const int cl = (g - g0) * CPG;               // channel offset of this wave's group inside xs
  const int oc0 = g * CPG + q * OCW;

  float acc[OCW][CPG][9];
#pragma unroll
  for (int o = 0; o < OCW; ++o)
#pragma unroll
    for (int c = 0; c < CPG; ++c)
#pragma unroll
      for (int k = 0; k < 9; ++k) acc[o][c][k] = 0.f;

  const int per_img = tiles_x * tiles_y;
  const int total = NB * per_img;
  for (int tile = blockIdx.x; tile < total; tile += parts) {
    const int n = tile / per_img;
    const int tr = tile - n * per_img;
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
    // HOIST: the dy values of all TH rows are requested first and parked in a per-lane LDS slot, so their HBM
    // latency hides behind the staging of the x patch instead of stalling every row (kept in registers across a
    // fully unrolled row loop they spill)
    const int ox = ox0 + lane;
    const float* dyp = dy + ((long)n * C + oc0) * H * W + ox;
    float dpre[HOIST ? TH : 1][OCW];
    if constexpr (HOIST) {
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        const bool ok = oy0 + r < H && ox < W;
#pragma unroll
        for (int o = 0; o < OCW; ++o) dpre[r][o] = ok ? dyp[((long)o * H + oy0 + r) * W] : 0.f;
      }
    }
    __syncthreads();  // every wave is done with the previous tile's xs / dys
    const float* xg = x + ((long)n * C + (long)g0 * CPG) * H * W;
    for (int idx = t; idx < XCH * IH * IWU; idx += 256) {
      const int c = idx / (IH * IWU);
      const int r = idx - c * (IH * IWU);
      const int ry = r / IWU, rx = r - ry * IWU;
      const int iy = oy0 - 1 + ry, ix = ox0 - 1 + rx;
      float v = 0.f;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = xg[((long)c * H + iy) * W + ix];
      xs[c][ry][rx] = v;
    }
    if constexpr (HOIST) {
#pragma unroll
      for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int o = 0; o < OCW; ++o) dys[wave][r][o][lane] = dpre[r][o];
    }
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < TH; ++r) {
      float d[OCW];
      if constexpr (HOIST) {
#pragma unroll
        for (int o = 0; o < OCW; ++o) d[o] = dys[wave][r][o][lane];
      } else {
        const bool ok = oy0 + r < H && ox < W;
#pragma unroll
        for (int o = 0; o < OCW; ++o) d[o] = ok ? dyp[((long)o * H + oy0 + r) * W] : 0.f;
      }
#pragma unroll
      for (int c = 0; c < CPG; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float xv = xs[cl + c][r + ky][lane + kx];
#pragma unroll
            for (int o = 0; o < OCW; ++o) acc[o][c][ky * 3 + kx] = fmaf(d[o], xv, acc[o][c][ky * 3 + kx]);
          }
    }
  }
  float* dst = part + (long)blockIdx.x * C * CPG * 9 + (long)oc0 * CPG * 9;
#pragma unroll
  for (int o = 0; o < OCW; ++o)
#pragma unroll
    for (int c = 0; c < CPG; ++c)
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float v = wave_sum(acc[o][c][k]);
        if (lane == 0) dst[(o * CPG + c) * 9 + k] = v;
      }
}

// -------------------------------------------------------------------------------------
// MFMA weight gradient of the grouped 3x3 convolution for 16 / 32 channels per group:
//   dW[oc][ci][ky][kx] = sum over pixels  dy[oc][p] * x[ci][p + (ky-1, kx-1)]
// is, per group and per tap, a (CPG x pixels) . (pixels x CPG) product — exactly one MFMA tile of output.  A block
// of 3 waves walks 8 x 16 pixel tiles of one group: dy is staged TRANSPOSED (dys[pixel][oc], row stride CPG + 1)
// and the x patch with an odd plane stride, so both operand fragments (lane = oc resp. ci, k lanes = adjacent
// pixels of a row) are conflict-free ds_read_b32 at compile-time offsets.  Wave w owns filter row ky = w and keeps
// three accumulator tiles (kx = 0..2) in registers across all tiles of its slice; partial sums go to a slab that
// slab_reduce adds in fixed order.  The generic B_WGRAD3 GEMM pads these shapes to 32 x 128 tiles (19 TF at
// 16 ch/group, 46 TF at 32).
// -------------------------------------------------------------------------------------
// RCPG < CPG: CPG / RCPG real groups of RCPG channels share one MFMA tile; the tile then also holds the products
// between different groups, which are simply not written (8 ch/group: half of the tile is used and the kernel
// still runs 2x faster than the VALU form; 4 ch/group: a quarter, +20 %)
template <int CPG, int RCPG = CPG, typename T = float>
__global__ __launch_bounds__(192) void gconv3_wgrad_mfma_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                float* __restrict__ part, int NB, int C, int H, int W,
                                                                int tiles_x, int tiles_y, int parts) {
  static_assert(CPG == 32 || CPG == 16, "one MFMA tile of (oc, ci)");
  constexpr int MF = CPG, KL = 64 / MF;          // k lanes of the MFMA = pixels per k step: 2 / 4
  constexpr int TH = 8, TW = 16, NPIX = TH * TW;
  constexpr int DLD = CPG + 1;                   // dys row stride (words): bank = (pixel + oc) % 32
  constexpr int ROWS = 20, PLANE = (TH + 2) * ROWS + 1;  // odd plane stride: bank = (9 ci + offset) % 32
  constexpr int NT = 192;
  using acc_t = typename std::conditional<MF == 32, g3_f32x16, g3_f32x4>::type;
  constexpr int NACC = MF == 32 ? 16 : 4;
  __shared__ float dys[NPIX * DLD];
  __shared__ float xs[CPG * PLANE];
  const int t = threadIdx.x, lane = t & 63, ky = t >> 6;
  const int g = blockIdx.y;
  const int lm = lane % MF, kq = lane / MF;
  const long HW = (long)H * W;

  acc_t acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < NACC; ++r) acc[i][r] = 0.f;

  // Staging maps with no per-element index arithmetic.  x patch: thread -> (ry, rx) = (t / 18, t % 18) of the
  // 10 x 18 patch (threads 180..191 idle), slot j = input channel j.  dy tile: thread -> (oc0, row, float4 q) =
  // (t / 32, (t / 4) % 8, t % 4), slot j = output channel oc0 + 6 j.  Bounds depend on the thread and the tile
  // only, not on the slot.  Tile i + 1 travels HBM -> registers while tile i is multiplied.
  constexpr int XSLOT = CPG, DSLOT = (CPG + 5) / 6;
  const int xry = t / 18, xrx = t - xry * 18;
  const bool xthr = t < 180;
  const int dq = t & 3, dr = (t >> 2) & 7, doc0 = t >> 5;
  float rxv[XSLOT];
  float4 rdv[DSLOT];
  const int per_img = tiles_x * tiles_y;
  const int total = NB * per_img;
  auto prefetch = [&](int tile) {
    const int n = tile / per_img;
    const int tr = tile - n * per_img;
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
    const long gbase = ((long)n * C + (long)g * CPG) * HW;
    const int iy = oy0 - 1 + xry, ix = ox0 - 1 + xrx;
    const bool xok = xthr && iy >= 0 && iy < H && ix >= 0 && ix < W;
    const T* __restrict__ xp = x + gbase + (xok ? (long)iy * W + ix : 0);
#pragma unroll
    for (int j = 0; j < XSLOT; ++j) {
      const float v = ld1(xp + (long)j * HW);
      rxv[j] = xok ? v : 0.f;
    }
    const int oy = oy0 + dr, ox = ox0 + dq * 4;
    const bool dok = oy < H && ox < W;   // W % 4 == 0: a float4 is inside or outside as a whole
    const T* __restrict__ dp = dy + gbase + (dok ? (long)oy * W + ox : 0);
#pragma unroll
    for (int j = 0; j < DSLOT; ++j) {
      const int oc = doc0 + 6 * j;
      const float4 v = ld4(dp + (long)(oc < CPG ? oc : 0) * HW);
      rdv[j] = dok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // (round 3: giving every block a CONTIGUOUS run of tiles instead of the stride `parts`, so that neighbouring tiles share one
  // XCD's L2, measured no faster — 0.314 vs 0.303 ms at 64 ch @192, 0.602 vs 0.572 at 32 ch @384: the kernel is bound by its
  // staging latency, not by the 3.2x HBM bytes of round 2's PMC pass)
  if ((int)blockIdx.x < total) prefetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < total; tile += parts) {
    __syncthreads();  // the previous tile has been consumed
    if (xthr) {
#pragma unroll
      for (int j = 0; j < XSLOT; ++j) xs[j * PLANE + xry * ROWS + xrx] = rxv[j];
    }
#pragma unroll
    for (int j = 0; j < DSLOT; ++j) {
      const int oc = doc0 + 6 * j;
      if (oc < CPG) {
        float* d0 = dys + (dr * TW + dq * 4) * DLD + oc;
        d0[0] = rdv[j].x; d0[DLD] = rdv[j].y; d0[2 * DLD] = rdv[j].z; d0[3 * DLD] = rdv[j].w;
      }
    }
    __syncthreads();
    if (tile + parts < total) prefetch(tile + parts);
    const float* __restrict__ ab = dys + kq * DLD + lm;
    const float* __restrict__ bb = xs + lm * PLANE + ky * ROWS + kq;
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int s = 0; s < TW / KL; ++s) {
        const float a = ab[(r * TW + s * KL) * DLD];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float b = bb[r * ROWS + s * KL + kx];
          if constexpr (MF == 32)
            acc[kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[kx], 0, 0, 0);
          else
            acc[kx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[kx], 0, 0, 0);
        }
      }
  }
  // C/D layout: column = lane % MF = ci, row = oc: 32x32 (reg & 3) + 8 (reg >> 2) + 4 kq; 16x16 4 kq + reg
  float* __restrict__ dst = part + (long)blockIdx.x * C * RCPG * 9 + (long)g * CPG * RCPG * 9;
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int r = 0; r < NACC; ++r) {
      const int oc = MF == 32 ? (r & 3) + 8 * (r >> 2) + 4 * kq : 4 * kq + r;
      if (oc / RCPG == lm / RCPG) dst[((long)oc * RCPG + lm % RCPG) * 9 + ky * 3 + kx] = acc[kx][r];
    }
}

// -------------------------------------------------------------------------------------
// Weight gradients in which ONE of the two channel counts is 1:
//   FLIP = false, KS = 4, S = 2: Conv2d(1, C, 4, 2, 1) — the encoder's first layer (1 -> 256 at 384x384) and the
//     PatchGAN's (1 -> 64):   dW[c][tap] = sum_p big[c][p] * small[S p - pad + tap],   big = dy, small = the image;
//   FLIP = true,  KS = 3, S = 1: Conv2d(C, 1, 3, 1, 1) — the decoder's output convolution (128 -> 1):
//     dW[c][tap] = sum_p x[c][p + tap - 1] dy[p] = sum_q big[c][q] * small[q + 1 - tap],   big = x, small = dy.
// Either way a (C x pixels) . (pixels x taps) product that is HBM-bound on the single read of `big` (1.2 / 2.4 GB
// at B = 32).  One v_mfma_f32_16x16x4_f32 tile = 16 channels x 16 tap columns; a block of 4 waves covers 64 channels
// and walks 4 x 32 pixel tiles: `big` staged as it lies in memory (bgs[c][pixel], row stride 132 words: the
// 16-byte stores and the fragment reads — lane = (channel, pixel k) -> bank 4 c + k — are conflict-free), the
// single-channel patch with a row stride that keeps the taps of a pixel on different banks; tile i + 1 is prefetched into registers while
// tile i is multiplied.  (Generic dconv_wgrad_kernel: 1.49 ms on the first layer; c1conv3_wgrad: 1.45 ms.)
// -------------------------------------------------------------------------------------
template <int KS, int S, bool FLIP, typename BT = float>
__global__ __launch_bounds__(256) void c1_wgrad_mfma_kernel(const BT* __restrict__ big, const float* __restrict__ small,
                                                            float* __restrict__ part, int NB, int C, int Hs, int Ws,
                                                            int Hb, int Wb, int pad, int tiles_x, int tiles_y, int parts) {
  // 4 x 32 pixel tiles: a tile row of one channel is one whole 128-byte line of `big` (16-pixel-wide tiles read half
  // lines and ran at 3.2 TB/s of useful bytes)
  constexpr int TH = 4, TW = 32, NPIX = TH * TW, CB = 64, DLD = NPIX + 4, KK = KS * KS;  // bgs[c][pixel], stride 132
  constexpr int PH = S * (TH - 1) + KS, PW = S * (TW - 1) + KS;      // 10 x 66 / 6 x 34
  constexpr int ROWS = S == 2 ? 68 : 36;                             // tap (ky, kx) -> bank 4 ky + kx
  constexpr int PSLOT = (PH * PW + 255) / 256;
  static_assert(KK <= 16, "taps fit the 16 MFMA columns");
  __shared__ __attribute__((aligned(16))) float bgs[CB * DLD];
  __shared__ float xs[PH * ROWS];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int c0 = blockIdx.y * CB;
  const int lm = lane & 15, kq = lane >> 4;          // A: channel 16 wave + lm;  B: tap lm;  k lane = pixel
  const int tky = lm < KK ? lm / KS : 0, tkx = lm < KK ? lm % KS : 0;   // columns >= KK compute garbage, never stored
  const int tapoff = FLIP ? (KS - 1 - tky) * ROWS + (KS - 1 - tkx) : tky * ROWS + tkx;
  const long HWb = (long)Hb * Wb;

  g3_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // staging maps: big — thread -> (q = t % 8, row = (t / 8) % 4, cl0 = t / 32), slot j = channel cl0 + 8 j;
  // patch — elements t + 256 j of the PH x PW patch
  const int dq = t & 7, dr = (t >> 3) & 3, dcl = t >> 5;
  float4 rd[8];
  float rp[PSLOT];
  const int per_img = tiles_x * tiles_y, total = NB * per_img;
  auto prefetch = [&](int tile) {
    const int n = tile / per_img;
    const int tr = tile - n * per_img;
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * TW;
    const int oy = oy0 + dr, ox = ox0 + dq * 4;
    const bool dok = oy < Hb && ox < Wb;            // Wb % 4 == 0
    const BT* __restrict__ dp = big + ((long)n * C + c0 + dcl) * HWb + (dok ? (long)oy * Wb + ox : 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 v = ld4(dp + (long)(8 * j) * HWb);
      rd[j] = dok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float* __restrict__ xp = small + (long)n * Hs * Ws;
    const int py0 = FLIP ? oy0 - pad : S * oy0 - pad, px0 = FLIP ? ox0 - pad : S * ox0 - pad;
#pragma unroll
    for (int j = 0; j < PSLOT; ++j) {
      const int idx = t + j * 256;
      const int ry = idx / PW, rx = idx - ry * PW;
      const int iy = py0 + ry, ix = px0 + rx;
      const bool ok = idx < PH * PW && iy >= 0 && iy < Hs && ix >= 0 && ix < Ws;
      const float v = xp[ok ? (long)iy * Ws + ix : 0];
      rp[j] = ok ? v : 0.f;
    }
  };
  if ((int)blockIdx.x < total) prefetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < total; tile += parts) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j)
      *reinterpret_cast<float4*>(bgs + (dcl + 8 * j) * DLD + dr * TW + dq * 4) = rd[j];
#pragma unroll
    for (int j = 0; j < PSLOT; ++j) {
      const int idx = t + j * 256;
      const int ry = idx / PW, rx = idx - ry * PW;
      if (idx < PH * PW) xs[ry * ROWS + rx] = rp[j];
    }
    __syncthreads();
    if (tile + parts < total) prefetch(tile + parts);
    const float* __restrict__ ab = bgs + (wave * 16 + lm) * DLD + kq;
    const float* __restrict__ bb = xs + tapoff + S * kq;
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int s = 0; s < TW / 4; ++s) {
        const float a = ab[r * TW + s * 4];
        const float b = bb[S * r * ROWS + 4 * S * s];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
      }
  }
  // C/D: column = lane % 16 = tap, row = channel 4 kq + reg of this wave's 16
  if (lm < KK) {
    float* __restrict__ dst = part + (long)blockIdx.x * C * KK + (long)(c0 + wave * 16) * KK;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(4 * kq + r) * KK + lm] = acc[r];
  }
}

}  // namespace

namespace wfae {
// Conv2d(1, C, 4, 2, 1) (flip = 0: big = dy (N,C,H/2,W/2), small = x (N,1,H,W)) or Conv2d(C, 1, 3, 1, 1) (flip = 1:
// big = x (N,C,H,W), small = dy (N,1,H,W)) weight gradient on c1_wgrad_mfma_kernel; WFAE_ERR_UNSUPPORTED for shapes
// it does not take (odd sizes, C % 64, unaligned `big`, workspace).  `big` is fp32 or bf16, `small` always fp32.
template <typename BT>
static int c1_wgrad_mfma_t(int flip, const BT* big, const float* small, float* dw, int NB, int C, int H, int W,
                           int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  const int Hb = flip ? H : H / 2, Wb = flip ? W : W / 2;
  if ((!flip && ((H & 1) || (W & 1))) || C % 64 != 0 || Wb % 4 != 0 ||
      (reinterpret_cast<uintptr_t>(big) & 15) != 0 || C / 64 > 65535)
    return WFAE_ERR_UNSUPPORTED;
  const int tiles_x = cdiv(Wb, 32), tiles_y = cdiv(Hb, 4);
  const long total = (long)NB * tiles_x * tiles_y;
  const int gy = C / 64, KK = flip ? 9 : 16;
  long parts = cdiv(1024, gy);
  if (parts > total / 2) parts = total / 2 > 0 ? total / 2 : 1;
  const size_t out_elems = (size_t)C * KK;
  if (!ws || (size_t)parts * out_elems * sizeof(float) > ws_bytes) return WFAE_ERR_UNSUPPORTED;
  dim3 grid((unsigned)parts, gy);
  if (flip)
    hipLaunchKernelGGL((c1_wgrad_mfma_kernel<3, 1, true, BT>), grid, dim3(256), 0, st, big, small, (float*)ws, NB, C, H, W, Hb,
                       Wb, 1, tiles_x, tiles_y, (int)parts);
  else
    hipLaunchKernelGGL((c1_wgrad_mfma_kernel<4, 2, false, BT>), grid, dim3(256), 0, st, big, small, (float*)ws, NB, C, H, W, Hb,
                       Wb, 1, tiles_x, tiles_y, (int)parts);
  int rc = check_launch("c1_wgrad_mfma");
  if (rc) return rc;
  return slab_reduce((const float*)ws, dw, nullptr, (long)out_elems, 1, (int)parts, accumulate, st);
}
int c1_wgrad_mfma(int flip, const float* big, const float* small, float* dw, int NB, int C, int H, int W, int accumulate,
                  void* ws, size_t ws_bytes, hipStream_t st) {
  return c1_wgrad_mfma_t(flip, big, small, dw, NB, C, H, W, accumulate, ws, ws_bytes, st);
}
int c1_wgrad_mfma(int flip, const bf16_t* big, const float* small, float* dw, int NB, int C, int H, int W, int accumulate,
                  void* ws, size_t ws_bytes, hipStream_t st) {
  return c1_wgrad_mfma_t(flip, big, small, dw, NB, C, H, W, accumulate, ws, ws_bytes, st);
}

// weight gradient of the grouped 3x3 conv for 4 / 8 / 16 / 32 channels per group on the MFMA kernel above;
// returns WFAE_ERR_UNSUPPORTED otherwise
template <typename T>
static int gconv3_wgrad_mfma_t(const T* dy, const T* x, float* dw, int NB, int C, int H, int W, int groups,
                               int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  const int cpg = C / groups;
  if (!(cpg == 4 || cpg == 8 || cpg == 16 || cpg == 32) || W % 4 != 0 || (reinterpret_cast<uintptr_t>(dy) & 15) != 0)
    return WFAE_ERR_UNSUPPORTED;
  const int vcpg = cpg == 32 ? 32 : 16;        // channels per MFMA tile ("virtual group")
  if (C % vcpg != 0 || C / vcpg > 65535) return WFAE_ERR_UNSUPPORTED;
  const int vgroups = C / vcpg;
  const int tiles_x = cdiv(W, 16), tiles_y = cdiv(H, 8);
  const long total = (long)NB * tiles_x * tiles_y;
  const size_t out_elems = (size_t)C * cpg * 9;
  long parts = cdiv(1536, vgroups);  // ~6 blocks of 3 waves per CU
  if (parts > total / 2) parts = total / 2 > 0 ? total / 2 : 1;  // >= 2 tiles per block: the register pipeline needs a successor
  while (parts > 1 && (size_t)parts * out_elems * sizeof(float) > ws_bytes) --parts;
  if (!ws || (size_t)parts * out_elems * sizeof(float) > ws_bytes)
    return fail(WFAE_ERR_WORKSPACE, "gconv3x3_bwd_weight: workspace %zu too small", ws_bytes);
  dim3 grid((unsigned)parts, vgroups), block(192);
  float* part = (float*)ws;
#define WFAE_G3W(V_, R_)                                                                                         \
  hipLaunchKernelGGL((gconv3_wgrad_mfma_kernel<V_, R_, T>), grid, block, 0, st, dy, x, part, NB, C, H, W, tiles_x, \
                     tiles_y, (int)parts)
  if (cpg == 32) WFAE_G3W(32, 32);
  else if (cpg == 16) WFAE_G3W(16, 16);
  else if (cpg == 8) WFAE_G3W(16, 8);
  else WFAE_G3W(16, 4);
#undef WFAE_G3W
  int rc = check_launch("gconv3_wgrad_mfma");
  if (rc) return rc;
  return slab_reduce(part, dw, nullptr, (long)out_elems, 1, (int)parts, accumulate, st);
}
int gconv3_wgrad_mfma(const float* dy, const float* x, float* dw, int NB, int C, int H, int W, int groups, int accumulate,
                      void* ws, size_t ws_bytes, hipStream_t st) {
  return gconv3_wgrad_mfma_t(dy, x, dw, NB, C, H, W, groups, accumulate, ws, ws_bytes, st);
}
int gconv3_wgrad_mfma(const bf16_t* dy, const bf16_t* x, float* dw, int NB, int C, int H, int W, int groups, int accumulate,
                      void* ws, size_t ws_bytes, hipStream_t st) {
  return gconv3_wgrad_mfma_t(dy, x, dw, NB, C, H, W, groups, accumulate, ws, ws_bytes, st);
}

// weight gradient of the grouped 3x3 conv for cpg in {4, 8, 16}; returns WFAE_ERR_UNSUPPORTED otherwise
int gconv3_wgrad_valu(const float* dy, const float* x, float* dw, int NB, int C, int H, int W, int groups,
                      int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  const int cpg = C / groups;
  if (!(cpg == 4 || cpg == 8 || cpg == 16)) return WFAE_ERR_UNSUPPORTED;
  if (cpg >= 16) return WFAE_ERR_UNSUPPORTED;   // measured: 16 ch/group 0.55 ms on the MFMA path vs 1.3 ms here
  // output channels per wave: 4 ch/group runs 1.4x faster with 72 accumulators per lane (twice the occupancy),
  // 8 ch/group is faster with 144 (it would re-read x twice as often otherwise)
  int ocw = 144 / (cpg * 9);
  if (cpg == 4) ocw = 2;
  const int units = groups * (cpg / ocw);
  if (units % 4 != 0) return WFAE_ERR_UNSUPPORTED;
  if (cpg == 4 && groups % 4 != 0) return WFAE_ERR_UNSUPPORTED;
  const int gy = units / 4;
  const int tiles_x = cdiv(W, 64), tiles_y = cdiv(H, 8);
  const long total = (long)NB * tiles_x * tiles_y;
  const size_t out_elems = (size_t)C * cpg * 9;
  long parts = 768 / gy;
  if (parts < 1) parts = 1;
  if (parts > total) parts = total;
  while (parts > 1 && (size_t)parts * out_elems * sizeof(float) > ws_bytes) --parts;
  if (!ws || (size_t)parts * out_elems * sizeof(float) > ws_bytes)
    return fail(WFAE_ERR_WORKSPACE, "gconv3x3_bwd_weight: workspace %zu too small", ws_bytes);
  dim3 grid((unsigned)parts, gy), block(256);
  float* part = (float*)ws;
  if (cpg == 4 && ocw == 2)
    hipLaunchKernelGGL((gconv3_wgrad_kernel<4, 2>), grid, block, 0, st, dy, x, part, NB, C, H, W, tiles_x, tiles_y, (int)parts);
  else if (cpg == 8 && ocw == 1)
    hipLaunchKernelGGL((gconv3_wgrad_kernel<8, 1>), grid, block, 0, st, dy, x, part, NB, C, H, W, tiles_x, tiles_y, (int)parts);
  else if (cpg == 4)
    hipLaunchKernelGGL((gconv3_wgrad_kernel<4, 4>), grid, block, 0, st, dy, x, part, NB, C, H, W, tiles_x, tiles_y, (int)parts);
  else if (cpg == 8)
    hipLaunchKernelGGL((gconv3_wgrad_kernel<8, 2>), grid, block, 0, st, dy, x, part, NB, C, H, W, tiles_x, tiles_y, (int)parts);
  else
    hipLaunchKernelGGL((gconv3_wgrad_kernel<16, 1>), grid, block, 0, st, dy, x, part, NB, C, H, W, tiles_x, tiles_y, (int)parts);
  int rc = check_launch("gconv3_wgrad");
  if (rc) return rc;
  return slab_reduce(part, dw, nullptr, (long)out_elems, 1, (int)parts, accumulate, st);
}
}  // namespace wfae
