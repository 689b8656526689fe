// dconv.hip — direct (im2col-free) VALU convolutions on NCHW tiles staged in LDS.
// Used where the channel count per group is too small for a 32x32 MFMA tile:
//   * grouped 3x3 pad-1 conv of Bottleneck (ae_64x8x8_lin.py:17): 4..32 ch/group
//   * 128->1 3x3 output conv (:84) and its data gradient (1->128)
//   * 1->256 4x4 stride-2 input conv (:31 with in_ch = 1)
// Each block stages an input patch (+halo) with coalesced row reads, weights
// are wave-uniform (scalar loads), every thread owns one output pixel and OCB
// output channels in registers.
#include "common.h"

using namespace wfae;

namespace {

struct DConvP {
  const float* x;
  const float* w;
  const float* bias;
  float* y;
  int Cin, Cout, H, W, Ho, Wo, pad, groups;
  int tiles_x;
};

// TRANSPOSED: weights are read as the data-gradient operator of a stride-1
// conv: w'(o,i,tap) = w[(g*IG + i)][o][KK-1-tap] with the original tensor laid
// out [groups*IG][OG][KS][KS].
template <int KS, int S, int OCB, bool TRANSPOSED>
__global__ __launch_bounds__(256) void dconv_fwd_kernel(DConvP p) {
  constexpr int KK = KS * KS;
  constexpr int TIH = 15 * S + KS;
  constexpr int TIW = 15 * S + KS;
  constexpr int CIB = (S == 1) ? 8 : 2;
  __shared__ float xs[CIB][TIH][TIW + 1];

  const int t = threadIdx.x;
  const int tx = t & 15, ty = t >> 4;
  const int tile = blockIdx.x;
  const int oy0 = (tile / p.tiles_x) * 16, ox0 = (tile % p.tiles_x) * 16;
  const int n = blockIdx.z;
  const int IG = p.Cin / p.groups, OG = p.Cout / p.groups;
  const int o0 = blockIdx.y * OCB;
  const int g = o0 / OG;
  const int ol0 = o0 - g * OG;

  float acc[OCB];
#pragma unroll
  for (int o = 0; o < OCB; ++o) acc[o] = 0.f;

  const float* xg = p.x + ((long)n * p.Cin + (long)g * IG) * p.H * p.W;
  const int iy_base = oy0 * S - p.pad, ix_base = ox0 * S - p.pad;

  for (int ci0 = 0; ci0 < IG; ci0 += CIB) {
    const int cn = min(CIB, IG - ci0);
    __syncthreads();
    for (int idx = t; idx < cn * TIH * TIW; idx += 256) {
      const int c = idx / (TIH * TIW);
      const int r = idx - c * (TIH * TIW);
      const int ry = r / TIW, rx = r - ry * TIW;
      const int iy = iy_base + ry, ix = ix_base + rx;
      float v = 0.f;
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) v = xg[((long)(ci0 + c) * p.H + iy) * p.W + ix];
      xs[c][ry][rx] = v;
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
      p.y[(((long)n * p.Cout + o0 + o) * p.Ho + oy) * p.Wo + ox] = v;
    }
  }
}

template <int KS, int S, bool TR>
int launch_dconv(const DConvP& p, int NB, hipStream_t st) {
  const int OG = p.Cout / p.groups;
  const int tiles = p.tiles_x * cdiv(p.Ho, 16);
  dim3 block(256);
  if (OG % 8 == 0) {
    hipLaunchKernelGGL((dconv_fwd_kernel<KS, S, 8, TR>), dim3(tiles, p.Cout / 8, NB), block, 0, st, p);
  } else if (OG % 4 == 0) {
    hipLaunchKernelGGL((dconv_fwd_kernel<KS, S, 4, TR>), dim3(tiles, p.Cout / 4, NB), block, 0, st, p);
  } else {
    hipLaunchKernelGGL((dconv_fwd_kernel<KS, S, 1, TR>), dim3(tiles, p.Cout, NB), block, 0, st, p);
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
  DConvP p = {};
  p.x = x; p.w = w; p.bias = bias; p.y = y;
  p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W; p.pad = pad; p.groups = groups;
  p.Ho = (H + 2 * pad - KS) / stride + 1;
  p.Wo = (W + 2 * pad - KS) / stride + 1;
  p.tiles_x = cdiv(p.Wo, 16);
  hipStream_t st = (hipStream_t)stream;
  if (KS == 3 && stride == 1) return launch_dconv<3, 1, false>(p, NB, st);
  if (KS == 4 && stride == 2) return launch_dconv<4, 2, false>(p, NB, st);
  return fail(WFAE_ERR_UNSUPPORTED, "dconv_fwd: KS=%d stride=%d unsupported", KS, stride);
}

int wfae_dconv_bwd_data(const float* dy, const float* w, float* dx, int NB, int Cin, int Cout,
                        int H, int W, int KS, int pad, int groups, wfae_stream_t stream) {
  WFAE_REQUIRE(dy && w && dx, WFAE_ERR_NULL_POINTER, "dconv_bwd_data: null pointer");
  WFAE_REQUIRE(NB > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && groups > 0 && Cin % groups == 0 &&
                   Cout % groups == 0,
               WFAE_ERR_BAD_SHAPE, "dconv_bwd_data: bad shape");
  WFAE_REQUIRE(KS == 3 && 2 * pad == KS - 1, WFAE_ERR_UNSUPPORTED,
               "dconv_bwd_data: only 3x3 'same' stride-1 convolutions");
  WFAE_REQUIRE(NB <= 65535, WFAE_ERR_BAD_SHAPE, "dconv_bwd_data: batch > 65535");
  // data gradient = convolution of dy (Cout channels) producing Cin channels with
  // transposed + spatially flipped weights and pad' = KS-1-pad.
  DConvP p = {};
  p.x = dy; p.w = w; p.bias = nullptr; p.y = dx;
  p.Cin = Cout; p.Cout = Cin; p.H = H; p.W = W; p.Ho = H; p.Wo = W;
  p.pad = KS - 1 - pad; p.groups = groups;
  p.tiles_x = cdiv(p.Wo, 16);
  return launch_dconv<3, 1, true>(p, NB, (hipStream_t)stream);
}

int wfae_dconv_bwd_weight(const float* dy, const float* x, float* dw, int NB, int Cin, int Cout,
                          int H, int W, int KS, int stride, int pad, int groups, int accumulate,
                          void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(dy && x && dw, WFAE_ERR_NULL_POINTER, "dconv_bwd_weight: null pointer");
  WFAE_REQUIRE(NB > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && groups > 0 && Cin % groups == 0 &&
                   Cout % groups == 0,
               WFAE_ERR_BAD_SHAPE, "dconv_bwd_weight: bad shape");
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
  } else {
    return fail(WFAE_ERR_UNSUPPORTED, "dconv_bwd_weight: KS=%d stride=%d unsupported", KS, stride);
  }
#undef WFAE_WG
  int rc = check_launch("dconv_wgrad");
  if (rc) return rc;
  return slab_reduce((const float*)ws, dw, nullptr, (long)out_elems, 1, (int)parts, accumulate, st);
}

}  // extern "C"
