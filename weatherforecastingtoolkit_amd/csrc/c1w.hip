// c1w.hip — weight gradients of the Bottleneck's 1x1 convolutions (pipeline/models/ae_64x8x8_lin.py:15,19):
//   dW[a][b] = sum over images and pixels  P[img][a][p] * Q[img][b][p]
// P = the operand with MORE rows (C channels: x for the C -> C/4 convolution, dy for the C/4 -> C one), Q the one with
// fewer (C/4); both are NCHW activation tensors whose rows are contiguous along the contraction (the pixels).
//
// gemm.hip serves this product by transposing every loaded float4 into its k-major LDS images (four ds_write_b32 per
// 16-byte load) and, at fp32 precision, on v_mfma_f32_32x32x2_f32 or with the bf16 split repeated by every consuming
// wave: 2.6 - 2.7 TB/s on a product whose arithmetic intensity says HBM-bound (profiles/r02_v8_*: 25 ms per step).  Both
// MFMA operands of a weight gradient are K-contiguous ROWS, which is the layout the matrix core's bf16 fragments want
// (8 consecutive k per lane = one ds_read_b128 of a 64-byte LDS row, the A operand of splitgemm.hip) — no transpose:
//   * fp32 tensors: a float4 (4 pixels of one channel row) is split once into the exact (h, m, l) bf16 planes on its way
//     into LDS (common.h split3: h + m + l == x), six v_mfma_f32_32x32x16_bf16 per tile product, fp32 accumulation;
//   * bf16-stored tensors ('medium'): 16-byte pieces go to LDS as they are, one product per tile;
//   * PRO: the BatchNorm + GELU in front of the convolution (a = gelu(x * bn_scale[c] + bn_shift[c]), c = the row's
//     channel) rebuilt in the loader of whichever operand x is, as wfae_conv1x1_bwd_weight_bnact does;
//   * split-K over (image, pixel chunk): every block owns a BM x BN tile of dW for one chunk and writes an fp32 slab;
//     slab_reduce adds the slabs in a fixed order (deterministic), transposing when P is the Cin side.
// Block = 8 waves as 4 (P rows) x 2 (Q rows), wave tile 32 TM x 32 TN, K-step 32, two LDS stages, the software pipeline of
// splitgemm.hip (one barrier per K-step, fragments of the next k-slab read under the MFMAs of the current one).
#include "common.h"

using namespace wfae;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct C1WP {
  const void* P;             // [NB][MP][HW]   (float or bf16_t)
  const void* Q;             // [NB][MQ][HW]
  float* slab;               // [splits][MP][MQ]
  const float* pro_scale;    // PRO 1: per row of P; PRO 2: per row of Q
  const float* pro_shift;
  int MP, MQ, HW;
  int cpi;                   // K chunks per image
  int kc;                    // pixels per chunk (multiple of 32)
  int ptiles;
};

constexpr int WNT = 512, WBK = 32;

__device__ __forceinline__ unsigned off_row(int r, int c) { return (unsigned)(r * 64 + ((c ^ ((r >> 2) & 3)) << 4)); }

__device__ __forceinline__ void split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  h = pack_bf16(a, b);
  const float ra = a - bf16_lo(h), rb = b - bf16_hi(h);
  m = pack_bf16(ra, rb);
  l = pack_bf16(ra - bf16_lo(m), rb - bf16_hi(m));
}

// T: element type of both tensors; NP = 3 (float: exact split) or 1 (bf16); PRO: 0 none, 1 on P, 2 on Q
template <typename T, int TM, int TN, int PRO>
__global__ __launch_bounds__(WNT, 2) void c1w_kernel(C1WP p) {
  constexpr bool F32 = sizeof(T) == 4;
  constexpr int NP = F32 ? 3 : 1;
  constexpr int BM = 128 * TM, BN = 64 * TN;
  constexpr int EPC = F32 ? 4 : 8;                       // elements per 16-byte load
  constexpr int LPR = WBK / EPC;                         // loads per 64-byte-of-bf16 row and K-step: 8 (fp32) / 4 (bf16)
  constexpr int P_LD = BM * LPR, Q_LD = BN * LPR;        // loads per K-step
  constexpr int P_IT = (P_LD + WNT - 1) / WNT, Q_IT = (Q_LD + WNT - 1) / WNT;
  constexpr int A_PLANE_B = BM * 64, B_PLANE_B = BN * 64;
  constexpr int A_STAGE_B = NP * A_PLANE_B, STAGE_B = NP * (A_PLANE_B + B_PLANE_B);
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_B];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int pt = blockIdx.x % p.ptiles, qt = blockIdx.x / p.ptiles;
  const int m0 = pt * BM, n0 = qt * BN;
  const int img = blockIdx.y / p.cpi, chunk = blockIdx.y - img * p.cpi;
  const int k_begin = chunk * p.kc;
  const int k_end = min(p.HW, k_begin + p.kc);
  const int nsteps = (k_end - k_begin) / WBK;

  // ---- loaders: load j of a thread covers row (idx / LPR), elements (idx % LPR) * EPC .. of the K-step, idx = t + j * WNT
  const T* Pb = reinterpret_cast<const T*>(p.P) + (long)img * p.MP * p.HW + k_begin;
  const T* Qb = reinterpret_cast<const T*>(p.Q) + (long)img * p.MQ * p.HW + k_begin;
  const T* p_src[P_IT];
  const T* q_src[Q_IT];
  unsigned p_dst[P_IT], q_dst[Q_IT];
  bool p_on[P_IT], q_on[Q_IT];
  float p_s[P_IT], p_h[P_IT], q_s[Q_IT], q_h[Q_IT];
#pragma unroll
  for (int j = 0; j < P_IT; ++j) {
    const int idx = t + j * WNT, r = idx / LPR, e = idx % LPR;
    p_on[j] = idx < P_LD;
    const int row = min(m0 + r, p.MP - 1);             // rows beyond the matrix: clamped, their products are never stored
    p_src[j] = Pb + (long)row * p.HW + e * EPC;
    // 64-byte LDS row of 32 bf16: element e * EPC sits in 16-byte chunk (e * EPC) / 8, byte (e * EPC % 8) * 2 of it
    p_dst[j] = off_row(r, (e * EPC) >> 3) + (unsigned)(((e * EPC) & 7) * 2);
    if constexpr (PRO == 1) {
      p_s[j] = p.pro_scale[row];
      p_h[j] = p.pro_shift[row];
    }
  }
#pragma unroll
  for (int j = 0; j < Q_IT; ++j) {
    const int idx = t + j * WNT, r = idx / LPR, e = idx % LPR;
    q_on[j] = idx < Q_LD;
    const int row = min(n0 + r, p.MQ - 1);
    q_src[j] = Qb + (long)row * p.HW + e * EPC;
    q_dst[j] = off_row(r, (e * EPC) >> 3) + (unsigned)(((e * EPC) & 7) * 2);
    if constexpr (PRO == 2) {
      q_s[j] = p.pro_scale[row];
      q_h[j] = p.pro_shift[row];
    }
  }
  u32x4 rp[P_IT], rq[Q_IT];
  auto load_global = [&]() {
#pragma unroll
    for (int j = 0; j < P_IT; ++j)
      if (P_LD % WNT == 0 || p_on[j]) rp[j] = *reinterpret_cast<const u32x4*>(p_src[j]);
#pragma unroll
    for (int j = 0; j < Q_IT; ++j)
      if (Q_LD % WNT == 0 || q_on[j]) rq[j] = *reinterpret_cast<const u32x4*>(q_src[j]);
  };
  auto advance = [&](bool more) {
#pragma unroll
    for (int j = 0; j < P_IT; ++j) p_src[j] += more ? WBK : 0;
#pragma unroll
    for (int j = 0; j < Q_IT; ++j) q_src[j] += more ? WBK : 0;
  };
  // one loaded 16-byte piece -> LDS: fp32: 4 values -> (optional activation) -> three planes of 8 bytes; bf16: as it is
  auto put = [&](unsigned char* base, int plane_b, unsigned dst, u32x4 r, bool act, float s, float h) {
    if constexpr (F32) {
      // (copied to scalars first: __builtin_bit_cast applied to an ext-vector ELEMENT expression read element 0 for all four)
      const unsigned u0 = r[0], u1 = r[1], u2 = r[2], u3 = r[3];
      float v0 = __uint_as_float(u0), v1 = __uint_as_float(u1), v2 = __uint_as_float(u2), v3 = __uint_as_float(u3);
      if (act) {   // bn_act_fwd_kernel<GELU>'s arithmetic
        v0 = gelu_f(fmaf(v0, s, h)); v1 = gelu_f(fmaf(v1, s, h));
        v2 = gelu_f(fmaf(v2, s, h)); v3 = gelu_f(fmaf(v3, s, h));
      }
      unsigned h0, m0_, l0, h1, m1, l1;
      split_pair(v0, v1, h0, m0_, l0);
      split_pair(v2, v3, h1, m1, l1);
      *reinterpret_cast<u32x2*>(base + dst) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(base + plane_b + dst) = u32x2{m0_, m1};
      *reinterpret_cast<u32x2*>(base + 2 * plane_b + dst) = u32x2{l0, l1};
    } else {
      if (act) {
        r.x = pack_bf16(gelu_f(fmaf(bf16_lo(r.x), s, h)), gelu_f(fmaf(bf16_hi(r.x), s, h)));
        r.y = pack_bf16(gelu_f(fmaf(bf16_lo(r.y), s, h)), gelu_f(fmaf(bf16_hi(r.y), s, h)));
        r.z = pack_bf16(gelu_f(fmaf(bf16_lo(r.z), s, h)), gelu_f(fmaf(bf16_hi(r.z), s, h)));
        r.w = pack_bf16(gelu_f(fmaf(bf16_lo(r.w), s, h)), gelu_f(fmaf(bf16_hi(r.w), s, h)));
      }
      *reinterpret_cast<u32x4*>(base + dst) = r;
    }
  };
  auto store_lds = [&](int buf) {
    unsigned char* s = smem + buf * STAGE_B;
#pragma unroll
    for (int j = 0; j < P_IT; ++j)
      if (P_LD % WNT == 0 || p_on[j]) put(s, A_PLANE_B, p_dst[j], rp[j], PRO == 1, PRO == 1 ? p_s[j] : 0.f, PRO == 1 ? p_h[j] : 0.f);
#pragma unroll
    for (int j = 0; j < Q_IT; ++j)
      if (Q_LD % WNT == 0 || q_on[j])
        put(s + A_STAGE_B, B_PLANE_B, q_dst[j], rq[j], PRO == 2, PRO == 2 ? q_s[j] : 0.f, PRO == 2 ? q_h[j] : 0.f);
  };

  // ---- fragments: rows of both operands, one ds_read_b128 per (32-row tile, plane, k-slab)
  const int wm0 = (wave >> 1) * (32 * TM), wn0 = (wave & 1) * (32 * TN);
  const int r31 = lane & 31, lh = lane >> 5;
  const unsigned a_rd = (unsigned)((wm0 + r31) * 64), b_rd = (unsigned)((wn0 + r31) * 64);
  const int a_x = (r31 >> 2) & 3;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  struct Frag {
    bf16x8 a[TM][NP], b[TN][NP];
  };
  auto read_frag = [&](Frag& f, int buf, int ks) {
    const unsigned char* s = smem + buf * STAGE_B;
    const unsigned a_c = (unsigned)(((2 * ks + lh) ^ a_x) << 4);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        f.a[i][pl] = *reinterpret_cast<const bf16x8*>(s + pl * A_PLANE_B + a_rd + i * (32 * 64) + a_c);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        f.b[j][pl] = *reinterpret_cast<const bf16x8*>(s + A_STAGE_B + pl * B_PLANE_B + b_rd + j * (32 * 64) + a_c);
  };
  auto mfma_frag = [&](const Frag& f) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        f32x16 c = acc[i][j];   // smallest terms first
        if constexpr (NP == 3) {
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][2], f.b[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][1], f.b[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][1], f.b[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][1], c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][0], c, 0, 0, 0);
        acc[i][j] = c;
      }
  };

  if (nsteps > 0) {
    Frag f0, f1;
    load_global();
    advance(nsteps > 1);
    store_lds(0);
    load_global();   // K-step 1 (or 0 again when there is only one: stored to the idle stage, never read)
    advance(nsteps > 2);
    __syncthreads();
    read_frag(f0, 0, 0);
    for (int st = 0; st < nsteps; ++st) {
      const int cur = st & 1;
      read_frag(f1, cur, 1);
      store_lds(cur ^ 1);   // K-step st + 1; in the last iteration a stale copy nobody reads
      load_global();        // K-step st + 2
      advance(st + 3 < nsteps);
      mfma_frag(f0);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      read_frag(f0, cur ^ 1, 0);
      mfma_frag(f1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: the partial tile of this (image, chunk) into its slab.  Accumulator register q of lane (r31, lh) is
  // C[(q & 3) + 8 (q >> 2) + 4 lh][r31] of its 32 x 32 tile: 32 lanes write 128 contiguous bytes of a slab row.
  float* __restrict__ Cb = p.slab + (long)blockIdx.y * p.MP * p.MQ;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn0 + 32 * j + r31;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = m0 + wm0 + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * lh;
        if (row < p.MP && col < p.MQ) Cb[(long)row * p.MQ + col] = acc[i][j][q];
      }
    }
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <typename T, int PRO>
int launch_c1w(C1WP& p, int NB, hipStream_t st, const char* what) {
  // tile: 256 x 128 (MQ > 64), 256 x 64 (MQ <= 64, MP > 128), 128 x 64 (MP <= 128)
  const int tm = p.MP > 128 ? 2 : 1, tn = p.MQ > 64 ? 2 : 1;
  const int bm = 128 * tm, bn = 64 * tn;
  p.ptiles = cdiv(p.MP, bm);
  const int qtiles = cdiv(p.MQ, bn);
  const dim3 grid((unsigned)(p.ptiles * qtiles), (unsigned)(NB * p.cpi)), block(WNT);
  if (tm == 2 && tn == 2) hipLaunchKernelGGL((c1w_kernel<T, 2, 2, PRO>), grid, block, 0, st, p);
  else if (tm == 2) hipLaunchKernelGGL((c1w_kernel<T, 2, 1, PRO>), grid, block, 0, st, p);
  else if (tn == 2) hipLaunchKernelGGL((c1w_kernel<T, 1, 2, PRO>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((c1w_kernel<T, 1, 1, PRO>), grid, block, 0, st, p);
  return check_launch(what);
}

inline bool c1w_shape_ok(int Cin, int Cout, int HW) {
  return HW % 32 == 0 && Cin % 32 == 0 && Cout % 32 == 0 && Cin >= 32 && Cout >= 32;
}

// dw (Cout, Cin) (+)= dy (NB,Cout,HW) . x'(NB,Cin,HW)^T, x' = x or gelu(x * bn_scale + bn_shift)
template <typename T>
int c1w_impl(const T* dy, const T* x, const float* bn_scale, const float* bn_shift, float* dw, int NB, int Cin, int Cout, int HW,
             int accumulate, void* ws, size_t ws_bytes, hipStream_t st, const char* what) {
  WFAE_REQUIRE(dy && x && dw && ws, WFAE_ERR_NULL_POINTER, "%s: null pointer", what);
  WFAE_REQUIRE(NB > 0 && Cin > 0 && Cout > 0 && HW > 0 && NB <= 65535, WFAE_ERR_BAD_SHAPE, "%s: bad shape", what);
  WFAE_REQUIRE(c1w_shape_ok(Cin, Cout, HW) && al16(dy) && al16(x), WFAE_ERR_UNSUPPORTED,
               "%s: needs HW %% 32 == 0, channel counts %% 32 == 0, 16-byte aligned tensors (Cin %d, Cout %d, HW %d)", what, Cin,
               Cout, HW);
  // P = the operand with more rows; the slab is [P rows][Q rows]
  const bool x_is_p = Cin >= Cout;
  C1WP p = {};
  p.P = x_is_p ? (const void*)x : (const void*)dy;
  p.Q = x_is_p ? (const void*)dy : (const void*)x;
  p.MP = x_is_p ? Cin : Cout;
  p.MQ = x_is_p ? Cout : Cin;
  p.HW = HW;
  p.pro_scale = bn_scale;
  p.pro_shift = bn_shift;
  const size_t slab = (size_t)Cin * Cout * sizeof(float);
  // K chunks per image: ~1024 blocks in flight, chunks of at least 256 pixels, slabs within the workspace
  const int tiles = cdiv(p.MP, p.MP > 128 ? 256 : 128) * cdiv(p.MQ, p.MQ > 64 ? 128 : 64);
  int cpi = cdiv(1024, (long)tiles * NB);
  const int max_cpi = HW / 256 > 0 ? HW / 256 : 1;
  if (cpi > max_cpi) cpi = max_cpi;
  if (cpi < 1) cpi = 1;
  while (cpi > 1 && (size_t)NB * cpi * slab > ws_bytes) --cpi;
  WFAE_REQUIRE((size_t)NB * cpi * slab <= ws_bytes, WFAE_ERR_WORKSPACE, "%s: workspace %zu < %zu", what, ws_bytes, (size_t)NB * slab);
  p.cpi = cpi;
  p.kc = cdiv(cdiv(HW, cpi), WBK) * WBK;
  p.cpi = cdiv(HW, p.kc);     // chunks that actually hold pixels
  p.slab = (float*)ws;
  int rc;
  const int pro = bn_scale ? (x_is_p ? 1 : 2) : 0;
  if (pro == 0) rc = launch_c1w<T, 0>(p, NB, st, what);
  else if (pro == 1) rc = launch_c1w<T, 1>(p, NB, st, what);
  else rc = launch_c1w<T, 2>(p, NB, st, what);
  if (rc) return rc;
  // slab [P][Q]: P = Cout -> dw as it is; P = Cin -> transposed into dw (Cout, Cin)
  return slab_reduce((const float*)ws, dw, nullptr, (long)Cin * Cout, p.MQ, NB * p.cpi, accumulate, st, x_is_p ? p.MP : 0);
}

}  // namespace

extern "C" {

int wfae_c1w_supported(int Cin, int Cout, int HW) { return c1w_shape_ok(Cin, Cout, HW) ? 1 : 0; }

int wfae_c1w_bwd_weight(const float* dy, const float* x, const float* bn_scale, const float* bn_shift, float* dw, int NB, int Cin,
                        int Cout, int HW, int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE((bn_scale != nullptr) == (bn_shift != nullptr), WFAE_ERR_NULL_POINTER, "c1w_bwd_weight: scale / shift go together");
  return c1w_impl(dy, x, bn_scale, bn_shift, dw, NB, Cin, Cout, HW, accumulate, ws, ws_bytes, (hipStream_t)stream, "c1w_bwd_weight");
}

int wfae_c1w_bwd_weight_bf16(const uint16_t* dy, const uint16_t* x, const float* bn_scale, const float* bn_shift, float* dw, int NB,
                             int Cin, int Cout, int HW, int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE((bn_scale != nullptr) == (bn_shift != nullptr), WFAE_ERR_NULL_POINTER, "c1w_bwd_weight_bf16: scale / shift go together");
  return c1w_impl(dy, x, bn_scale, bn_shift, dw, NB, Cin, Cout, HW, accumulate, ws, ws_bytes, (hipStream_t)stream,
                  "c1w_bwd_weight_bf16");
}

}  // extern "C"
