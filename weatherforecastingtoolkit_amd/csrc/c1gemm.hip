// c1gemm.hip — the Bottleneck's 1x1 convolutions (pipeline/models/ae_64x8x8_lin.py:15,19) forward and data gradient on the
// bf16 matrix pipe at fp32 accuracy, with the fp32 -> three-bf16-plane split done ONCE per loaded value on the way into LDS.
//
// Round 2 ran these GEMMs on gemm.hip's fp32 kernel with the split applied behind the LDS fragment reads ("PREC 2"): every
// wave that consumes a value splits it again (5.5 VALU instructions per value and consuming wave), which left the C >= 512
// shapes VALU-bound (matrix pipe 0.30-0.40 busy, profiles/r02_v8_pmc_conv1_gemm_wave_cycles.txt) and the C <= 256 shapes
// on v_mfma_f32_32x32x2_f32, whose time equals their HBM time (no overlap: 45-62 % of the ideal).  Here
//   Y[img][m][p] = sum_k W[m][k] f(X[img][k][p])          m < M (output channels), k < K (input channels), p < HW
// runs with
//   * W as three bf16 planes written once per step by c1_split_weights_kernel (W and W^T: the data gradient is the same
//     product with W^T as the weight), staged like splitgemm.hip's A operand;
//   * X read as fp32 dwordx4 straight from the NCHW tensor (whole 128-byte lines per k-row), optionally activated
//     (f = GELU(x * bn_scale[k] + bn_shift[k]): the fused BatchNorm + GELU prologue of gemm.hip, same arithmetic), split
//     into (h, m, l) in registers and stored as three bf16 k-row images [k][n] — the layout splitgemm.hip reads through
//     ds_read_b64_tr_b16 — so HBM traffic stays 4 bytes per element and each value is split once per BLOCK;
//   * six v_mfma_f32_32x32x16_bf16 per 32 x 32 x 16 tile product, fp32 accumulators (smallest terms first);
//   * the result tile staged through the (then idle) operand LDS and written as dwordx4 rows, with the epilogues of the
//     Bottleneck fused in: residual add (:22), BatchNorm sums of the result for the next BatchNorm (fp64 partial rows, the
//     format of wfae_conv1x1_fwd_stats), and the BatchNorm + GELU BACKWARD of the layer in front of the convolution:
//       BNRED  per-channel sum dU, sum dU * xhat (dU = dA * gelu'(u)) reduced while dA = W^T dT is still on chip — the
//              separate reduce pass over (dA, x) disappears; with store = 0 dA is not even written (recompute form);
//       BNDX   dX = gamma * invstd * (dU - mean dU - xhat * mean(dU xhat)) + residual gradient, from a recomputed dA.
// Block tiles 256 x 128, 128 x 256 and 64 x 256 (8 waves; wave tile 64 x 64 or 32 x 64), K-step 32, two LDS stages,
// register-staged global loads one K-step ahead, one barrier per K-step (the pipeline of splitgemm.hip).
#include "common.h"
#include <stdlib.h>

using namespace wfae;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

enum Epi { EPI_PLAIN = 0, EPI_BNRED = 1, EPI_BNDX = 2 };

struct C1P {
  const unsigned short* A;   // weight planes [3][M][K]
  long a_plane;
  const float* B;            // activations [NB][K][HW]
  float* C;                  // result [NB][M][HW]
  const float* res;          // PLAIN: residual [NB][M][HW] or null; BNRED / BNDX: the BatchNorm input x [NB][M][HW]
  const float* res2;         // BNDX: residual-branch gradient [NB][M][HW] or null
  const float* pro_scale;    // PRO: folded BatchNorm scale / shift of the INPUT channels [K]
  const float* pro_shift;
  const float* e_scale;      // BNRED / BNDX: folded scale / shift, batch mean / invstd of the OUTPUT channels [M]
  const float* e_shift;
  const float* e_mean;
  const float* e_invstd;
  const float* e_gamma;      // BNDX
  const float* e_coef;       // BNDX: [2 M] sums (sum dU, sum dU xhat) left by the finalize kernel
  float inv_count;           // BNDX: 1 / (NB * HW), 0 in eval mode
  double* part0;             // PLAIN: sums [ntiles][M]; BNRED: sum dU [ntiles][M]; null = off
  double* part1;             // PLAIN: sums of squares; BNRED: sum dU * xhat
  int M, K, HW;
  long N;                    // NB * HW
  int mtiles;
  int store;                 // BNRED: write the result tensor (0 = reduce only)
};

constexpr int CNT = 512, CBK = 32;

__device__ __forceinline__ unsigned off_row(int r, int c) { return (unsigned)(r * 64 + ((c ^ ((r >> 2) & 3)) << 4)); }
__device__ __forceinline__ unsigned sw_tr(int k) { return (unsigned)(((k & 3) << 2) | ((k >> 2) & 3)); }

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));   // v_cvt_pk_bf16_f32, RNE
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// (a, b) -> packed (h, m, l) pairs with a == h + m + l exactly (common.h split3, two values per conversion)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  h = pk_bf16(a, b);
  const float ra = a - bf_lo(h), rb = b - bf_hi(h);
  m = pk_bf16(ra, rb);
  l = pk_bf16(ra - bf_lo(m), rb - bf_hi(m));
}

// sum over the lanes of a DPP row group: lane 31 (and 63) end up with the total of lanes 0..31 (32..63); W64: lane 63 with
// the total of the whole wave
#define C1_DPP_F64(v, CTRL, RMASK)                                                                                   \
  __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, RMASK, 0xf, false),                        \
                   __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, RMASK, 0xf, false))
#define C1_DPP_F64_Z(v, CTRL)                                                                                        \
  __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true),                           \
                   __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true))
template <bool W64>
__device__ __forceinline__ double row_total(double v) {
  v += C1_DPP_F64_Z(v, 0x111);       // row_shr:1 (lanes shifted in from outside the 16-lane row read 0)
  v += C1_DPP_F64_Z(v, 0x112);
  v += C1_DPP_F64_Z(v, 0x114);
  v += C1_DPP_F64_Z(v, 0x118);       // lane 15 of every row: the row's total
  v += C1_DPP_F64(v, 0x142, 0xa);    // row_bcast:15 into rows 1 and 3: lanes 31 / 63 = totals of 32 lanes
  if (W64) v += C1_DPP_F64(v, 0x143, 0xc);   // row_bcast:31 into rows 2, 3: lane 63 = the wave's total
  return v;
}

// TM: 32-row MFMA tiles per wave (wave tile 32 TM x 64); WM x WN = 8 waves; block tile BM = 32 TM WM, BN = 64 WN
template <int TM, int WM, int WN, int PRO, int EPI>
__global__ __launch_bounds__(CNT, 2) void c1gemm_kernel(C1P p) {
  static_assert(WM * WN == 8, "8 waves");
  constexpr int BM = 32 * TM * WM, BN = 64 * WN;
  constexpr int A_PLANE_B = BM * 64, B_ROW_B = BN * 2, B_PLANE_B = 32 * B_ROW_B;
  constexpr int A_STAGE_B = 3 * A_PLANE_B, STAGE_B = 3 * (A_PLANE_B + B_PLANE_B);
  constexpr int A_CH = BM * 4;                          // 16-byte chunks of one A plane per K-step
  constexpr int A_IT = A_CH >= CNT ? A_CH / CNT : 1;    // per thread (rows ar, ar + 128)
  constexpr bool A_ALL = A_CH >= CNT;                   // every thread loads A (else only t < A_CH)
  constexpr int QPR = BN / 4, RPP = CNT / QPR, B_IT = CBK / RPP;   // float4 per k-row, k-rows per pass, passes
  static_assert(BM * BN * 4 <= 2 * STAGE_B, "result tile must fit the operand LDS");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_B];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);   // one contiguous run of tiles per XCD
  const int m0 = (bid % p.mtiles) * BM;
  const int nt = bid / p.mtiles;
  const long n0 = (long)nt * BN;
  const int nsteps = p.K / CBK;

  // ---- loaders
  const int ac = t & 3, ar = t >> 2;
  const unsigned short* a_src[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) a_src[i] = p.A + (long)min(m0 + ar + 128 * i, p.M - 1) * p.K + ac * 8;
  const unsigned a_dst = off_row(ar, ac);
  const bool a_on = A_ALL || t < A_CH;
  // B: thread (kq = t / QPR, nq = t % QPR) loads the float4 at k = kq + RPP i, n = n0 + 4 nq of every K-step
  const int nq = t % QPR, kq = t / QPR;
  const float* b_src;
  {
    long n = n0 + 4 * nq;
    if (n >= p.N) n = p.N - 4;   // clamped columns are computed and never stored
    const long img = n / p.HW;
    b_src = p.B + img * (long)p.K * p.HW + (n - img * p.HW) + (long)kq * p.HW;
  }
  const long b_row = (long)RPP * p.HW;
  int pro_k = kq;
  u32x4 ra[3][A_IT];
  float4 rb[B_IT];
  float ps[B_IT], ph[B_IT];
  auto load_global = [&]() {
    if (a_on) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int i = 0; i < A_IT; ++i) ra[pl][i] = *reinterpret_cast<const u32x4*>(a_src[i] + pl * p.a_plane);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      rb[i] = *reinterpret_cast<const float4*>(b_src + i * b_row);
      if constexpr (PRO) {
        ps[i] = p.pro_scale[pro_k + RPP * i];
        ph[i] = p.pro_shift[pro_k + RPP * i];
      }
    }
  };
  auto advance = [&](bool more) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) a_src[i] += more ? CBK : 0;
    b_src += more ? (long)CBK * p.HW : 0;
    pro_k += more ? CBK : 0;
  };
  auto store_lds = [&](int buf) {
    unsigned char* s = smem + buf * STAGE_B;
    if (a_on) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<u32x4*>(s + pl * A_PLANE_B + a_dst + i * (128 * 64)) = ra[pl][i];
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      float4 v = rb[i];
      if constexpr (PRO) {   // bn_act_fwd_kernel<GELU>'s arithmetic (norm_act.hip)
        v.x = gelu_f(fmaf(v.x, ps[i], ph[i]));
        v.y = gelu_f(fmaf(v.y, ps[i], ph[i]));
        v.z = gelu_f(fmaf(v.z, ps[i], ph[i]));
        v.w = gelu_f(fmaf(v.w, ps[i], ph[i]));
      }
      unsigned h0, m0_, l0, h1, m1, l1;
      split_pair(v.x, v.y, h0, m0_, l0);
      split_pair(v.z, v.w, h1, m1, l1);
      const u32x2 h = {h0, h1}, m = {m0_, m1}, l = {l0, l1};
      const int k = kq + RPP * i;
      const unsigned off = (unsigned)(k * B_ROW_B) + ((((unsigned)nq >> 1) ^ sw_tr(k)) << 4) + 8u * (nq & 1);
      *reinterpret_cast<u32x2*>(s + A_STAGE_B + off) = h;
      *reinterpret_cast<u32x2*>(s + A_STAGE_B + B_PLANE_B + off) = m;
      *reinterpret_cast<u32x2*>(s + A_STAGE_B + 2 * B_PLANE_B + off) = l;
    }
  };

  // ---- fragments
  const int wm0 = (wave / WN) * (32 * TM), wn0 = (wave % WN) * 64;
  const int r31 = lane & 31, lh = lane >> 5;
  const unsigned a_rd = (unsigned)((wm0 + r31) * 64);
  const int a_x = (r31 >> 2) & 3;
  const int i16 = lane & 15, g = lane >> 4, tq = i16 >> 2, tp = i16 & 3;

  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  struct Frag {
    bf16x8 a[TM][3], b[2][3];
  };
  auto read_frag = [&](Frag& f, int buf, int ks) {
    const unsigned char* s = smem + buf * STAGE_B;
    const unsigned a_c = (unsigned)(((2 * ks + lh) ^ a_x) << 4);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        f.a[i][pl] = *reinterpret_cast<const bf16x8*>(s + pl * A_PLANE_B + a_rd + i * (32 * 64) + a_c);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        s16x4 part[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int row = 16 * ks + 8 * (g >> 1) + 4 * hf + tq;
          const unsigned ch = (unsigned)(((wn0 + 32 * j) >> 3) + 2 * (g & 1) + (tp >> 1));
          const unsigned off = (unsigned)(row * B_ROW_B) + ((ch ^ sw_tr(row)) << 4) + 8u * (tp & 1);
          part[hf] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(s + A_STAGE_B + pl * B_PLANE_B + off));
        }
        const s16x8 v = __builtin_shufflevector(part[0], part[1], 0, 1, 2, 3, 4, 5, 6, 7);
        f.b[j][pl] = __builtin_bit_cast(bf16x8, v);
      }
  };
  auto mfma_frag = [&](const Frag& f) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = acc[i][j];   // smallest terms first
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][2], f.b[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][1], f.b[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][1], f.b[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][0], c, 0, 0, 0);
        acc[i][j] = c;
      }
  };

  // Software pipeline of splitgemm.hip: k-slab 0 MFMAs || (k-slab 1 fragment reads, split + LDS stores of K-step st + 1,
  // global loads of K-step st + 2); barrier; k-slab 1 MFMAs || k-slab 0 fragment reads of the next stage.
  constexpr int NMF = TM * 12;                                           // MFMAs per k-slab
  constexpr int VALU1 = B_IT * (22 + (PRO ? 80 : 0)) + 16;               // vector ALU work of store_lds per K-step
  constexpr int DSW1 = 3 * A_IT + 3 * B_IT, VM1 = 3 * A_IT + B_IT * (PRO ? 3 : 1), DSR1 = 3 * TM + 12;
  {
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
#pragma unroll
      for (int q = 0; q < NMF; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, (DSR1 + NMF - 1) / NMF, 0);     // DS read
        __builtin_amdgcn_sched_group_barrier(0x002, (VALU1 + NMF - 1) / NMF, 0);    // VALU
        __builtin_amdgcn_sched_group_barrier(0x200, (DSW1 + NMF - 1) / NMF, 0);     // DS write
        __builtin_amdgcn_sched_group_barrier(0x020, (VM1 + NMF - 1) / NMF, 0);      // VMEM read
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      read_frag(f0, cur ^ 1, 0);
      mfma_frag(f1);
#pragma unroll
      for (int q = 0; q < NMF; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (DSR1 + NMF - 1) / NMF, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: the result tile goes through LDS (fp32 [BM][BN]) and leaves as dwordx4 rows.
  // accumulator register q of lane (r31, lh) is C[(q & 3) + 8 (q >> 2) + 4 lh][r31] of its 32 x 32 tile
  __syncthreads();   // every wave has read its last fragments: the operand stages are free
  float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = wm0 + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * lh;
        tile[row * BN + wn0 + 32 * j + r31] = acc[i][j][q];
      }
  __syncthreads();
  constexpr int RPO = CNT / QPR;            // rows per pass of the store phase (16 for BN = 128, 8 for BN = 256)
  const int er = t / QPR;                   // this thread's row within a pass; its column quad is nq (as in the loader)
  const long nn = n0 + 4 * nq;
  const bool col_ok = nn < p.N;
  long c_base = 0;
  {
    const long n = col_ok ? nn : 0;
    const long img = n / p.HW;
    c_base = img * (long)p.M * p.HW + (n - img * p.HW);
  }
  const bool red_lane = (BN == 128) ? ((lane & 31) == 31) : (lane == 63);
#pragma unroll 4
  for (int r = er; r < BM; r += RPO) {
    const int m = m0 + r;
    const bool ok = col_ok && m < p.M;
    const int mc = m < p.M ? m : p.M - 1;
    const long off = c_base + (long)mc * p.HW;
    float4 v = *reinterpret_cast<const float4*>(tile + r * BN + 4 * nq);
    if constexpr (EPI == EPI_PLAIN) {
      if (p.res) {
        const float4 rv = *reinterpret_cast<const float4*>(p.res + (ok ? off : 0));
        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
      }
      if (ok) *reinterpret_cast<float4*>(p.C + off) = v;
      if (p.part0) {   // BatchNorm sums of the stored values: fp32 sums of four, fp64 across the row (chan_reduce_kernel's arithmetic)
        const float z = ok ? 1.f : 0.f;
        double s1 = (double)(z * ((v.x + v.y) + (v.z + v.w)));
        double s2 = (double)(z * (fmaf(v.x, v.x, v.y * v.y) + fmaf(v.z, v.z, v.w * v.w)));
        s1 = row_total<BN == 256>(s1);
        s2 = row_total<BN == 256>(s2);
        if (red_lane && m < p.M) {
          const long slot = (long)nt * p.M + m;
          p.part0[slot] = s1;
          p.part1[slot] = s2;
        }
      }
    } else {
      const float4 xv = *reinterpret_cast<const float4*>(p.res + (ok ? off : 0));
      const float a = p.e_scale[mc], b = p.e_shift[mc], mu = p.e_mean[mc], is = p.e_invstd[mc];
      const float d0 = v.x * gelu_grad_f(fmaf(xv.x, a, b)), d1 = v.y * gelu_grad_f(fmaf(xv.y, a, b));
      const float d2 = v.z * gelu_grad_f(fmaf(xv.z, a, b)), d3 = v.w * gelu_grad_f(fmaf(xv.w, a, b));
      const float h0 = (xv.x - mu) * is, h1 = (xv.y - mu) * is, h2 = (xv.z - mu) * is, h3 = (xv.w - mu) * is;
      if constexpr (EPI == EPI_BNRED) {
        if (p.store && ok) *reinterpret_cast<float4*>(p.C + off) = v;
        const float z = ok ? 1.f : 0.f;   // bn_act_bwd_reduce_kernel's quad()
        double s1 = (double)(z * ((d0 + d1) + (d2 + d3)));
        double s2 = (double)(z * (fmaf(d0, h0, d1 * h1) + fmaf(d2, h2, d3 * h3)));
        s1 = row_total<BN == 256>(s1);
        s2 = row_total<BN == 256>(s2);
        if (red_lane && m < p.M) {
          const long slot = (long)nt * p.M + m;
          p.part0[slot] = s1;
          p.part1[slot] = s2;
        }
      } else {   // bn_act_bwd_dx_kernel's one()
        const float gi = p.e_gamma[mc] * is;
        const float k1 = p.e_coef[2 * mc] * p.inv_count, k2 = p.e_coef[2 * mc + 1] * p.inv_count;
        float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.res2) rv = *reinterpret_cast<const float4*>(p.res2 + (ok ? off : 0));
        float4 o;
        o.x = gi * (d0 - k1 - h0 * k2) + rv.x;
        o.y = gi * (d1 - k1 - h1 * k2) + rv.y;
        o.z = gi * (d2 - k1 - h2 * k2) + rv.z;
        o.w = gi * (d3 - k1 - h3 * k2) + rv.w;
        if (ok) *reinterpret_cast<float4*>(p.C + off) = o;
      }
    }
  }
}

// w [Cout][Cin] -> W3 [3][Cout][Cin] and Wt3 [3][Cin][Cout] (planes h, m, l of w and of w^T)
__global__ __launch_bounds__(256) void c1_split_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ W3,
                                                               unsigned short* __restrict__ Wt3, int Cout, int Cin) {
  const long n = (long)Cout * Cin;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned short h, m, l;
  split3(w[i], h, m, l);
  W3[i] = h;
  W3[n + i] = m;
  W3[2 * n + i] = l;
  const long co = i / Cin, ci = i - co * Cin;
  const long j = ci * Cout + co;
  Wt3[j] = h;
  Wt3[n + j] = m;
  Wt3[2 * n + j] = l;
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// tile choice: 0 = 256 x 128, 1 = 128 x 256, 2 = 64 x 256; -1 = not served
inline int pick_tile(int M, int K, int HW) {
  if (M < 64 || M % 64 != 0 || K % CBK != 0 || K < CBK || HW % 4 != 0) return -1;
  if (M % 256 == 0) return 0;
  if (M % 128 == 0) return 1;
  return 2;
}

template <int PRO, int EPI>
int launch_c1(C1P& p, int tile, hipStream_t st, const char* what) {
  const int bm = tile == 0 ? 256 : (tile == 1 ? 128 : 64), bn = tile == 0 ? 128 : 256;
  p.mtiles = cdiv(p.M, bm);
  const long tiles = (long)p.mtiles * cdiv(p.N, bn);
  WFAE_REQUIRE(tiles < (1l << 31), WFAE_ERR_BAD_SHAPE, "%s: grid too large", what);
  const dim3 grid((unsigned)tiles), block(CNT);
  if (tile == 0) hipLaunchKernelGGL((c1gemm_kernel<2, 4, 2, PRO, EPI>), grid, block, 0, st, p);
  else if (tile == 1) hipLaunchKernelGGL((c1gemm_kernel<2, 2, 4, PRO, EPI>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((c1gemm_kernel<1, 2, 4, PRO, EPI>), grid, block, 0, st, p);
  return check_launch(what);
}

int c1_common(C1P& p, const uint16_t* W3, const float* x, float* y, int NB, int K, int M, int HW, const char* what) {
  WFAE_REQUIRE(W3 && x, WFAE_ERR_NULL_POINTER, "%s: null pointer", what);
  WFAE_REQUIRE(NB > 0 && K > 0 && M > 0 && HW > 0 && (int64_t)NB * HW < (1ll << 31), WFAE_ERR_BAD_SHAPE, "%s: bad shape", what);
  WFAE_REQUIRE(pick_tile(M, K, HW) >= 0, WFAE_ERR_UNSUPPORTED,
               "%s: needs M %% 64 == 0, K %% 32 == 0, HW %% 4 == 0 (M %d, K %d, HW %d)", what, M, K, HW);
  WFAE_REQUIRE(al16(W3) && al16(x) && al16(y), WFAE_ERR_UNSUPPORTED, "%s: tensors must be 16-byte aligned", what);
  p.A = W3; p.a_plane = (long)M * K; p.B = x; p.C = y;
  p.M = M; p.K = K; p.HW = HW; p.N = (long)NB * HW;
  return WFAE_OK;
}

}  // namespace

extern "C" {

int wfae_c1gemm_supported(int M, int K, int HW) { return pick_tile(M, K, HW) >= 0 ? 1 : 0; }

int wfae_c1gemm_stat_rows(int M, int K, int NB, int HW) {
  const int tile = pick_tile(M, K, HW);
  if (tile < 0) return 0;
  return cdiv((int64_t)NB * HW, tile == 0 ? 128 : 256);   // one partial row per column tile
}

int wfae_c1gemm_split_weights(const float* w, uint16_t* W3, uint16_t* Wt3, int Cout, int Cin, wfae_stream_t stream) {
  WFAE_REQUIRE(w && W3 && Wt3, WFAE_ERR_NULL_POINTER, "c1gemm_split_weights: null pointer");
  WFAE_REQUIRE(Cout > 0 && Cin > 0, WFAE_ERR_BAD_SHAPE, "c1gemm_split_weights: bad shape");
  const long n = (long)Cout * Cin;
  hipLaunchKernelGGL(c1_split_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, W3, Wt3,
                     Cout, Cin);
  return check_launch("c1gemm_split_weights");
}

int wfae_c1gemm_fwd(const uint16_t* W3, const float* x, const float* pro_scale, const float* pro_shift, const float* res,
                    float* y, int NB, int K, int M, int HW, double* stat_part, int64_t stat_capacity, int* stat_rows,
                    wfae_stream_t stream) {
  C1P p = {};
  int rc = c1_common(p, W3, x, y, NB, K, M, HW, "c1gemm_fwd");
  if (rc) return rc;
  WFAE_REQUIRE(y, WFAE_ERR_NULL_POINTER, "c1gemm_fwd: null pointer");
  WFAE_REQUIRE((pro_scale != nullptr) == (pro_shift != nullptr) && (stat_part != nullptr) == (stat_rows != nullptr),
               WFAE_ERR_NULL_POINTER, "c1gemm_fwd: scale / shift and stat_part / stat_rows go together");
  WFAE_REQUIRE(!res || al16(res), WFAE_ERR_UNSUPPORTED, "c1gemm_fwd: residual must be 16-byte aligned");
  p.res = res;
  p.pro_scale = pro_scale; p.pro_shift = pro_shift;
  const int tile = pick_tile(M, K, HW);
  if (stat_part) {
    const int rows = wfae_c1gemm_stat_rows(M, K, NB, HW);
    WFAE_REQUIRE(stat_capacity >= 2 * (int64_t)rows * M, WFAE_ERR_WORKSPACE, "c1gemm_fwd: stat_part holds %lld doubles, needs %lld",
                 (long long)stat_capacity, (long long)(2 * (int64_t)rows * M));
    *stat_rows = rows;
    p.part0 = stat_part;
    p.part1 = stat_part + (long)rows * M;
  }
  return pro_scale ? launch_c1<1, EPI_PLAIN>(p, tile, (hipStream_t)stream, "c1gemm_fwd")
                   : launch_c1<0, EPI_PLAIN>(p, tile, (hipStream_t)stream, "c1gemm_fwd");
}

int wfae_c1gemm_bnred(const uint16_t* W3, const float* dt, const float* x, const float* bn_scale, const float* bn_shift,
                      const float* save_mean, const float* save_invstd, float* da, int NB, int K, int M, int HW,
                      double* part, int64_t part_capacity, int* part_rows, wfae_stream_t stream) {
  C1P p = {};
  int rc = c1_common(p, W3, dt, da, NB, K, M, HW, "c1gemm_bnred");
  if (rc) return rc;
  WFAE_REQUIRE(x && bn_scale && bn_shift && save_mean && save_invstd && part && part_rows, WFAE_ERR_NULL_POINTER,
               "c1gemm_bnred: null pointer");
  WFAE_REQUIRE(al16(x), WFAE_ERR_UNSUPPORTED, "c1gemm_bnred: x must be 16-byte aligned");
  const int rows = wfae_c1gemm_stat_rows(M, K, NB, HW);
  WFAE_REQUIRE(part_capacity >= 2 * (int64_t)rows * M, WFAE_ERR_WORKSPACE, "c1gemm_bnred: part holds %lld doubles, needs %lld",
               (long long)part_capacity, (long long)(2 * (int64_t)rows * M));
  *part_rows = rows;
  p.res = x;
  p.e_scale = bn_scale; p.e_shift = bn_shift; p.e_mean = save_mean; p.e_invstd = save_invstd;
  p.part0 = part;
  p.part1 = part + (long)rows * M;
  p.store = da != nullptr;
  return launch_c1<0, EPI_BNRED>(p, pick_tile(M, K, HW), (hipStream_t)stream, "c1gemm_bnred");
}

int wfae_c1gemm_bndx(const uint16_t* W3, const float* dt, const float* x, const float* gamma, const float* bn_scale,
                     const float* bn_shift, const float* save_mean, const float* save_invstd, const float* coef,
                     const float* res, float* dx, int NB, int K, int M, int HW, int training, wfae_stream_t stream) {
  C1P p = {};
  int rc = c1_common(p, W3, dt, dx, NB, K, M, HW, "c1gemm_bndx");
  if (rc) return rc;
  WFAE_REQUIRE(x && gamma && bn_scale && bn_shift && save_mean && save_invstd && coef && dx, WFAE_ERR_NULL_POINTER,
               "c1gemm_bndx: null pointer");
  WFAE_REQUIRE(al16(x) && (!res || al16(res)), WFAE_ERR_UNSUPPORTED, "c1gemm_bndx: tensors must be 16-byte aligned");
  p.res = x; p.res2 = res;
  p.e_scale = bn_scale; p.e_shift = bn_shift; p.e_mean = save_mean; p.e_invstd = save_invstd;
  p.e_gamma = gamma; p.e_coef = coef;
  p.inv_count = training ? (float)(1.0 / ((double)NB * HW)) : 0.f;
  return launch_c1<0, EPI_BNDX>(p, pick_tile(M, K, HW), (hipStream_t)stream, "c1gemm_bndx");
}

}  // extern "C"
