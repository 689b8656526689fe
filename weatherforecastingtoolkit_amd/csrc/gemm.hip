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
#include <stdlib.h>
#include <type_traits>

using namespace wfae;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

#ifndef WFAE_DEFAULT_MFMA
#define WFAE_DEFAULT_MFMA 32
#endif

constexpr int BK = 16;
constexpr int BN = 128;
constexpr int NT = 256;

// k-contiguous operands: thread idx loads the float4 (row kc_row(idx), k = kc_k(idx) .. +3) and scatters it into
// the transposed LDS image As[k][m] with four ds_write_b32.  The LDS has 32 banks and a 64-lane ds_write_b32
// is served 32 lanes per cycle; with row strides == 4 (mod 32) words the bank of a write is 4 (kq + j) + m.  The
// natural map (m = idx >> 2, kq = 4 (idx & 3)) puts kq = 0 and kq = 8 of one half-wave on the same banks
// (PMC: SQ_LDS_BANK_CONFLICT = 21-30 % of the LDS-active cycles of these kernels, 0 for the m-contiguous
// operand kinds); this map gives a half-wave 16 rows x 2 k-quads = 32 distinct banks.  The global side still
// reads whole 64-byte row segments per wave-instruction.
__device__ __forceinline__ int kc_row(int idx) { return (idx & 15) | ((idx >> 6) << 4); }
__device__ __forceinline__ int kc_k(int idx) { return ((idx >> 4) & 3) * 4; }

enum AKind { A_KCONTIG = 0, A_MCONTIG = 1 };
enum BKind { B_NCONTIG = 0, B_KCONTIG = 1, B_DOWN = 2, B_UP = 3, B_WGRAD = 4, B_WGRAD3 = 5, B_TAPN = 6, B_TAPK = 7 };
enum EKind { E_BATCHED = 0, E_SLAB = 1, E_UP = 2 };

struct GemmP {
  // A / B / C / res are typed by the kernel's element-type parameters AT / BT / CT (float, or bf16_t for activation tensors
  // in the bf16-storage mode): offsets and strides below are always in ELEMENTS
  const void* A;
  const void* B;
  void* C;
  const float* bias;  // [M] or null
  const void* res;    // residual (element type CT), E_BATCHED only
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
  // batch over blockIdx.y (plain operand kinds): element offsets added per y (Winograd: y = transform position)
  long a_y, b_y, c_y;
  int big_ok;        // 256-row tiles allowed for this launch (long-K plain GEMMs)
  int a_vec, b_vec, c_vec;  // 16-byte vector accesses are legal for this operand / the result
  // BatchNorm statistics of the result, fused into the E_BATCHED vector epilogue (kernels with two wave columns):
  // (fp64: the 16-lane DPP reduction runs on doubles, so the partial sums are exact to fp64 rounding like the separate
  // statistics pass) stat_sum / stat_sq [2 * ntiles][M] receive, per output row m (= channel) and 64-column wave tile, the sum and the
  // sum of squares of the values stored by that wave; null = off
  double* stat_sum;
  double* stat_sq;
  // BatchNorm-apply + GELU prologue on the B operand (PRO kernels): B'(k,n) = gelu(B(k,n) * b_scale[c] + b_shift[c]) with
  // c the CHANNEL index of the element — k for B_NCONTIG (1x1 forward: B = x, k = input channel), n for B_KCONTIG
  // (1x1 weight gradient: B = x, n = input channel).  Applied between the global load and the LDS store, so the
  // activated tensor a = gelu(bn(x)) of the reference's BN -> GELU -> Conv1x1 chain (ae_64x8x8_lin.py:14-15) is never
  // written to HBM; the arithmetic is the one of bn_act_fwd_kernel (norm_act.hip), bit for bit.
  const float* b_scale;
  const float* b_shift;
  // 4x4 s2 geometry (gather kinds): lo side Hlo x Wlo, hi side 2Hlo x 2Wlo
  int Chi, Clo, Hlo, Wlo;
  // grouped 3x3 weight gradient (B_WGRAD3): blockIdx.y = group, Chi = total channels,
  // Hlo x Wlo = the (stride-1) image size, cpg = channels per group
  int cpg;
  // flat-shift 4x4 stride-1 convolution (B_TAPN / B_TAPK): the B operand is a zero-padded activation
  // stored as planes of stride b_ld = P floats with rows of Wlo = Wp floats; an output position is the
  // flat index q = oy*Wp + ox, and tap (ky,kx) of channel c reads plane c at q + ky*Wp + kx — a constant
  // shift per tap, so the convolution is a GEMM over k = tap*Chi + c (Chi = channels, Chi % 16 == 0)
  // whose B rows are shifted, unaligned views of the same planes.  Columns ox >= Wo are computed and
  // dropped by the caller.  B_TAPN: B(k,n=q) (forward / data gradient); B_TAPK: B(k=(img,q), n=tap*Chi+c)
  // (weight gradient).
};

// inclusive prefix sum over the 16 lanes of a DPP row (row_shr:1,2,4,8 with bound_ctrl: lanes shifted in from
// outside the row read 0): lane 15 of every row ends up with the row total — VALU adds, no LDS permutes
__device__ __forceinline__ float dpp_row_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
  return v;
}

// 16-byte global access that is only 4-byte aligned (gfx9 global memory runs in unaligned access mode)
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef vf4 __attribute__((aligned(4))) vf4u;
__device__ __forceinline__ float4 load4u(const float* ptr) {
  const vf4 v = *reinterpret_cast<const vf4u*>(ptr);
  return make_float4(v.x, v.y, v.z, v.w);
}

// gelu(v * s + h) on four elements: exactly bn_act_fwd_kernel<GELU>'s arithmetic (norm_act.hip)
__device__ __forceinline__ float4 bn_gelu4(float4 v, float s, float h) {
  return make_float4(gelu_f(fmaf(v.x, s, h)), gelu_f(fmaf(v.y, s, h)), gelu_f(fmaf(v.z, s, h)), gelu_f(fmaf(v.w, s, h)));
}

// the same inclusive prefix sum on doubles (two DPP moves per step); lane 15 of every row ends up with the row total
#define WFAE_DPP_F64(v, CTRL)                                                                                    \
  __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true),                       \
                   __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true))
__device__ __forceinline__ double dpp_row_sum(double v) {
  v += WFAE_DPP_F64(v, 0x111);
  v += WFAE_DPP_F64(v, 0x112);
  v += WFAE_DPP_F64(v, 0x114);
  v += WFAE_DPP_F64(v, 0x118);
  return v;
}

// VEC: every operand/result row is 16-byte aligned and a multiple of 4 floats long, so all
// global traffic is dwordx4.  Loaders are BRANCH-FREE: out-of-range elements load from a clamped
// (always valid) address and are zeroed by a select, so the compiler issues every global load of
// a stage back-to-back and waits once, after the MFMAs of the current stage.
// second launch-bound argument = minimum waves per SIMD: 2 for the 256-row tile (128 accumulator registers),
// 4 otherwise (3 for the down/up gathers, which spill at 128) — the unified VGPR/AGPR file has 512 entries per lane and SIMD
// PREC: WFAE_PRECISION_FP32 — v_mfma_f32_32x32x2_f32; WFAE_PRECISION_BF16 — same loaders, LDS images and epilogue,
// but the operand fragments are rounded to bf16 (v_cvt_pk_bf16_f32, RNE) after the LDS read and one
// v_mfma_f32_32x32x16_bf16 consumes a whole 16-deep stage (fp32 accumulation).
// AT / BT / CT: element types of the A / B operands and of the result (+ residual): float, or bf16_t for ACTIVATION tensors in
// the bf16-storage mode (plain operand kinds only; four elements per access = 8-byte loads / stores, widened to fp32
// before the LDS images, which stay fp32; results rounded to nearest even)
template <int BM, int WMW, int WNW, int AK, int BKD, int EK, bool VEC, int MF, int PREC = 0, int PRO = 0, typename AT = float,
          typename BT = float, typename CT = float>
__global__ __launch_bounds__(NT, (BM >= 256 ? 2 : ((BKD == B_DOWN || BKD == B_UP || PREC != 0) ? 3 : 4))) void gemm_kernel(GemmP p) {
  static_assert(WMW * WNW == 4, "4 waves");
  static_assert((std::is_same<AT, float>::value && std::is_same<BT, float>::value && std::is_same<CT, float>::value) ||
                    ((BKD == B_NCONTIG || BKD == B_KCONTIG) && EK != E_UP),
                "bf16 storage: plain operand kinds");
  static_assert(EK == E_BATCHED || std::is_same<CT, float>::value, "split-K slabs are fp32");
  // PRO 1: prologue on the B operand; PRO 2: on a K-contiguous A operand (the weight gradient with swapped roles)
  static_assert(PRO == 0 || (VEC && (BKD == B_NCONTIG || BKD == B_KCONTIG)), "prologue: vector kernels, plain B kinds");
  static_assert(PRO != 2 || AK == A_KCONTIG, "A-side prologue: K-contiguous A");
  static_assert(MF == 32, "v_mfma_f32_32x32x2_f32 (the 16x16x4 form measured the same rate and was dropped)");
  constexpr int WROWS = BM / WMW, WCOLS = BN / WNW;  // wave tile
  constexpr int TM = WROWS / MF;
  constexpr int TN = WCOLS / MF;
  static_assert(TM >= 1 && TN >= 1, "tile");
  // LDS row strides: operand reads of one 32-lane group touch 2 (MF=32) / 2x2 (MF=16) k-rows;
  // +4 (resp. +16) words shift consecutive rows to disjoint banks
  constexpr int LDA_S = BM + (MF == 32 ? 4 : 16);
  constexpr int LDB_S = BN + (MF == 32 ? 4 : 16);
  using acc_t = typename std::conditional<MF == 32, f32x16, f32x4>::type;
  constexpr int NREG = MF == 32 ? 16 : 4;
  constexpr int A_FLOATS = 2 * BK * LDA_S, B_FLOATS = 2 * BK * LDB_S;
  __shared__ __attribute__((aligned(16))) float smem[A_FLOATS + B_FLOATS];
  float(*As)[BK][LDA_S] = reinterpret_cast<float(*)[BK][LDA_S]>(smem);
  float(*Bs)[BK][LDB_S] = reinterpret_cast<float(*)[BK][LDB_S]>(smem + A_FLOATS);

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int mtiles = (p.M + BM - 1) / BM;
  const int bid = blockIdx.x;
  const int m0 = (bid % mtiles) * BM;
  const int n0 = (bid / mtiles) * BN;
  const int z = blockIdx.z;

  int k_begin = 0, k_end = p.K;
  int py = 0, px = 0;
  const AT* __restrict__ Ap = reinterpret_cast<const AT*>(p.A);
  if constexpr (BKD == B_UP) {
    py = z >> 1;
    px = z & 1;
    Ap += (long)z * p.K * p.M;  // packed per-phase weights [phase][k][m]
  } else {
    k_begin = z * p.k_per_split;
    k_end = min(p.K, k_begin + p.k_per_split);
  }
  const BT* __restrict__ Bp = reinterpret_cast<const BT*>(p.B);
  if constexpr (BKD == B_WGRAD3) Ap += (long)blockIdx.y * p.cpg * p.a_hw;  // dY channels of this group
  if constexpr (BKD == B_NCONTIG || BKD == B_KCONTIG) {
    Ap += (long)blockIdx.y * p.a_y;
    Bp += (long)blockIdx.y * p.b_y;
  }

  // ------------------------------------------------ per-thread loader state
  constexpr int A_CNT = BM * BK / 4;
  constexpr int A_IT = (A_CNT + NT - 1) / NT;
  constexpr int B_IT = (BN * BK / 4) / NT;  // 2
  float4 ra[A_IT];
  float4 rb[B_IT];
  float rg[8];  // gather kinds
  float pa_s[A_IT], pa_h[A_IT];  // PRO 2: the same for the rows (channels) of a K-contiguous A operand
  float pb_s[B_IT], pb_h[B_IT];  // PRO: folded BatchNorm scale / shift of the channel each staged B row belongs to
  int pb_k = 0;                  // PRO, B_NCONTIG: channel of this thread's first row in the NEXT lean stage

  const int H = 2 * p.Hlo, W = 2 * p.Wlo;
  const int HWlo = p.Hlo * p.Wlo;

  long bn_base = 0;  // B_NCONTIG: n is fixed per thread
  bool bn_ok = false;
  const int g_nl = t & (BN - 1);  // gather kinds
  const int g_kh = t >> 7;
  long g_base = 0;
  unsigned g_rmask = 0, g_cmask = 0;
  bool g_ok = false;
  int g_hi = 0, g_ky = 0, g_kx = 0;

  if constexpr (BKD == B_NCONTIG || BKD == B_TAPN) {
    const int n = n0 + (t & 31) * 4;
    bn_ok = n < p.N;
    const int nn = bn_ok ? n : 0;
    const int img = nn / p.b_hw;
    bn_base = (long)img * p.b_img + (nn - img * p.b_hw);
  } else if constexpr (BKD == B_DOWN) {
    const int n = n0 + g_nl;
    g_ok = n < p.N;
    const int nn = g_ok ? n : 0;
    const int img = nn / HWlo;
    const int r = nn - img * HWlo;
    const int oy = r / p.Wlo, ox = r - oy * p.Wlo;
    const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
    g_base = (long)img * p.Chi * H * W + (long)iy0 * W + ix0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int iy = iy0 + g_kh * 2 + i;
      if (g_ok && iy >= 0 && iy < H) g_rmask |= 1u << i;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ix = ix0 + i;
      if (ix >= 0 && ix < W) g_cmask |= 1u << i;
    }
  } else if constexpr (BKD == B_UP) {
    const int n = n0 + g_nl;
    g_ok = n < p.N;
    const int nn = g_ok ? n : 0;
    const int img = nn / HWlo;
    const int r = nn - img * HWlo;
    const int a = r / p.Wlo, b = r - a * p.Wlo;
    g_base = (long)img * p.Clo * HWlo + (long)(a + py) * p.Wlo + (b + px);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = a + py - i, col = b + px - i;
      if (g_ok && row >= 0 && row < p.Hlo) g_rmask |= 1u << i;
      if (col >= 0 && col < p.Wlo) g_cmask |= 1u << i;
    }
  } else if constexpr (BKD == B_WGRAD) {
    const int n = n0 + g_nl;
    g_ok = n < p.N;
    const int nn = g_ok ? n : 0;
    g_hi = nn >> 4;
    g_ky = (nn >> 2) & 3;
    g_kx = nn & 3;
  } else if constexpr (BKD == B_WGRAD3) {
    const int n = n0 + g_nl;  // n = ci*9 + ky*3 + kx
    g_ok = n < p.N;
    const int nn = g_ok ? n : 0;
    const int ci = nn / 9, tap = nn - ci * 9;
    g_hi = blockIdx.y * p.cpg + ci;
    g_ky = tap / 3;
    g_kx = tap - g_ky * 3;
  }

  // Out-of-range elements are loaded from a clamped (valid) address; the select that zeroes them is
  // applied only when the registers are written to LDS (store_a / store_b), AFTER the MFMAs of the
  // current stage, so the wave never waits for its prefetch before it starts multiplying.
  unsigned a_okbits = 0, b_okbits = 0, g_okbits = 0;
  auto ld4 = [&](auto base, long off, bool ok) -> float4 {  // immediate select (scalar fallback paths)
    float4 v = wfae::ld4(base + (ok ? off : 0));
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
  };
  auto ld4raw = [&](auto base, long off, bool ok) -> float4 { return wfae::ld4(base + (ok ? off : 0)); };
  auto ld4raw_u = [&](const float* base, long off, bool ok) -> float4 {
    return load4u(base + (ok ? off : 0));
  };
  auto tap_off = [&](int tap) -> int { return (tap >> 2) * p.Wlo + (tap & 3); };
  (void)ld4raw_u;
  (void)tap_off;
  auto ld1 = [&](auto base, long off, bool ok) -> float {
    const float v = wfae::ld1(base + (ok ? off : 0));
    return ok ? v : 0.f;
  };
  auto ld1raw = [&](auto base, long off, bool ok) -> float { return wfae::ld1(base + (ok ? off : 0)); };
  auto sel4 = [&](float4 v, unsigned bits, int i) -> float4 {
    return ((bits >> i) & 1u) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  (void)ld4;


  // ---------------------------------------------------------------------------------------------
  // Lean loaders (VEC kernels): the MFMA pipe shares the SIMD's issue port with VALU work, and
  // PMC shows MFMA utilisation falling from 75 % to 61 % as VALU instructions per MFMA rise from
  // 5 to 9 — so all address arithmetic is incremental (pointer bumps / pixel cursors), masks of
  // the gather operands are loop-invariant, out-of-range taps load a per-thread always-valid
  // address and are zeroed when the registers are written to LDS.
  // ---------------------------------------------------------------------------------------------
  long la_off[A_IT];
  int la_k[A_IT], la_pk[A_IT];
  bool la_st[A_IT];
  long lb_off[B_IT];
  int lb_k[B_IT], lb_pk[B_IT];
  bool lb_st[B_IT];
  long lg_off = 0;        // gather: element offset of the thread's window for the current stage
  int lg_rel[8];          // gather: loop-invariant relative offsets (clamped to a valid tap)
  unsigned lg_static = 0; // gather: loop-invariant validity bits
  int lg_k = 0, c_x = 0, c_iy = 0, c_img = 0;  // wgrad pixel cursor
  int lt_tap = 0, lt_c = 0;                     // B_TAPN: tap / first channel of the current stage (wave-uniform)
  if constexpr (VEC) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int idx = t + i * NT;
      const bool inb = (A_IT * NT == A_CNT) || idx < A_CNT;
      if constexpr (AK == A_KCONTIG) {
        const int m = m0 + kc_row(idx);
        const int k = k_begin + kc_k(idx);
        la_st[i] = inb && m < p.M;
        if constexpr (PRO == 2) {
          pa_s[i] = p.b_scale[la_st[i] ? m : 0];
          pa_h[i] = p.b_shift[la_st[i] ? m : 0];
        }
        const int img = k / p.a_hw;
        la_pk[i] = k - img * p.a_hw;
        la_off[i] = (long)img * p.a_img + (long)(la_st[i] ? m : 0) * p.a_ld + la_pk[i];
        la_k[i] = k;
      } else {
        const int kr = idx / (BM / 4);
        const int m = m0 + (idx % (BM / 4)) * 4;
        la_st[i] = inb && m < p.M;
        la_k[i] = k_begin + kr;
        la_pk[i] = 0;
        la_off[i] = (long)la_k[i] * p.a_ld + (la_st[i] ? m : 0);
      }
    }
    if constexpr (BKD == B_NCONTIG) {
      pb_k = k_begin + (t >> 5);
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        lb_k[i] = k_begin + (t >> 5) + i * 8;
        lb_st[i] = bn_ok;
        lb_pk[i] = 0;
        lb_off[i] = bn_base + (long)lb_k[i] * p.b_ld;
      }
    } else if constexpr (BKD == B_KCONTIG) {
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int idx = t + i * NT;
        const int n = n0 + kc_row(idx);
        const int k = k_begin + kc_k(idx);
        lb_st[i] = n < p.N;
        if constexpr (PRO == 1) {  // the channel of a K-contiguous row never changes
          pb_s[i] = p.b_scale[lb_st[i] ? n : 0];
          pb_h[i] = p.b_shift[lb_st[i] ? n : 0];
        }
        const int img = k / p.b_hw;
        lb_pk[i] = k - img * p.b_hw;
        lb_off[i] = (long)img * p.b_img + (long)(lb_st[i] ? n : 0) * p.b_ld + lb_pk[i];
        lb_k[i] = k;
      }
    } else if constexpr (BKD == B_TAPN) {
      lt_tap = k_begin / p.Chi;
      lt_c = k_begin - lt_tap * p.Chi;
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        lb_k[i] = k_begin + (t >> 5) + i * 8;
        lb_st[i] = bn_ok;
        lb_pk[i] = 0;
        lb_off[i] = bn_base + (long)(lt_c + (t >> 5) + i * 8) * p.b_ld + tap_off(lt_tap);
      }
    } else if constexpr (BKD == B_TAPK) {
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int idx = t + i * NT;
        const int n = n0 + kc_row(idx);
        const int k = k_begin + kc_k(idx);
        lb_st[i] = n < p.N;
        const int nn = lb_st[i] ? n : 0;
        const int tap = nn / p.Chi, c = nn - tap * p.Chi;
        const int img = k / p.b_hw;
        lb_pk[i] = k - img * p.b_hw;
        lb_off[i] = (long)img * p.b_img + (long)c * p.b_ld + tap_off(tap) + lb_pk[i];
        lb_k[i] = k;
      }
    } else if constexpr (BKD == B_DOWN) {
      lg_off = g_base + (long)(k_begin >> 4) * H * W + (long)(g_kh * 2) * W;
      const int safe = g_kh == 0 ? W + 1 : 1;  // pixel (2oy, 2ox) / (2oy+1, 2ox) is always inside the image
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = ((g_rmask >> i) & 1u) && ((g_cmask >> j) & 1u);
          lg_static |= (unsigned)ok << (i * 4 + j);
          lg_rel[i * 4 + j] = ok ? i * W + j : safe;
        }
    } else if constexpr (BKD == B_UP) {
      lg_off = g_base + (long)((k_begin >> 2) + g_kh * 2) * HWlo;
      const int safe = -py * p.Wlo - px;  // tap (ty, tx) = (py, px) reads lo[a][b]: always inside
#pragma unroll
      for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int ty = 0; ty < 2; ++ty)
#pragma unroll
          for (int tx = 0; tx < 2; ++tx) {
            const bool ok = ((g_rmask >> ty) & 1u) && ((g_cmask >> tx) & 1u);
            lg_static |= (unsigned)ok << (l * 4 + ty * 2 + tx);
            lg_rel[l * 4 + ty * 2 + tx] = ok ? l * HWlo - ty * p.Wlo - tx : safe;
          }
    } else if constexpr (BKD == B_WGRAD || BKD == B_WGRAD3) {
      constexpr int ST = BKD == B_WGRAD ? 2 : 1;  // stride of the convolution
      lg_k = k_begin + g_kh * 8;
      c_img = lg_k / HWlo;
      const int r = lg_k - c_img * HWlo;
      const int cy = r / p.Wlo;
      c_x = r - cy * p.Wlo;
      c_iy = ST * cy - 1 + g_ky;
    }
  }

  // Full K stages need no predicates at all: rows m >= M of A and columns n >= N of B are loaded from a
  // clamped valid address and may hold anything — they only reach accumulator rows / columns that the
  // epilogue never stores.  Only the halo taps of the conv gathers must be exact zeros; that is one
  // v_and with a loop-invariant mask word per element when the registers go to LDS.  A partial last
  // stage (K % 16 != 0) is handled after the loop by the predicated generic loaders.
  unsigned lg_mask[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) lg_mask[j] = ((lg_static >> j) & 1u) ? 0xFFFFFFFFu : 0u;
  const bool a_batched = p.a_img != 0, b_batched = p.b_img != 0;
  if constexpr (VEC) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      if (!la_st[i]) la_off[i] = 0;
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      if constexpr (BKD == B_NCONTIG || BKD == B_KCONTIG || BKD == B_TAPN || BKD == B_TAPK)
        if (!lb_st[i]) lb_off[i] = 0;
  }

  auto load_a_lean = [&]() {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      ra[i] = wfae::ld4(Ap + la_off[i]);
      if constexpr (AK == A_KCONTIG) {
        la_off[i] += BK;
        if (a_batched) {  // wave-uniform: plain weight matrices skip the image-wrap arithmetic
          la_pk[i] += BK;
          if (la_pk[i] >= p.a_hw) {
            la_pk[i] -= p.a_hw;
            la_off[i] += p.a_img - p.a_hw;
          }
        }
      } else {
        la_off[i] += (long)BK * p.a_ld;
      }
    }
    a_okbits = 0xFFFFFFFFu;
  };

  auto load_b_lean = [&]() {
    if constexpr (BKD == B_NCONTIG || BKD == B_KCONTIG || BKD == B_TAPN || BKD == B_TAPK) {
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        if constexpr (BKD == B_TAPN || BKD == B_TAPK)
          rb[i] = load4u(Bp + lb_off[i]);
        else
          rb[i] = wfae::ld4(Bp + lb_off[i]);
        if constexpr (PRO == 1 && BKD == B_NCONTIG) {  // full stages: k < K, no clamp needed
          pb_s[i] = p.b_scale[pb_k + i * 8];
          pb_h[i] = p.b_shift[pb_k + i * 8];
        }
        if constexpr (BKD == B_NCONTIG || BKD == B_TAPN) {
          lb_off[i] += (long)BK * p.b_ld;
        } else {
          lb_off[i] += BK;
          if (b_batched) {
            lb_pk[i] += BK;
            if (lb_pk[i] >= p.b_hw) {
              lb_pk[i] -= p.b_hw;
              lb_off[i] += p.b_img - p.b_hw;
            }
          }
        }
      }
      if constexpr (PRO == 1 && BKD == B_NCONTIG) pb_k += BK;
      if constexpr (BKD == B_TAPN) {  // next tap after Chi channels: rewind the channel walk, shift by one tap
        lt_c += BK;
        if (lt_c >= p.Chi) {
          lt_c = 0;
          const long d = (long)(((lt_tap & 3) == 3) ? p.Wlo - 3 : 1) - (long)p.Chi * p.b_ld;
          ++lt_tap;
#pragma unroll
          for (int i = 0; i < B_IT; ++i) lb_off[i] += d;
        }
      }
      b_okbits = 0xFFFFFFFFu;
    } else if constexpr (BKD == B_DOWN) {
      const float* __restrict__ bp = Bp + lg_off;
#pragma unroll
      for (int j = 0; j < 8; ++j) rg[j] = bp[lg_rel[j]];
      lg_off += (long)H * W;
    } else if constexpr (BKD == B_UP) {
      const float* __restrict__ bp = Bp + lg_off;
#pragma unroll
      for (int j = 0; j < 8; ++j) rg[j] = bp[lg_rel[j]];
      lg_off += (long)4 * HWlo;
    } else {  // B_WGRAD / B_WGRAD3: 8 consecutive output pixels of one row (Wlo % 8 == 0)
      constexpr int ST = BKD == B_WGRAD ? 2 : 1;
      const int Hh = ST * p.Hlo, Ww = ST * p.Wlo;
      const bool okrow = c_iy >= 0 && c_iy < Hh;
      const int iyc = min(max(c_iy, 0), Hh - 1);
      const int ixb = ST * c_x - 1 + g_kx;
      const unsigned m0w = (okrow && ixb >= 0) ? 0xFFFFFFFFu : 0u;
      const unsigned mmw = okrow ? 0xFFFFFFFFu : 0u;
      const unsigned m7w = (okrow && ixb + 7 * ST < Ww) ? 0xFFFFFFFFu : 0u;
      // 32-bit element offset (host guarantees the tensor has < 2^31 elements)
      const float* __restrict__ bp = Bp + ((c_img * p.Chi + g_hi) * Hh + iyc) * Ww;
      rg[0] = bp[max(ixb, 0)];
#pragma unroll
      for (int j = 1; j < 7; ++j) rg[j] = bp[ixb + ST * j];
      rg[7] = bp[min(ixb + 7 * ST, Ww - 1)];
      lg_mask[0] = m0w;
#pragma unroll
      for (int j = 1; j < 7; ++j) lg_mask[j] = mmw;
      lg_mask[7] = m7w;
      // advance the cursor by one stage (BK = 16 pixels; rows are at least 8 pixels wide)
      c_x += BK;
#pragma unroll
      for (int w = 0; w < 2; ++w)
        if (c_x >= p.Wlo) {
          c_x -= p.Wlo;
          c_iy += ST;
          if (c_iy - g_ky + 1 >= Hh) {  // ST * cy reached the image height: next image
            c_iy -= Hh;
            c_img += 1;
          }
        }
    }
  };

  auto load_a = [&](int k0) {
    if constexpr (AK == A_KCONTIG) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int idx = t + i * NT;
        const int m = m0 + kc_row(idx);
        const int k = k0 + kc_k(idx);
        const bool ok = (A_IT * NT == A_CNT || idx < A_CNT) && m < p.M && k < k_end;
        if constexpr (VEC) {
          const int kk = ok ? k : 0;
          const int img = kk / p.a_hw;
          ra[i] = ld4raw(Ap, (long)img * p.a_img + (long)m * p.a_ld + (kk - img * p.a_hw), ok);
          a_okbits = (a_okbits & ~(1u << i)) | ((unsigned)ok << i);
        } else {
          a_okbits |= 1u << i;
          float e[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool okj = ok && (k + j) < k_end;
            const int kk = okj ? k + j : 0;
            const int img = kk / p.a_hw;
            e[j] = ld1(Ap, (long)img * p.a_img + (long)m * p.a_ld + (kk - img * p.a_hw), okj);
          }
          ra[i] = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
    } else {  // A_MCONTIG
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int idx = t + i * NT;
        const int kr = idx / (BM / 4);
        const int m = m0 + (idx % (BM / 4)) * 4;
        const int k = k0 + kr;
        const bool ok = (A_IT * NT == A_CNT || idx < A_CNT) && k < k_end && m < p.M;
        const long off = (long)k * p.a_ld + m;
        if constexpr (VEC) {
          ra[i] = ld4raw(Ap, off, ok);
          a_okbits = (a_okbits & ~(1u << i)) | ((unsigned)ok << i);
        } else {
          a_okbits |= 1u << i;
          ra[i] = make_float4(ld1(Ap, off, ok), ld1(Ap, off + 1, ok && m + 1 < p.M),
                              ld1(Ap, off + 2, ok && m + 2 < p.M), ld1(Ap, off + 3, ok && m + 3 < p.M));
        }
      }
    }
  };

  bool lean_regs = VEC;  // which loader filled the staging registers (the K-tail stage uses the predicated generic loaders)
  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int idx = t + i * NT;
      if (A_IT * NT == A_CNT || idx < A_CNT) {
        float4 v = ra[i];
        if constexpr (PRO == 2) v = bn_gelu4(v, pa_s[i], pa_h[i]);
        if (!lean_regs) v = sel4(v, a_okbits, i);  // wave-uniform branch: full stages carry no predicate
        if constexpr (AK == A_KCONTIG) {
          const int ml = kc_row(idx), kq = kc_k(idx);
          As[buf][kq + 0][ml] = v.x;
          As[buf][kq + 1][ml] = v.y;
          As[buf][kq + 2][ml] = v.z;
          As[buf][kq + 3][ml] = v.w;
        } else {
          const int kr = idx / (BM / 4), ml = (idx % (BM / 4)) * 4;
          *reinterpret_cast<float4*>(&As[buf][kr][ml]) = v;
        }
      }
    }
  };

  auto load_b = [&](int k0) {
    if constexpr (BKD == B_NCONTIG) {
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int k = k0 + (t >> 5) + i * 8;
        const bool ok = bn_ok && k < k_end;
        if constexpr (PRO == 1) {
          pb_s[i] = p.b_scale[k < k_end ? k : 0];
          pb_h[i] = p.b_shift[k < k_end ? k : 0];
        }
        if constexpr (VEC) {
          rb[i] = ld4raw(Bp, bn_base + (long)k * p.b_ld, ok);
          b_okbits = (b_okbits & ~(1u << i)) | ((unsigned)ok << i);
        } else {
          b_okbits |= 1u << i;
          const int n = n0 + (t & 31) * 4;
          float e[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool okj = ok && (n + j) < p.N;
            const int nn = okj ? n + j : 0;
            const int img = nn / p.b_hw;
            e[j] = ld1(Bp, (long)img * p.b_img + (long)k * p.b_ld + (nn - img * p.b_hw), okj);
          }
          rb[i] = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
    } else if constexpr (BKD == B_KCONTIG) {
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int idx = t + i * NT;
        const int n = n0 + kc_row(idx);
        const int k = k0 + kc_k(idx);
        const bool ok = n < p.N && k < k_end;
        if constexpr (VEC) {
          const int kk = ok ? k : 0;
          const int img = kk / p.b_hw;
          rb[i] = ld4raw(Bp, (long)img * p.b_img + (long)n * p.b_ld + (kk - img * p.b_hw), ok);
          b_okbits = (b_okbits & ~(1u << i)) | ((unsigned)ok << i);
        } else {
          b_okbits |= 1u << i;
          float e[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool okj = ok && (k + j) < k_end;
            const int kk = okj ? k + j : 0;
            const int img = kk / p.b_hw;
            e[j] = ld1(Bp, (long)img * p.b_img + (long)n * p.b_ld + (kk - img * p.b_hw), okj);
          }
          rb[i] = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
    } else if constexpr (BKD == B_TAPN) {
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int k = k0 + (t >> 5) + i * 8;
        const bool ok = bn_ok && k < k_end;
        const int kk = ok ? k : 0;
        const int tap = kk / p.Chi, c = kk - tap * p.Chi;
        const long row = (long)c * p.b_ld + tap_off(tap);
        if constexpr (VEC) {
          rb[i] = ld4raw_u(Bp, bn_base + row, ok);
          b_okbits = (b_okbits & ~(1u << i)) | ((unsigned)ok << i);
        } else {
          b_okbits |= 1u << i;
          const int n = n0 + (t & 31) * 4;
          float e[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool okj = ok && (n + j) < p.N;
            const int nn = okj ? n + j : 0;
            const int img = nn / p.b_hw;
            e[j] = ld1(Bp, (long)img * p.b_img + row + (nn - img * p.b_hw), okj);
          }
          rb[i] = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
    } else if constexpr (BKD == B_TAPK) {
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int idx = t + i * NT;
        const int n = n0 + kc_row(idx);
        const int k = k0 + kc_k(idx);
        const bool ok = n < p.N && k < k_end;
        const int nn = n < p.N ? n : 0;
        const int tap = nn / p.Chi, c = nn - tap * p.Chi;
        const long row = (long)c * p.b_ld + tap_off(tap);
        if constexpr (VEC) {
          const int kk = ok ? k : 0;
          const int img = kk / p.b_hw;
          rb[i] = ld4raw_u(Bp, (long)img * p.b_img + row + (kk - img * p.b_hw), ok);
          b_okbits = (b_okbits & ~(1u << i)) | ((unsigned)ok << i);
        } else {
          b_okbits |= 1u << i;
          float e[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool okj = ok && (k + j) < k_end;
            const int kk = okj ? k + j : 0;
            const int img = kk / p.b_hw;
            e[j] = ld1(Bp, (long)img * p.b_img + row + (kk - img * p.b_hw), okj);
          }
          rb[i] = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
    } else if constexpr (BKD == B_DOWN) {
      // k = hi*16 + ky*4 + kx ; one hi channel per stage
      const int hi = k0 >> 4;
      const long base = g_base + (long)hi * H * W + (long)(g_kh * 2) * W;
      const bool okc = hi < p.Chi;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = okc && ((g_rmask >> i) & 1u) && ((g_cmask >> j) & 1u);
          rg[i * 4 + j] = ld1raw(Bp, base + i * W + j, ok);
          g_okbits = (g_okbits & ~(1u << (i * 4 + j))) | ((unsigned)ok << (i * 4 + j));
        }
    } else if constexpr (BKD == B_UP) {
      // k = lo*4 + ty*2 + tx ; four lo channels per stage, this thread two of them
      const int lo0 = (k0 >> 2) + g_kh * 2;
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const int lo = lo0 + l;
        const long base = g_base + (long)lo * HWlo;
        const bool okc = lo < p.Clo;
#pragma unroll
        for (int ty = 0; ty < 2; ++ty)
#pragma unroll
          for (int tx = 0; tx < 2; ++tx) {
            const bool ok = okc && ((g_rmask >> ty) & 1u) && ((g_cmask >> tx) & 1u);
            rg[l * 4 + ty * 2 + tx] = ld1raw(Bp, base - ty * p.Wlo - tx, ok);
            g_okbits = (g_okbits & ~(1u << (l * 4 + ty * 2 + tx))) | ((unsigned)ok << (l * 4 + ty * 2 + tx));
          }
      }
    } else if constexpr (BKD == B_WGRAD3) {
      // stride-1 3x3: k = (img, y, x) pixel index, n = (ci, ky, kx); Hlo x Wlo is the image size
      const int kb = k0 + g_kh * 8;
      const int Hh = p.Hlo, Ww = p.Wlo;
      if ((Ww & 7) == 0) {
        const bool okk = g_ok && kb < k_end;
        const int kc = okk ? kb : 0;
        const int img = kc / HWlo;
        const int r = kc - img * HWlo;
        const int yy = r / Ww, x0 = r - yy * Ww;
        const int iy = yy + g_ky - 1;
        const bool ok = okk && iy >= 0 && iy < Hh;
        const int ixb = x0 + g_kx - 1;
        const long base = (((long)img * p.Chi + g_hi) * Hh + iy) * Ww + ixb;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool okj = ok && (ixb + j) >= 0 && (ixb + j) < Ww;
          rg[j] = ld1raw(Bp, base + j, okj);
          g_okbits = (g_okbits & ~(1u << j)) | ((unsigned)okj << j);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = kb + j;
          const bool okk = g_ok && k < k_end;
          const int kc = okk ? k : 0;
          const int img = kc / HWlo;
          const int r = kc - img * HWlo;
          const int yy = r / Ww, xx = r - yy * Ww;
          const int iy = yy + g_ky - 1, ix = xx + g_kx - 1;
          rg[j] = ld1(Bp, (((long)img * p.Chi + g_hi) * Hh + iy) * Ww + ix,
                      okk && iy >= 0 && iy < Hh && ix >= 0 && ix < Ww);
          g_okbits |= 1u << j;
        }
      }
    } else {  // B_WGRAD: k = (img, oy, ox) pixel index, n = (hi, ky, kx)
      const int kb = k0 + g_kh * 8;
      if ((p.Wlo & 7) == 0) {
        // 8 consecutive pixels share (img, oy)
        const bool okk = g_ok && kb < k_end;
        const int kc = okk ? kb : 0;
        const int img = kc / HWlo;
        const int r = kc - img * HWlo;
        const int oy = r / p.Wlo, ox0 = r - oy * p.Wlo;
        const int iy = 2 * oy - 1 + g_ky;
        const bool ok = okk && iy >= 0 && iy < H;
        const int ixb = 2 * ox0 - 1 + g_kx;
        const long base = ((long)img * p.Chi + g_hi) * H * W + (long)iy * W + ixb;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ix = ixb + 2 * j;
          const bool okj = ok && ix >= 0 && ix < W;
          rg[j] = ld1raw(Bp, base + 2 * j, okj);
          g_okbits = (g_okbits & ~(1u << j)) | ((unsigned)okj << j);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = kb + j;
          const bool okk = g_ok && k < k_end;
          const int kc = okk ? k : 0;
          const int img = kc / HWlo;
          const int r = kc - img * HWlo;
          const int oy = r / p.Wlo, ox = r - oy * p.Wlo;
          const int iy = 2 * oy - 1 + g_ky, ix = 2 * ox - 1 + g_kx;
          rg[j] = ld1(Bp, ((long)img * p.Chi + g_hi) * H * W + (long)iy * W + ix,
                      okk && iy >= 0 && iy < H && ix >= 0 && ix < W);
          g_okbits |= 1u << j;
        }
      }
    }
  };

  auto store_b = [&](int buf) {
    if constexpr (BKD == B_NCONTIG || BKD == B_TAPN) {
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        float4 v = rb[i];
        if constexpr (PRO == 1) v = bn_gelu4(v, pb_s[i], pb_h[i]);
        if (!lean_regs) v = sel4(v, b_okbits, i);  // after the prologue: elements beyond K / N must stay exact zeros
        *reinterpret_cast<float4*>(&Bs[buf][(t >> 5) + i * 8][(t & 31) * 4]) = v;
      }
    } else if constexpr (BKD == B_KCONTIG || BKD == B_TAPK) {
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        const int idx = t + i * NT;
        const int nl = kc_row(idx), kq = kc_k(idx);
        float4 v = rb[i];
        if constexpr (PRO == 1) v = bn_gelu4(v, pb_s[i], pb_h[i]);
        if (!lean_regs) v = sel4(v, b_okbits, i);
        Bs[buf][kq + 0][nl] = v.x;
        Bs[buf][kq + 1][nl] = v.y;
        Bs[buf][kq + 2][nl] = v.z;
        Bs[buf][kq + 3][nl] = v.w;
      }
    } else {
      if (lean_regs) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          Bs[buf][g_kh * 8 + j][g_nl] = __uint_as_float(__float_as_uint(rg[j]) & lg_mask[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) Bs[buf][g_kh * 8 + j][g_nl] = ((g_okbits >> j) & 1u) ? rg[j] : 0.f;
      }
    }
  };

  // ------------------------------------------------------------- main loop
  acc_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < NREG; ++r) acc[i][j][r] = 0.f;

  const int wm0 = (wave / WNW) * WROWS;
  const int wn0 = (wave % WNW) * WCOLS;
  const int l31 = lane & 31, lh = lane >> 5;

  // INTERLEAVED sub-tiles: MFMA tile i of a wave covers rows  wm0 + l*TM + i  (l = 0..31), not a
  // contiguous 32-row block, and tile j covers columns  wn0 + l*TN + j.  The TM (TN) operand values a
  // lane needs for one k are then ADJACENT in the natural As[k][m] / Bs[k][n] images, so one
  // ds_read_b128 / ds_read_b64 replaces 4 / 2 ds_read_b32.  A register-free probe of this loop
  // (tools/probe/lds_mfma.hip) sustains 149 TF for 64x64 wave tiles against 114 TF with scalar reads:
  // the LDS instruction count, not the MFMA pipe, was the ceiling.  Only the epilogue's row/column
  // map changes.  Fragments of k-step s+1 are read before the MFMAs of k-step s are issued.
  typedef float vecA __attribute__((ext_vector_type(TM)));
  typedef float vecB __attribute__((ext_vector_type(TN)));
  auto compute = [&](int buf) {
    static_assert(MF == 32, "interleaved operand reads are built for v_mfma_f32_32x32x2_f32");
    if constexpr (PREC == 1) {
      // lane half lh supplies k = 8 lh .. 8 lh + 7 of both operands (any k order both operands share is a
      // valid contraction order); BK == 16 is exactly one MFMA deep
      static_assert(BK == 16, "one v_mfma_f32_32x32x16_bf16 per stage");
      vecA af[8];
      vecB bfr[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        af[q] = *reinterpret_cast<const vecA*>(&As[buf][lh * 8 + q][wm0 + l31 * TM]);
        bfr[q] = *reinterpret_cast<const vecB*>(&Bs[buf][lh * 8 + q][wn0 + l31 * TN]);
      }
      bf16x8 a8[TM], b8[TN];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a8[i][q] = (__bf16)af[q][i];
#pragma unroll
        for (int j = 0; j < TN; ++j) b8[j][q] = (__bf16)bfr[q][j];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
      return;
    }
    if constexpr (PREC == 2) {
      // fp32 product on the bf16 pipe (splitgemm.hip): every fragment value is split into its three bf16 planes in
      // registers (exact: x == h + m + l), six MFMAs per tile and stage.  ~5.5 VALU instructions per value: the stage
      // becomes VALU-bound at about half the fp32 instruction's 2048 matrix-pipe cycles.
      static_assert(BK == 16, "one 16-deep v_mfma_f32_32x32x16_bf16 k-slab per stage");
      vecA af[8];
      vecB bfr[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        af[q] = *reinterpret_cast<const vecA*>(&As[buf][lh * 8 + q][wm0 + l31 * TM]);
        bfr[q] = *reinterpret_cast<const vecB*>(&Bs[buf][lh * 8 + q][wn0 + l31 * TN]);
      }
      bf16x8 a3[TM][3], b3[TN][3];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float x = af[q][i];
          const __bf16 h = (__bf16)x;
          const float r1 = x - (float)h;
          const __bf16 m = (__bf16)r1;
          a3[i][0][q] = h;
          a3[i][1][q] = m;
          a3[i][2][q] = (__bf16)(r1 - (float)m);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float x = bfr[q][j];
          const __bf16 h = (__bf16)x;
          const float r1 = x - (float)h;
          const __bf16 m = (__bf16)r1;
          b3[j][0][q] = h;
          b3[j][1][q] = m;
          b3[j][2][q] = (__bf16)(r1 - (float)m);
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc_t c = acc[i][j];   // smallest terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][2], b3[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][0], b3[j][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][1], b3[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][1], b3[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][0], b3[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[i][0], b3[j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
      return;
    }
    constexpr int NSTEP = BK / 2;
    vecA a[2];
    vecB b[2];
    auto frag = [&](int st, int slot) {
      a[slot] = *reinterpret_cast<const vecA*>(&As[buf][st * 2 + lh][wm0 + l31 * TM]);
      b[slot] = *reinterpret_cast<const vecB*>(&Bs[buf][st * 2 + lh][wn0 + l31 * TN]);
    };
    frag(0, 0);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int cur = st & 1;
      if (st + 1 < NSTEP) frag(st + 1, cur ^ 1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
        }
    }
  };

  // Stages [0, nlean) are complete (16 k each) and use the lean loaders; a partial last stage and the
  // scalar (non-VEC) kernels use the predicated generic loaders.
  const int nstages = (k_end - k_begin + BK - 1) / BK;
  const int nlean = VEC ? (k_end - k_begin) / BK : 0;
  auto load_stage = [&](int s) {
    bool lean = false;
    if constexpr (VEC) lean = s < nlean;
    if (lean) {
      if constexpr (VEC) {
        load_a_lean();
        load_b_lean();
      }
    } else {
      load_a(k_begin + s * BK);
      load_b(k_begin + s * BK);
    }
    lean_regs = lean;
  };
  if (nstages > 0) {
    load_stage(0);
    store_a(0);
    store_b(0);
  }
  __syncthreads();
  for (int s = 0; s < nstages; ++s) {
    const int buf = s & 1;
    if (s + 1 < nstages) load_stage(s + 1);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch's consumers below the MFMAs
    compute(buf);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < nstages) {
      store_a(buf ^ 1);
      store_b(buf ^ 1);
    }
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue
  // C/D map of 32x32 MFMA: column = lane & 31, row R = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5);
  // with the interleaved sub-tiles: m = m0 + wm0 + R*TM + i,  n = n0 + wn0 + (lane & 31)*TN + j.
  if constexpr (VEC && EK != E_UP) {
    // Stage slices of the wave tile through LDS (the operand buffers are free after the final barrier)
    // and write whole 16-byte pieces: 4x fewer, 4x wider stores / residual loads.  One slice = the 8
    // tile rows R in [8g, 8g+8) of all TM sub-tiles = 8*TM consecutive output rows.
    constexpr int COLS = WCOLS;
    constexpr int SROWS = 8 * TM;        // rows per slice
    constexpr int F4R = COLS / 4;        // float4 per row
    constexpr int RPI = 64 / F4R;        // rows per wave-instruction
    static_assert(4 * SROWS * COLS <= A_FLOATS + B_FLOATS, "epilogue staging fits the operand buffers");
    float* sw = smem + wave * (SROWS * COLS);
    const int c4 = lane % F4R, rsub = lane / F4R;
    const int n = n0 + wn0 + c4 * 4;
    const bool nok = n < p.N;
    CT* cb = reinterpret_cast<CT*>(p.C);
    const CT* rbp = nullptr;
    long mstride;
    if constexpr (EK == E_BATCHED) {
      cb += (long)blockIdx.y * p.c_y;
      const int nn = nok ? n : 0;
      const int img = nn / p.c_hw;
      const int pn = nn - img * p.c_hw;
      cb += (long)img * p.c_img + pn;
      if (p.res) rbp = reinterpret_cast<const CT*>(p.res) + (long)img * p.res_img + pn;
      mstride = p.c_ld;
    } else {
      cb += ((long)z * gridDim.y + blockIdx.y) * p.M * p.N + n;
      mstride = p.N;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // R in [8g, 8g+8): registers 4g..4g+3, both lane halves
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int row = (r4 + 4 * lh) * TM + i;  // (R - 8g) * TM + i
#pragma unroll
          for (int j = 0; j < TN; ++j) sw[row * COLS + l31 * TN + j] = acc[i][j][4 * g + r4];
        }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < SROWS / RPI; ++it) {
        const int row = it * RPI + rsub;
        float4 v = *reinterpret_cast<const float4*>(&sw[row * COLS + c4 * 4]);
        const int m = m0 + wm0 + g * SROWS + row;
        if (nok && m < p.M) {
          CT* dst = cb + (long)m * mstride;
          if constexpr (EK == E_BATCHED) {
            if (p.bias) {
              const float bv = p.bias[m];
              v.x += bv; v.y += bv; v.z += bv; v.w += bv;
            }
            if (rbp) {
              const float4 rv = wfae::ld4(rbp + (long)m * mstride);
              v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            }
            if (p.beta) {
              const float4 ov = wfae::ld4(dst);
              v.x += ov.x; v.y += ov.y; v.z += ov.z; v.w += ov.w;
            }
          }
          if constexpr (sizeof(CT) == 2) {   // the BatchNorm sums below are those of the values a reader of C will see
            const unsigned q0 = pack_bf16(v.x, v.y), q1 = pack_bf16(v.z, v.w);
            v = make_float4(bf16_lo(q0), bf16_hi(q0), bf16_lo(q1), bf16_hi(q1));
          }
          wfae::st4(dst, v);
        }
        if constexpr (EK == E_BATCHED && WNW == 2) {
          if (p.stat_sum) {  // wave-uniform: every lane takes part in the shuffles
            double s1 = 0.0, s2 = 0.0;
            if (nok && m < p.M) {   // fp32 sums of four enter the fp64 reduction, exactly as in chan_reduce_kernel
              s1 = (double)((v.x + v.y) + (v.z + v.w));
              s2 = (double)(fmaf(v.x, v.x, v.y * v.y) + fmaf(v.z, v.z, v.w * v.w));
            }
            // the F4R = 16 lanes of one output row are one DPP row: four row_shr adds leave the total in its lane 15
            static_assert(F4R == 16, "one DPP row per output row");
            s1 = dpp_row_sum(s1);
            s2 = dpp_row_sum(s2);
            if (c4 == F4R - 1 && m < p.M) {
              const long prow = (long)(n0 / BN) * 2 + (wave % WNW);
              p.stat_sum[prow * p.M + m] = s1;
              p.stat_sq[prow * p.M + m] = s2;
            }
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn0 + l31 * TN + j;
      if (n >= p.N) continue;
      CT* cb;
      const CT* rbp = nullptr;
      long mstride;
      if constexpr (EK == E_BATCHED) {
        const int img = n / p.c_hw;
        const int pn = n - img * p.c_hw;
        cb = reinterpret_cast<CT*>(p.C) + (long)blockIdx.y * p.c_y + (long)img * p.c_img + pn;
        if (p.res) rbp = reinterpret_cast<const CT*>(p.res) + (long)img * p.res_img + pn;
        mstride = p.c_ld;
      } else if constexpr (EK == E_SLAB) {
        cb = reinterpret_cast<CT*>(p.C) + ((long)z * gridDim.y + blockIdx.y) * p.M * p.N + n;
        mstride = p.N;
      } else {  // E_UP
        const int img = n / HWlo;
        const int r = n - img * HWlo;
        const int a = r / p.Wlo, b = r - a * p.Wlo;
        cb = reinterpret_cast<CT*>(p.C) + (long)img * p.Chi * H * W + (long)(2 * a + py) * W + (2 * b + px);
        mstride = (long)H * W;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
          const int m = m0 + wm0 + ((r & 3) + 8 * (r >> 2) + 4 * lh) * TM + i;
          if (m < p.M) {
            float v = acc[i][j][r];
            if constexpr (EK == E_BATCHED) {
              if (p.bias) v += p.bias[m];
              if (rbp) v += wfae::ld1(rbp + (long)m * mstride);
              if (p.beta) v += wfae::ld1(cb + (long)m * mstride);
            }
            wfae::st1(cb + (long)m * mstride, v);
          }
        }
    }
  }
}

// transpose_m > 0: the slab is [M = transpose_m][N] and `out` is [N][M]
__device__ __forceinline__ long out_index(long i, int N, int transpose_m) {
  if (transpose_m <= 0) return i;
  const long m = i / N, n = i - m * N;
  return n * transpose_m + m;
}

__global__ void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                     const float* __restrict__ bias_n, long MN, int N, int splits,
                                     int beta, int transpose_m) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= MN) return;
  float s = 0.f;
  for (int z = 0; z < splits; ++z) s += slab[(long)z * MN + i];
  if (bias_n) s += bias_n[i % N];
  const long o = out_index(i, N, transpose_m);
  if (beta) s += out[o];
  out[o] = s;
}

// Many slabs, few outputs: 64 outputs x 16 z-lanes per block, each z-lane sums every 16th
// slab (4 independent accumulators), then a fixed-order LDS combine.
__global__ __launch_bounds__(1024) void splitk_reduce_wide_kernel(const float* __restrict__ slab,
                                                                  float* __restrict__ out,
                                                                  const float* __restrict__ bias_n, long MN,
                                                                  int N, int splits, int beta, int transpose_m) {
  __shared__ float sm[16][65];
  const int il = threadIdx.x & 63, zl = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + il;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < MN) {
    int z = zl;
    for (; z + 48 < splits; z += 64) {
      s0 += slab[(long)z * MN + i];
      s1 += slab[(long)(z + 16) * MN + i];
      s2 += slab[(long)(z + 32) * MN + i];
      s3 += slab[(long)(z + 48) * MN + i];
    }
    for (; z < splits; z += 16) s0 += slab[(long)z * MN + i];
  }
  sm[zl][il] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (zl == 0 && i < MN) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += sm[k][il];
    if (bias_n) s += bias_n[i % N];
    const long o = out_index(i, N, transpose_m);
    if (beta) s += out[o];
    out[o] = s;
  }
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

// ---- flat-shift layout helpers of the stride-1 4x4 convolution --------------------------------
// dst plane (stride P, rows of Wp floats) = src plane (H x W) placed at (oy, ox), zero elsewhere
__global__ void embed_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int Wp, int P,
                             int oy, int ox, long total) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long pl = i / P;
    const int r = (int)(i - pl * P);
    const int yy = r / Wp - oy, xx = r % Wp - ox;
    dst[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? src[(pl * H + yy) * W + xx] : 0.f;
  }
}

// dst plane (H x W) = the first W columns of the first H rows of a src plane (stride P, rows of Wp)
__global__ void extract_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int Wp, int P,
                               long total) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int xx = (int)(i % W);
    const long r = i / W;
    const int yy = (int)(r % H);
    const long pl = r / H;
    dst[i] = src[pl * P + (long)yy * Wp + xx];
  }
}

// forward:    wp[co][tap*Cin + ci]        = w[co][ci][tap]
// transposed: wp[ci][(15 - tap)*Cout + co] = w[co][ci][tap]     (flipped taps, swapped channels)
__global__ void pack_taps_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin,
                                 int transposed) {
  const long total = (long)Cout * Cin * 16;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int tap = (int)(i & 15);
  const long r = i >> 4;
  const int ci = (int)(r % Cin), co = (int)(r / Cin);
  const long o = transposed ? ((long)ci * 16 + (15 - tap)) * Cout + co : ((long)co * 16 + tap) * Cin + ci;
  wp[o] = w[i];
}

// dw[co][ci][tap] (+)= dwp[co][tap*Cin + ci]
__global__ void unpack_taps_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Cout, int Cin,
                                   int beta) {
  const long total = (long)Cout * Cin * 16;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int tap = (int)(i & 15);
  const long r = i >> 4;
  const int ci = (int)(r % Cin), co = (int)(r / Cin);
  float v = dwp[((long)co * 16 + tap) * Cin + ci];
  if (beta) v += dw[i];
  dw[i] = v;
}

// Tile selection: BM x 128 x 16 with 4 waves.  BM = 256 (wave tile 128 x 64, 128 accumulator
// registers) halves the loads / LDS traffic / address arithmetic per MFMA — the SIMD issues VALU, LDS
// and MFMA instructions from one port, so non-MFMA instructions per MFMA set the achieved rate
// (PMC: SQ_VALU_MFMA_BUSY_CYCLES vs SQ_INSTS_VALU).
// tile height: the tallest of 128 / 64 / 32 rows that M fills, stepped down while the grid would leave most of the
// chip's 1024 resident-block slots empty (small batches, the 24x24 stage: 288 blocks of 128 rows ran 0.157 ms,
// 576 of 64 rows 0.132 ms).  Large grids are unaffected.
inline int pick_bm(int M, long col_blocks) {
  int bm = M > 64 ? 128 : (M > 32 ? 64 : 32);
  while (bm > 32 && (long)cdiv(M, bm) * col_blocks < 512) bm >>= 1;
  return bm;
}

template <int AK, int BKD, int EK, bool VEC, int PRO = 0, typename AT = float, typename BT = float, typename CT = float>
int launch_gemm_v(const GemmP& p_in, int zdim, hipStream_t st, const char* what, int ydim = 1) {
  constexpr bool use256 = true;
  GemmP p = p_in;
  const int ntiles = cdiv(p.N, BN);
  dim3 block(NT);
  bool big = false;
  constexpr bool all_f32 = std::is_same<AT, float>::value && std::is_same<BT, float>::value && std::is_same<CT, float>::value;
  if constexpr (VEC) {
    if (wfae::matmul_precision() == WFAE_PRECISION_BF16) {
      // bf16 operands: the MFMA time of a stage falls 16x, the kernels turn loader / HBM bound and the
      // 128-row tile (3 waves per SIMD) hides that latency best
      const int bm = pick_bm(p.M, (long)ntiles * ydim * zdim);
      if (bm == 128) {
        dim3 grid(cdiv(p.M, 128) * ntiles, ydim, zdim);
        hipLaunchKernelGGL((gemm_kernel<128, 2, 2, AK, BKD, EK, true, 32, 1, PRO, AT, BT, CT>), grid, block, 0, st, p);
      } else if (bm == 64) {
        dim3 grid(cdiv(p.M, 64) * ntiles, ydim, zdim);
        hipLaunchKernelGGL((gemm_kernel<64, 2, 2, AK, BKD, EK, true, 32, 1, PRO, AT, BT, CT>), grid, block, 0, st, p);
      } else {
        dim3 grid(cdiv(p.M, 32) * ntiles, ydim, zdim);
        hipLaunchKernelGGL((gemm_kernel<32, 1, 4, AK, BKD, EK, true, 32, 1, PRO, AT, BT, CT>), grid, block, 0, st, p);
      }
      return check_launch(what);
    }
  }
  if constexpr (!all_f32) {
    // bf16-stored activations exist only in the bf16 arithmetic mode (and on the vector kernels)
    return fail(WFAE_ERR_UNSUPPORTED, "%s: bf16 activation storage needs wfae_set_matmul_precision(WFAE_PRECISION_BF16) and "
                "vector-aligned operands", what);
  } else {
    if constexpr (VEC) {
      if constexpr (BKD == B_NCONTIG || BKD == B_KCONTIG || BKD == B_TAPN || BKD == B_TAPK) {
        // fp32 products on the bf16 matrix pipe, operands split in registers (PREC 2): for the MFMA-bound shapes
        constexpr int split_min_k = 128, split_min_m = 64;   // (K >= 32..128, M >= 32..64 all within 1 % in round 2)
        if (wfae::split_gemm_enabled() && p.K >= split_min_k && p.M >= split_min_m) {
          int bm = pick_bm(p.M, (long)ntiles * ydim * zdim);
          if (bm == 128) {
            dim3 grid(cdiv(p.M, 128) * ntiles, ydim, zdim);
            hipLaunchKernelGGL((gemm_kernel<128, 2, 2, AK, BKD, EK, true, 32, 2, PRO>), grid, block, 0, st, p);
            return check_launch(what);
          } else if (bm == 64) {
            dim3 grid(cdiv(p.M, 64) * ntiles, ydim, zdim);
            hipLaunchKernelGGL((gemm_kernel<64, 2, 2, AK, BKD, EK, true, 32, 2, PRO>), grid, block, 0, st, p);
            return check_launch(what);
          }
        }
      }
      // measured (tools/kbench.py): pays for the gather GEMMs once the grid fills the 512 resident slots twice
      constexpr bool gather = BKD == B_DOWN || BKD == B_UP || BKD == B_WGRAD;
      if ((gather || p.big_ok) && use256 && p.M >= 256 && (long)cdiv(p.M, 256) * ntiles * zdim * ydim >= 1024) {
        big = true;
        dim3 grid(cdiv(p.M, 256) * ntiles, ydim, zdim);
        hipLaunchKernelGGL((gemm_kernel<256, 2, 2, AK, BKD, EK, true, 32, 0, PRO>), grid, block, 0, st, p);
      }
    }
    int bm = big ? 256 : pick_bm(p.M, (long)ntiles * ydim * zdim);
    if (big) {
    } else if (bm == 128) {
      dim3 grid(cdiv(p.M, 128) * ntiles, ydim, zdim);
      hipLaunchKernelGGL((gemm_kernel<128, 2, 2, AK, BKD, EK, VEC, 32, 0, PRO>), grid, block, 0, st, p);
    } else if (bm == 64) {
      dim3 grid(cdiv(p.M, 64) * ntiles, ydim, zdim);
      hipLaunchKernelGGL((gemm_kernel<64, 2, 2, AK, BKD, EK, VEC, 32, 0, PRO>), grid, block, 0, st, p);
    } else {
      dim3 grid(cdiv(p.M, 32) * ntiles, ydim, zdim);
      hipLaunchKernelGGL((gemm_kernel<32, 1, 4, AK, BKD, EK, VEC, 32, 0, PRO>), grid, block, 0, st, p);
    }
    return check_launch(what);
  }
}

// p.a_vec / p.b_vec / p.c_vec say whether 16-byte accesses are legal for that operand; the
// all-vector kernel needs all three (gather operands are scalar by nature and always "legal").
template <int AK, int BKD, int EK>
int launch_gemm(const GemmP& p, int zdim, hipStream_t st, const char* what, int ydim = 1) {
  if (p.a_vec && p.b_vec && p.c_vec) return launch_gemm_v<AK, BKD, EK, true>(p, zdim, st, what, ydim);
  return launch_gemm_v<AK, BKD, EK, false>(p, zdim, st, what, ydim);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// choose split count so tiles*splits covers the chip a few times over
inline int pick_splits(int M, int N, int K, size_t ws_bytes, int* k_per_split, bool big_tiles = false) {
  int bm = big_tiles && M >= 256 ? 256 : (M > 64 ? 128 : (M > 32 ? 64 : 32));
  // short reductions (the 512..2048-deep linear layers of the token models): shrink the tile before splitting K —
  // 2048 x 512 x 512 used to run as 64 tiles x 16 splits of two stages each plus a 16-slab reduce
  if (K <= 4096)
    while (bm > 32 && (long)cdiv(M, bm) * cdiv(N, BN) < 512) bm >>= 1;
  const long tiles = (long)cdiv(M, bm) * cdiv(N, BN);
  const int stages = cdiv(K, BK);
  long want = (1024 + tiles - 1) / tiles;
  if (want < 1) want = 1;
  if (want > stages) want = stages;
  const size_t slab = (size_t)M * N * sizeof(float);
  while (want > 1 && (size_t)want * slab > ws_bytes) --want;
  int per = cdiv(stages, (int)want) * BK;
  *k_per_split = per;
  return cdiv(K, per);
}

}  // namespace

namespace wfae {
int slab_reduce(const float* slab, float* out, const float* bias_n, long MN, int N, int splits, int beta,
                hipStream_t st, int transpose_m) {
  if (splits >= 16) {
    hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3(cdiv(MN, 64)), dim3(1024), 0, st, slab, out, bias_n, MN, N,
                       splits, beta, transpose_m);
  } else {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(MN, 256)), dim3(256), 0, st, slab, out, bias_n, MN, N,
                       splits, beta, transpose_m);
  }
  return check_launch("slab_reduce");
}
}  // namespace wfae

namespace {
int splitk_finish(const float* slab, float* out, const float* bias_n, long MN, int N, int splits,
                  int beta, hipStream_t st, int transpose_m = 0) {
  return wfae::slab_reduce(slab, out, bias_n, MN, N, splits, beta, st, transpose_m);
}

}  // namespace

extern "C" {

extern "C++" {
template <typename T>
static int conv1x1_fwd_impl(const T* x, const float* bn_scale, const float* bn_shift, const float* w, const float* bias,
                            const T* res, int64_t res_img_stride, T* y, int NB, int Cin, int Cout, int HW,
                            double* stat_part, int64_t stat_capacity, int* stat_rows, wfae_stream_t stream) {
  WFAE_REQUIRE(x && w && y, WFAE_ERR_NULL_POINTER, "conv1x1_fwd: null pointer");
  WFAE_REQUIRE(NB > 0 && Cin > 0 && Cout > 0 && HW > 0, WFAE_ERR_BAD_SHAPE, "conv1x1_fwd: bad shape");
  WFAE_REQUIRE((int64_t)NB * HW < (1ll << 31), WFAE_ERR_BAD_SHAPE, "conv1x1_fwd: NB*HW too large");
  GemmP p = {};
  p.A = w; p.B = x; p.C = y; p.bias = bias; p.res = res;
  p.b_scale = bn_scale; p.b_shift = bn_shift;
  p.M = Cout; p.N = NB * HW; p.K = Cin; p.k_per_split = cdiv(Cin, BK) * BK;
  p.a_hw = Cin; p.a_img = 0; p.a_ld = Cin;
  p.b_hw = HW; p.b_img = (long)Cin * HW; p.b_ld = HW;
  p.c_hw = HW; p.c_img = (long)Cout * HW; p.c_ld = HW; p.res_img = res_img_stride;
  p.a_vec = (Cin % 4 == 0) && aligned16(w);
  p.b_vec = (HW % 4 == 0) && aligned16(x);
  p.c_vec = (HW % 4 == 0) && aligned16(y) && (!res || (aligned16(res) && res_img_stride % 4 == 0));
  const bool vec = p.a_vec && p.b_vec && p.c_vec;
  if (bn_scale) {
    WFAE_REQUIRE(vec, WFAE_ERR_UNSUPPORTED,
                 "conv1x1_fwd_bnact: needs Cin %% 4 == 0, HW %% 4 == 0 and 16-byte aligned tensors (Cin %d, HW %d)", Cin, HW);
    // one M tile of 256 rows when Cout fills it: every element of x is then loaded (and activated) exactly once
    p.big_ok = 1;
  }
  if (stat_rows) {
    // the statistics ride in the vector epilogue of the two-wave-column kernels; anything else reports 0 rows and the
    // caller runs wfae_bn_stats_train on y
    const int ntiles = cdiv(p.N, BN);
    const int64_t rows = 2 * (int64_t)ntiles;
    const bool ok = stat_part && vec && pick_bm(p.M, ntiles) != 32 && stat_capacity >= 2 * rows * Cout;
    *stat_rows = ok ? (int)rows : 0;
    if (ok) {
      p.stat_sum = stat_part;
      p.stat_sq = stat_part + rows * Cout;
    }
  }
  if constexpr (!std::is_same<T, float>::value) {
    WFAE_REQUIRE(vec, WFAE_ERR_UNSUPPORTED, "conv1x1_fwd (bf16 storage): needs Cin %% 4 == 0, HW %% 4 == 0, aligned tensors");
    if (bn_scale)
      return launch_gemm_v<A_KCONTIG, B_NCONTIG, E_BATCHED, true, 1, float, T, T>(p, 1, (hipStream_t)stream, "conv1x1_fwd_bnact");
    return launch_gemm_v<A_KCONTIG, B_NCONTIG, E_BATCHED, true, 0, float, T, T>(p, 1, (hipStream_t)stream, "conv1x1_fwd");
  } else {
    if (bn_scale) return launch_gemm_v<A_KCONTIG, B_NCONTIG, E_BATCHED, true, 1>(p, 1, (hipStream_t)stream, "conv1x1_fwd_bnact");
    return launch_gemm<A_KCONTIG, B_NCONTIG, E_BATCHED>(p, 1, (hipStream_t)stream, "conv1x1_fwd");
  }
}
}  // extern "C++"

int wfae_conv1x1_fwd(const float* x, const float* w, const float* bias, const float* res,
                     int64_t res_img_stride, float* y, int NB, int Cin, int Cout, int HW,
                     wfae_stream_t stream) {
  return conv1x1_fwd_impl(x, nullptr, nullptr, w, bias, res, res_img_stride, y, NB, Cin, Cout, HW, nullptr, 0, nullptr, stream);
}
/* bf16 storage: stat_part / stat_rows may be null (no sums); bn_scale / bn_shift may be null (no prologue) */
int wfae_conv1x1_fwd_bf16(const uint16_t* x, const float* bn_scale, const float* bn_shift, const float* w, const float* bias,
                          const uint16_t* res, int64_t res_img_stride, uint16_t* y, int NB, int Cin, int Cout, int HW,
                          double* stat_part, int64_t stat_capacity, int* stat_rows, wfae_stream_t stream) {
  WFAE_REQUIRE((bn_scale != nullptr) == (bn_shift != nullptr) && (stat_part != nullptr) == (stat_rows != nullptr),
               WFAE_ERR_NULL_POINTER, "conv1x1_fwd_bf16: scale / shift and stat_part / stat_rows go together");
  return conv1x1_fwd_impl(x, bn_scale, bn_shift, w, bias, res, res_img_stride, y, NB, Cin, Cout, HW, stat_part, stat_capacity,
                          stat_rows, stream);
}

int wfae_conv1x1_fwd_stats(const float* x, const float* w, const float* bias, const float* res,
                           int64_t res_img_stride, float* y, int NB, int Cin, int Cout, int HW, double* stat_part,
                           int64_t stat_capacity, int* stat_rows, wfae_stream_t stream) {
  WFAE_REQUIRE(stat_part && stat_rows, WFAE_ERR_NULL_POINTER, "conv1x1_fwd_stats: null pointer");
  return conv1x1_fwd_impl(x, nullptr, nullptr, w, bias, res, res_img_stride, y, NB, Cin, Cout, HW, stat_part, stat_capacity,
                          stat_rows, stream);
}

int wfae_conv1x1_fwd_bnact(const float* x, const float* bn_scale, const float* bn_shift, const float* w,
                           const float* bias, const float* res, int64_t res_img_stride, float* y, int NB, int Cin,
                           int Cout, int HW, double* stat_part, int64_t stat_capacity, int* stat_rows,
                           wfae_stream_t stream) {
  WFAE_REQUIRE(x && bn_scale && bn_shift && w && y, WFAE_ERR_NULL_POINTER, "conv1x1_fwd_bnact: null pointer");
  WFAE_REQUIRE((stat_part != nullptr) == (stat_rows != nullptr), WFAE_ERR_NULL_POINTER,
               "conv1x1_fwd_bnact: stat_part and stat_rows go together");
  return conv1x1_fwd_impl(x, bn_scale, bn_shift, w, bias, res, res_img_stride, y, NB, Cin, Cout, HW, stat_part, stat_capacity,
                          stat_rows, stream);
}

// dW (M = Cout, N = Cin) = dY * A^T, A = x or (bn_scale given) gelu(bn(x)) rebuilt in the loader of whichever operand x is: B
// normally, A when the roles are swapped (Cin is the small side: dW^T = A dY^T, no MFMA work spent on padding the 128-wide
// N tile; the slab reduce writes it back transposed)
extern "C++" {
template <typename T>
static int conv1x1_bwd_weight_impl(const T* dy, const T* x, const float* bn_scale, const float* bn_shift, float* dw, int NB,
                                   int Cin, int Cout, int HW, int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream,
                                   const char* what) {
  WFAE_REQUIRE(dy && x && dw, WFAE_ERR_NULL_POINTER, "%s: null pointer", what);
  WFAE_REQUIRE(NB > 0 && Cin > 0 && Cout > 0 && HW > 0, WFAE_ERR_BAD_SHAPE, "%s: bad shape", what);
  WFAE_REQUIRE((int64_t)NB * HW < (1ll << 31), WFAE_ERR_BAD_SHAPE, "%s: NB*HW too large", what);
  const size_t slab = (size_t)Cout * Cin * sizeof(float);
  WFAE_REQUIRE(ws && ws_bytes >= slab, WFAE_ERR_WORKSPACE, "%s: workspace %zu < %zu", what, ws_bytes, slab);
  const bool swap = Cin < Cout && Cin < 128;
  const T* a = swap ? x : dy;
  const T* b = swap ? dy : x;
  const int Ma = swap ? Cin : Cout, Nb = swap ? Cout : Cin;
  GemmP p = {};
  p.K = NB * HW;
  p.A = a; p.B = b; p.C = (float*)ws;
  p.b_scale = bn_scale; p.b_shift = bn_shift;
  p.M = Ma; p.N = Nb;
  p.a_hw = HW; p.a_img = (long)Ma * HW; p.a_ld = HW;
  p.b_hw = HW; p.b_img = (long)Nb * HW; p.b_ld = HW;
  p.a_vec = (HW % 4 == 0) && HW >= BK && aligned16(a);   // lean cursors step one image at most per stage
  p.b_vec = (HW % 4 == 0) && HW >= BK && aligned16(b);
  p.c_vec = (Nb % 4 == 0) && aligned16(ws);
  const bool vec = p.a_vec && p.b_vec && p.c_vec;
  const int splits = pick_splits(p.M, p.N, p.K, ws_bytes, &p.k_per_split);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if constexpr (!std::is_same<T, float>::value) {
    WFAE_REQUIRE(vec, WFAE_ERR_UNSUPPORTED, "%s (bf16 storage): needs HW %% 4 == 0, HW >= 16, channel counts %% 4 == 0", what);
    if (!bn_scale) rc = launch_gemm_v<A_KCONTIG, B_KCONTIG, E_SLAB, true, 0, T, T, float>(p, splits, st, what);
    else if (swap) rc = launch_gemm_v<A_KCONTIG, B_KCONTIG, E_SLAB, true, 2, T, T, float>(p, splits, st, what);
    else rc = launch_gemm_v<A_KCONTIG, B_KCONTIG, E_SLAB, true, 1, T, T, float>(p, splits, st, what);
  } else {
    if (bn_scale) {
      WFAE_REQUIRE(vec, WFAE_ERR_UNSUPPORTED, "%s: needs HW %% 4 == 0, HW >= 16, channel counts %% 4 == 0, aligned tensors", what);
      rc = swap ? launch_gemm_v<A_KCONTIG, B_KCONTIG, E_SLAB, true, 2>(p, splits, st, what)
                : launch_gemm_v<A_KCONTIG, B_KCONTIG, E_SLAB, true, 1>(p, splits, st, what);
    } else {
      rc = launch_gemm<A_KCONTIG, B_KCONTIG, E_SLAB>(p, splits, st, what);
    }
  }
  if (rc) return rc;
  return splitk_finish((float*)ws, dw, nullptr, (long)Cout * Cin, Nb, splits, accumulate, st, swap ? Ma : 0);
}
}  // extern "C++"

int wfae_conv1x1_bwd_weight_bnact(const float* dy, const float* x, const float* bn_scale, const float* bn_shift,
                                  float* dw, int NB, int Cin, int Cout, int HW, int accumulate, void* ws,
                                  size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(bn_scale && bn_shift, WFAE_ERR_NULL_POINTER, "conv1x1_bwd_weight_bnact: null pointer");
  return conv1x1_bwd_weight_impl(dy, x, bn_scale, bn_shift, dw, NB, Cin, Cout, HW, accumulate, ws, ws_bytes, stream,
                                 "conv1x1_bwd_weight_bnact");
}
/* bf16 storage: dy, x bf16; dw fp32; bn_scale / bn_shift null = plain weight gradient */
int wfae_conv1x1_bwd_weight_bf16(const uint16_t* dy, const uint16_t* x, const float* bn_scale, const float* bn_shift, float* dw,
                                 int NB, int Cin, int Cout, int HW, int accumulate, void* ws, size_t ws_bytes,
                                 wfae_stream_t stream) {
  WFAE_REQUIRE((bn_scale != nullptr) == (bn_shift != nullptr), WFAE_ERR_NULL_POINTER, "conv1x1_bwd_weight_bf16: scale / shift go together");
  return conv1x1_bwd_weight_impl(dy, x, bn_scale, bn_shift, dw, NB, Cin, Cout, HW, accumulate, ws, ws_bytes, stream,
                                 "conv1x1_bwd_weight_bf16");
}

extern "C++" {
template <typename T>
static int conv1x1_bwd_data_impl(const T* dy, const float* w, T* dx, int NB, int Cin, int Cout, int HW, wfae_stream_t stream) {
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
  p.c_vec = (HW % 4 == 0) && aligned16(dx);
  if constexpr (!std::is_same<T, float>::value) {
    WFAE_REQUIRE(p.a_vec && p.b_vec && p.c_vec, WFAE_ERR_UNSUPPORTED,
                 "conv1x1_bwd_data (bf16 storage): needs Cin %% 4 == 0, HW %% 4 == 0, aligned tensors");
    return launch_gemm_v<A_MCONTIG, B_NCONTIG, E_BATCHED, true, 0, float, T, T>(p, 1, (hipStream_t)stream, "conv1x1_bwd_data");
  } else {
    return launch_gemm<A_MCONTIG, B_NCONTIG, E_BATCHED>(p, 1, (hipStream_t)stream, "conv1x1_bwd_data");
  }
}
}  // extern "C++"
int wfae_conv1x1_bwd_data(const float* dy, const float* w, float* dx, int NB, int Cin, int Cout,
                          int HW, wfae_stream_t stream) {
  return conv1x1_bwd_data_impl(dy, w, dx, NB, Cin, Cout, HW, stream);
}
int wfae_conv1x1_bwd_data_bf16(const uint16_t* dy, const float* w, uint16_t* dx, int NB, int Cin, int Cout, int HW,
                               wfae_stream_t stream) {
  return conv1x1_bwd_data_impl(dy, w, dx, NB, Cin, Cout, HW, stream);
}

int wfae_conv1x1_bwd_weight(const float* dy, const float* x, float* dw, int NB, int Cin, int Cout,
                            int HW, int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream) {
  return conv1x1_bwd_weight_impl(dy, x, nullptr, nullptr, dw, NB, Cin, Cout, HW, accumulate, ws, ws_bytes, stream,
                                 "conv1x1_bwd_weight");
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
  p.c_vec = (Out % 4 == 0) && aligned16(ws);
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
  p.c_vec = (In % 4 == 0) && aligned16(dx);
  return launch_gemm<A_KCONTIG, B_NCONTIG, E_BATCHED>(p, 1, (hipStream_t)stream, "linear_bwd_data");
}

// dx = dy w with the Out dimension split over blockIdx.z (see wfae_linear_bwd_weight_splitk)
int wfae_linear_bwd_data_splitk(const float* dy, const float* w, float* dx, int B, int In, int Out, void* ws,
                                size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(dy && w && dx, WFAE_ERR_NULL_POINTER, "linear_bwd_data_splitk: null pointer");
  WFAE_REQUIRE(B > 0 && In > 0 && Out > 0, WFAE_ERR_BAD_SHAPE, "linear_bwd_data_splitk: bad shape");
  const size_t slab = (size_t)B * In * sizeof(float);
  WFAE_REQUIRE(ws && ws_bytes >= slab, WFAE_ERR_WORKSPACE, "linear_bwd_data_splitk: workspace %zu < %zu", ws_bytes, slab);
  GemmP p = {};  // dx(m=b, n=i) = sum_o dy[b,o] w[o,i]
  p.A = dy; p.B = w; p.C = (float*)ws;
  p.M = B; p.N = In; p.K = Out;
  p.a_hw = Out; p.a_img = 0; p.a_ld = Out;
  p.b_hw = In; p.b_img = 0; p.b_ld = In;
  p.a_vec = (Out % 4 == 0) && aligned16(dy);
  p.b_vec = (In % 4 == 0) && aligned16(w);
  p.c_vec = (In % 4 == 0) && aligned16(ws);
  const int splits = pick_splits(p.M, p.N, p.K, ws_bytes, &p.k_per_split);
  int rc = launch_gemm<A_KCONTIG, B_NCONTIG, E_SLAB>(p, splits, (hipStream_t)stream, "linear_bwd_data_splitk");
  if (rc) return rc;
  return splitk_finish((float*)ws, dx, nullptr, (long)B * In, In, splits, 0, (hipStream_t)stream);
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
  p.c_vec = (In % 4 == 0) && aligned16(dw);
  return launch_gemm<A_MCONTIG, B_NCONTIG, E_BATCHED>(p, 1, (hipStream_t)stream, "linear_bwd_weight");
}

// the same product with the batch dimension split over blockIdx.z: transformer-sized layers (a few thousand rows,
// 512..2048 features) give only Out/128 * In/128 = 16..64 tiles, far fewer than the 256 CUs
int wfae_linear_bwd_weight_splitk(const float* dy, const float* x, float* dw, int B, int In, int Out, int accumulate,
                                  void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(dy && x && dw, WFAE_ERR_NULL_POINTER, "linear_bwd_weight_splitk: null pointer");
  WFAE_REQUIRE(B > 0 && In > 0 && Out > 0, WFAE_ERR_BAD_SHAPE, "linear_bwd_weight_splitk: bad shape");
  const size_t slab = (size_t)Out * In * sizeof(float);
  WFAE_REQUIRE(ws && ws_bytes >= slab, WFAE_ERR_WORKSPACE, "linear_bwd_weight_splitk: workspace %zu < %zu", ws_bytes, slab);
  GemmP p = {};
  p.A = dy; p.B = x; p.C = (float*)ws;
  p.M = Out; p.N = In; p.K = B;
  p.a_ld = Out;
  p.b_hw = In; p.b_img = 0; p.b_ld = In;
  p.a_vec = (Out % 4 == 0) && aligned16(dy);
  p.b_vec = (In % 4 == 0) && aligned16(x);
  p.c_vec = (In % 4 == 0) && aligned16(ws);
  const int splits = pick_splits(p.M, p.N, p.K, ws_bytes, &p.k_per_split);
  int rc = launch_gemm<A_MCONTIG, B_NCONTIG, E_SLAB>(p, splits, (hipStream_t)stream, "linear_bwd_weight_splitk");
  if (rc) return rc;
  return splitk_finish((float*)ws, dw, nullptr, (long)Out * In, In, splits, accumulate, (hipStream_t)stream);
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
  p.b_vec = 1;
  p.c_vec = ((Hlo * Wlo) % 4 == 0) && aligned16(lo);
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
  p.b_vec = (Clo % 4 == 0);  // every K stage then holds four complete lo channels
  p.c_vec = 1;
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
  p.a_vec = (HWlo % 4 == 0) && HWlo >= BK && aligned16(lo);
  p.b_vec = (Wlo % 8 == 0) && (int64_t)NB * Chi * 4 * HWlo < (1ll << 31);  // pixel cursor + 32-bit offsets
  p.c_vec = aligned16(ws);
  p.Chi = Chi; p.Clo = Clo; p.Hlo = Hlo; p.Wlo = Wlo;
  // 256-row tiles (launch_gemm picks them for grids >= 1024 blocks): size the split for that tiling
  const int splits = pick_splits(p.M, p.N, p.K, ws_bytes, &p.k_per_split, true);
  int rc = launch_gemm<A_KCONTIG, B_WGRAD, E_SLAB>(p, splits, (hipStream_t)stream, "conv4x4s2_wgrad");
  if (rc) return rc;
  return splitk_finish((float*)ws, dw, nullptr, (long)Clo * Chi * 16, Chi * 16, splits, accumulate,
                       (hipStream_t)stream);
}

int wfae_gconv3x3_bwd_weight_bf16(const uint16_t* dy, const uint16_t* x, float* dw, int NB, int C, int H, int W, int groups,
                                  int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(dy && x && dw, WFAE_ERR_NULL_POINTER, "gconv3x3_bwd_weight_bf16: null pointer");
  WFAE_REQUIRE(NB > 0 && C > 0 && H > 0 && W > 0 && groups > 0 && groups <= 65535 && C % groups == 0,
               WFAE_ERR_BAD_SHAPE, "gconv3x3_bwd_weight_bf16: bad shape");
  const int rc = gconv3_wgrad_mfma(dy, x, dw, NB, C, H, W, groups, accumulate, ws, ws_bytes, (hipStream_t)stream);
  if (rc == WFAE_ERR_UNSUPPORTED)
    return fail(rc, "gconv3x3_bwd_weight_bf16: needs 4 / 8 / 16 / 32 channels per group, W %% 4 == 0, aligned tensors");
  return rc;
}

int wfae_gconv3x3_bwd_weight(const float* dy, const float* x, float* dw, int NB, int C, int H, int W,
                             int groups, int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(dy && x && dw, WFAE_ERR_NULL_POINTER, "gconv3x3_bwd_weight: null pointer");
  WFAE_REQUIRE(NB > 0 && C > 0 && H > 0 && W > 0 && groups > 0 && groups <= 65535 && C % groups == 0,
               WFAE_ERR_BAD_SHAPE, "gconv3x3_bwd_weight: bad shape");
  WFAE_REQUIRE((int64_t)NB * H * W < (1ll << 31), WFAE_ERR_BAD_SHAPE, "gconv3x3_bwd_weight: too large");
  const int cpg = C / groups;
  WFAE_REQUIRE(cpg <= 32, WFAE_ERR_UNSUPPORTED, "gconv3x3_bwd_weight: %d channels per group > 32", cpg);
  {
    // the dedicated one-tile-per-tap MFMA kernel (csrc/dconv.hip); it declines unaligned shapes
    const int rc = gconv3_wgrad_mfma(dy, x, dw, NB, C, H, W, groups, accumulate, ws, ws_bytes, (hipStream_t)stream);
    if (rc != WFAE_ERR_UNSUPPORTED) return rc;
  }
  {
    // fallbacks: the pixel-parallel VALU kernel (4 / 8 / 16 channels per group), then the generic GEMM
    const int rc = gconv3_wgrad_valu(dy, x, dw, NB, C, H, W, groups, accumulate, ws, ws_bytes, (hipStream_t)stream);
    if (rc != WFAE_ERR_UNSUPPORTED) return rc;
  }
  const int HW = H * W;
  GemmP p = {};
  p.A = dy; p.B = x; p.C = (float*)ws;
  p.M = cpg; p.N = cpg * 9; p.K = NB * HW;
  p.a_hw = HW; p.a_img = (long)C * HW; p.a_ld = HW;
  p.a_vec = (HW % 4 == 0) && HW >= BK && aligned16(dy);
  p.b_vec = (W % 8 == 0) && (int64_t)NB * C * HW < (1ll << 31);
  p.c_vec = (p.N % 4 == 0) && aligned16(ws);
  p.Chi = C; p.Hlo = H; p.Wlo = W; p.cpg = cpg;
  const size_t slab = (size_t)groups * p.M * p.N * sizeof(float);
  WFAE_REQUIRE(ws && ws_bytes >= slab, WFAE_ERR_WORKSPACE, "gconv3x3_bwd_weight: workspace %zu < %zu", ws_bytes, slab);
  const long tiles = (long)groups * cdiv(p.N, BN);
  const int stages = cdiv(p.K, BK);
  long want = (1024 + tiles - 1) / tiles;
  if (want > stages) want = stages;
  while (want > 1 && (size_t)want * slab > ws_bytes) --want;
  p.k_per_split = cdiv(stages, (int)want) * BK;
  const int splits = cdiv(p.K, p.k_per_split);
  int rc = launch_gemm<A_KCONTIG, B_WGRAD3, E_SLAB>(p, splits, (hipStream_t)stream, "gconv3x3_bwd_weight", groups);
  if (rc) return rc;
  return splitk_finish((float*)ws, dw, nullptr, (long)groups * p.M * p.N, p.N, splits, accumulate,
                       (hipStream_t)stream);
}

namespace {
struct FlatGeom {
  int pe, Ho, Wo, Hp, Wp, Q, P;
};
inline FlatGeom flat_geom(int H, int W, int pe) {
  FlatGeom g;
  g.pe = pe;
  g.Ho = H + 2 * pe - 3; g.Wo = W + 2 * pe - 3;
  g.Hp = H + 2 * pe; g.Wp = W + 2 * pe;
  g.Q = cdiv(g.Ho * g.Wp, 16) * 16;           // per-image GEMM extent (16 | Q: K stages never straddle images)
  g.P = cdiv(g.Q + 3 * g.Wp + 3, 4) * 4;      // plane stride: the last tap of the last position stays inside
  return g;
}
inline int ew_grid(long n) {
  long b = (n + 255) / 256;
  return (int)(b > 65535 * 4 ? 65535 * 4 : b);
}
}  // namespace

int wfae_conv4x4s1_fwd(const float* x, const float* w, float* y, int NB, int Cin, int Cout, int H, int W, int pad,
                       int transposed, void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(x && w && y, WFAE_ERR_NULL_POINTER, "conv4x4s1_fwd: null pointer");
  WFAE_REQUIRE(NB > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && pad >= 0 && pad <= 3, WFAE_ERR_BAD_SHAPE,
               "conv4x4s1_fwd: bad shape");
  const int Cc = transposed ? Cout : Cin;   // channels of the operand that is read
  const int M = transposed ? Cin : Cout;    // channels that are produced
  WFAE_REQUIRE(Cc % 16 == 0, WFAE_ERR_UNSUPPORTED, "conv4x4s1_fwd: %d input channels (need a multiple of 16)", Cc);
  const FlatGeom g = flat_geom(H, W, transposed ? 3 - pad : pad);
  WFAE_REQUIRE(g.Ho > 0 && g.Wo > 0, WFAE_ERR_BAD_SHAPE, "conv4x4s1_fwd: empty output");
  WFAE_REQUIRE((int64_t)NB * g.Q < (1ll << 31) && (int64_t)Cc * g.P < (1ll << 31), WFAE_ERR_BAD_SHAPE,
               "conv4x4s1_fwd: too large");
  const size_t n_xp = (size_t)NB * Cc * g.P, n_yq = (size_t)NB * M * g.Q, n_wp = (size_t)M * Cc * 16;
  const size_t need = (n_xp + n_yq + n_wp) * sizeof(float);
  WFAE_REQUIRE(ws && ws_bytes >= need, WFAE_ERR_WORKSPACE, "conv4x4s1_fwd: workspace %zu < %zu", ws_bytes, need);
  float* xp = (float*)ws;
  float* yq = xp + n_xp;
  float* wp = yq + n_yq;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(embed_kernel, dim3(ew_grid((long)n_xp)), dim3(256), 0, st, x, xp, H, W, g.Wp, g.P, g.pe, g.pe,
                     (long)n_xp);
  hipLaunchKernelGGL(pack_taps_kernel, dim3(cdiv((long)n_wp, 256)), dim3(256), 0, st, w, wp, Cout, Cin, transposed);
  int rc = check_launch("conv4x4s1 embed/pack");
  if (rc) return rc;
  GemmP p = {};
  p.A = wp; p.B = xp; p.C = yq;
  p.M = M; p.N = NB * g.Q; p.K = 16 * Cc; p.k_per_split = p.K;
  p.a_hw = p.K; p.a_img = 0; p.a_ld = p.K;
  p.b_hw = g.Q; p.b_img = (long)Cc * g.P; p.b_ld = g.P;
  p.c_hw = g.Q; p.c_img = (long)M * g.Q; p.c_ld = g.Q;
  p.a_vec = p.b_vec = p.c_vec = 1;
  p.Chi = Cc; p.Wlo = g.Wp;
  rc = launch_gemm<A_KCONTIG, B_TAPN, E_BATCHED>(p, 1, st, "conv4x4s1_fwd");
  if (rc) return rc;
  const long n_y = (long)NB * M * g.Ho * g.Wo;
  hipLaunchKernelGGL(extract_kernel, dim3(ew_grid(n_y)), dim3(256), 0, st, yq, y, g.Ho, g.Wo, g.Wp, g.Q, n_y);
  return check_launch("conv4x4s1 extract");
}

int wfae_conv4x4s1_bwd_weight(const float* dy, const float* x, float* dw, int NB, int Cin, int Cout, int H, int W,
                              int pad, int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(dy && x && dw, WFAE_ERR_NULL_POINTER, "conv4x4s1_bwd_weight: null pointer");
  WFAE_REQUIRE(NB > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && pad >= 0 && pad <= 3, WFAE_ERR_BAD_SHAPE,
               "conv4x4s1_bwd_weight: bad shape");
  const FlatGeom g = flat_geom(H, W, pad);
  WFAE_REQUIRE(g.Ho > 0 && g.Wo > 0, WFAE_ERR_BAD_SHAPE, "conv4x4s1_bwd_weight: empty output");
  WFAE_REQUIRE((int64_t)NB * g.Q < (1ll << 31) && (int64_t)Cin * g.P < (1ll << 31), WFAE_ERR_BAD_SHAPE,
               "conv4x4s1_bwd_weight: too large");
  const size_t n_xp = (size_t)NB * Cin * g.P, n_dq = (size_t)NB * Cout * g.Q, n_w = (size_t)Cout * Cin * 16;
  const size_t fixed = (n_xp + n_dq + n_w) * sizeof(float);
  WFAE_REQUIRE(ws && ws_bytes >= fixed + n_w * sizeof(float), WFAE_ERR_WORKSPACE,
               "conv4x4s1_bwd_weight: workspace %zu < %zu", ws_bytes, fixed + n_w * sizeof(float));
  float* xp = (float*)ws;
  float* dq = xp + n_xp;
  float* dwp = dq + n_dq;
  float* slabs = dwp + n_w;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(embed_kernel, dim3(ew_grid((long)n_xp)), dim3(256), 0, st, x, xp, H, W, g.Wp, g.P, pad, pad,
                     (long)n_xp);
  hipLaunchKernelGGL(embed_kernel, dim3(ew_grid((long)n_dq)), dim3(256), 0, st, dy, dq, g.Ho, g.Wo, g.Wp, g.Q, 0, 0,
                     (long)n_dq);
  int rc = check_launch("conv4x4s1 embed");
  if (rc) return rc;
  GemmP p = {};
  p.A = dq; p.B = xp; p.C = slabs;
  p.M = Cout; p.N = 16 * Cin; p.K = NB * g.Q;
  p.a_hw = g.Q; p.a_img = (long)Cout * g.Q; p.a_ld = g.Q;
  p.b_hw = g.Q; p.b_img = (long)Cin * g.P; p.b_ld = g.P;
  p.a_vec = p.b_vec = p.c_vec = 1;
  p.Chi = Cin; p.Wlo = g.Wp;
  const int splits = pick_splits(p.M, p.N, p.K, ws_bytes - fixed, &p.k_per_split);
  rc = launch_gemm<A_KCONTIG, B_TAPK, E_SLAB>(p, splits, st, "conv4x4s1_bwd_weight");
  if (rc) return rc;
  rc = splitk_finish(slabs, dwp, nullptr, (long)n_w, p.N, splits, 0, st);
  if (rc) return rc;
  hipLaunchKernelGGL(unpack_taps_kernel, dim3(cdiv((long)n_w, 256)), dim3(256), 0, st, dwp, dw, Cout, Cin,
                     accumulate ? 1 : 0);
  return check_launch("conv4x4s1 unpack");
}

// ---- Winograd forms of the three 4x4 stride-2 operations (transforms in wino.hip) ---------------------
namespace {
struct WinoGeom {
  long T;      // tiles
  int K4;      // 4 * Chi
  int NX;      // transform positions: 9 (F(2x2,2x2)) or 25 (F(4x4,2x2))
  size_t nU, nV, nM;
};
inline bool wino_geom(int variant, int NB, int Chi, int Clo, int Hlo, int Wlo, WinoGeom* g) {
  if (variant != 0 && variant != 1) return false;
  if (NB <= 0 || Chi <= 0 || Clo <= 0 || Hlo <= 0 || Wlo <= 0) return false;
  const int M = variant ? 4 : 2;
  if ((Hlo % M) || (Wlo % M)) return false;
  g->T = (long)NB * (Hlo / M) * (Wlo / M);
  // 16-byte rows for the all-vector GEMM kernels
  if (g->T % 4 != 0 || g->T >= (1ll << 31) || Chi % 4 != 0 || Clo % 4 != 0) return false;
  if (NB > 65535 || Chi > 65535 || Clo > 65535) return false;
  g->K4 = 4 * Chi;
  g->NX = variant ? 25 : 9;
  g->nU = (size_t)g->NX * Clo * g->K4;
  g->nV = (size_t)g->NX * g->K4 * g->T;
  g->nM = (size_t)g->NX * Clo * g->T;
  return true;
}
}  // namespace

int wfae_wino_sizes(int variant, int NB, int Chi, int Clo, int Hlo, int Wlo, int64_t* out4) {
  WFAE_REQUIRE(out4, WFAE_ERR_NULL_POINTER, "wino_sizes: null pointer");
  WinoGeom g;
  if (!wino_geom(variant, NB, Chi, Clo, Hlo, Wlo, &g)) return WFAE_ERR_UNSUPPORTED;  // a query, not a failure
  out4[0] = g.T; out4[1] = (int64_t)g.nU; out4[2] = (int64_t)g.nV; out4[3] = (int64_t)g.nM;
  return WFAE_OK;
}

#define WFAE_WINO_TILE_CHECK(what, C_)                                                                          \
  WFAE_REQUIRE((variant == 0 || variant == 1) && NB > 0 && NB <= 65535 && (C_) > 0 && (C_) <= 65535 && Hlo > 0 && \
                   Wlo > 0 && Hlo % (variant ? 4 : 2) == 0 && Wlo % (variant ? 4 : 2) == 0,                       \
               WFAE_ERR_BAD_SHAPE, what ": bad shape")

int wfae_wino_weights(int variant, const float* w, float* U, int Chi, int Clo, wfae_stream_t stream) {
  WFAE_REQUIRE(w && U, WFAE_ERR_NULL_POINTER, "wino_weights: null pointer");
  WFAE_REQUIRE((variant == 0 || variant == 1) && Chi > 0 && Clo > 0, WFAE_ERR_BAD_SHAPE, "wino_weights: bad shape");
  return wino_weights(variant, w, U, Clo, Chi, (hipStream_t)stream);
}

int wfae_wino_in(int variant, const float* hi, float* V, int NB, int Chi, int Hlo, int Wlo, wfae_stream_t stream) {
  WFAE_REQUIRE(hi && V, WFAE_ERR_NULL_POINTER, "wino_in: null pointer");
  WFAE_WINO_TILE_CHECK("wino_in", Chi);
  return wino_in(variant, hi, V, NB, Chi, Hlo, Wlo, (hipStream_t)stream);
}

int wfae_wino_out_t(int variant, const float* lo, float* Mt, int NB, int Clo, int Hlo, int Wlo, wfae_stream_t stream) {
  WFAE_REQUIRE(lo && Mt, WFAE_ERR_NULL_POINTER, "wino_out_t: null pointer");
  WFAE_WINO_TILE_CHECK("wino_out_t", Clo);
  return wino_out_t(variant, lo, Mt, NB, Clo, Hlo, Wlo, (hipStream_t)stream);
}

int wfae_wino_out(int variant, const float* M, float* lo, int NB, int Clo, int Hlo, int Wlo, wfae_stream_t stream) {
  WFAE_REQUIRE(M && lo, WFAE_ERR_NULL_POINTER, "wino_out: null pointer");
  WFAE_WINO_TILE_CHECK("wino_out", Clo);
  return wino_out(variant, M, lo, NB, Clo, Hlo, Wlo, (hipStream_t)stream);
}

int wfae_wino_out_stats(int variant, const float* M, float* lo, int NB, int Clo, int Hlo, int Wlo, double* part,
                        int64_t part_capacity, int* splits_out, wfae_stream_t stream) {
  WFAE_REQUIRE(M && lo && part && splits_out, WFAE_ERR_NULL_POINTER, "wino_out_stats: null pointer");
  WFAE_WINO_TILE_CHECK("wino_out_stats", Clo);
  const int splits = wino_out_stat_splits(variant, NB, Hlo, Wlo);
  WFAE_REQUIRE(part_capacity >= (int64_t)splits * Clo * 2, WFAE_ERR_WORKSPACE, "wino_out_stats: part holds %lld doubles, needs %lld",
               (long long)part_capacity, (long long)splits * Clo * 2);
  *splits_out = splits;
  return wino_out_stats(variant, M, lo, NB, Clo, Hlo, Wlo, part, (hipStream_t)stream);
}

int wfae_wino_in_t(int variant, const float* dV, float* hi, int NB, int Chi, int Hlo, int Wlo, wfae_stream_t stream) {
  WFAE_REQUIRE(dV && hi, WFAE_ERR_NULL_POINTER, "wino_in_t: null pointer");
  WFAE_WINO_TILE_CHECK("wino_in_t", Chi);
  return wino_in_t(variant, dV, hi, NB, Chi, Hlo, Wlo, (hipStream_t)stream);
}

int wfae_wino_in_t_stats(int variant, const float* dV, float* hi, int NB, int Chi, int Hlo, int Wlo, double* part,
                         int64_t part_capacity, int* splits_out, wfae_stream_t stream) {
  WFAE_REQUIRE(dV && hi && part && splits_out, WFAE_ERR_NULL_POINTER, "wino_in_t_stats: null pointer");
  WFAE_WINO_TILE_CHECK("wino_in_t_stats", Chi);
  const int splits = wino_out_stat_splits(variant, NB, Hlo, Wlo);
  WFAE_REQUIRE(part_capacity >= (int64_t)splits * Chi * 2, WFAE_ERR_WORKSPACE, "wino_in_t_stats: part holds %lld doubles, needs %lld",
               (long long)part_capacity, (long long)splits * Chi * 2);
  *splits_out = splits;
  return wino_in_t_stats(variant, dV, hi, NB, Chi, Hlo, Wlo, part, (hipStream_t)stream);
}

int wfae_wino_gemm_down(int variant, const float* U, const float* V, float* M, int NB, int Chi, int Clo, int Hlo, int Wlo,
                        wfae_stream_t stream) {
  WFAE_REQUIRE(U && V && M, WFAE_ERR_NULL_POINTER, "wino_gemm_down: null pointer");
  WinoGeom g;
  WFAE_REQUIRE(wino_geom(variant, NB, Chi, Clo, Hlo, Wlo, &g), WFAE_ERR_UNSUPPORTED,
               "wino_gemm_down: needs Hlo, Wlo divisible by the tile, channels %% 4 == 0 and a tile count divisible by 4");
  hipStream_t st = (hipStream_t)stream;
  GemmP p = {};  // M_xi (Clo x T) = U_xi (Clo x 4Chi) * V_xi (4Chi x T)
  p.A = U; p.B = V; p.C = M;
  p.M = Clo; p.N = (int)g.T; p.K = g.K4; p.k_per_split = p.K;
  p.a_hw = p.K; p.a_img = 0; p.a_ld = p.K;
  p.b_hw = p.N; p.b_img = 0; p.b_ld = p.N;
  p.c_hw = p.N; p.c_img = 0; p.c_ld = p.N;
  p.a_y = (long)Clo * g.K4; p.b_y = (long)g.K4 * g.T; p.c_y = (long)Clo * g.T;
  p.a_vec = aligned16(U); p.b_vec = aligned16(V); p.c_vec = aligned16(M);
  p.big_ok = 1;
  return launch_gemm<A_KCONTIG, B_NCONTIG, E_BATCHED>(p, 1, st, "wino_gemm_down", g.NX);
}

int wfae_wino_gemm_up(int variant, const float* U, const float* Mt, float* dV, int NB, int Chi, int Clo, int Hlo, int Wlo,
                      wfae_stream_t stream) {
  WFAE_REQUIRE(U && Mt && dV, WFAE_ERR_NULL_POINTER, "wino_gemm_up: null pointer");
  WinoGeom g;
  WFAE_REQUIRE(wino_geom(variant, NB, Chi, Clo, Hlo, Wlo, &g), WFAE_ERR_UNSUPPORTED,
               "wino_gemm_up: needs Hlo, Wlo divisible by the tile, channels %% 4 == 0 and a tile count divisible by 4");
  hipStream_t st = (hipStream_t)stream;
  GemmP p = {};  // dV_xi (4Chi x T) = U_xi^T (4Chi x Clo) * Mt_xi (Clo x T)
  p.A = U; p.B = Mt; p.C = dV;
  p.M = g.K4; p.N = (int)g.T; p.K = Clo; p.k_per_split = cdiv(Clo, BK) * BK;
  p.a_ld = g.K4;  // A(m = c, k = l) = U[l * 4Chi + c]
  p.b_hw = p.N; p.b_img = 0; p.b_ld = p.N;
  p.c_hw = p.N; p.c_img = 0; p.c_ld = p.N;
  p.a_y = (long)Clo * g.K4; p.b_y = (long)Clo * g.T; p.c_y = (long)g.K4 * g.T;
  p.a_vec = aligned16(U); p.b_vec = aligned16(Mt); p.c_vec = aligned16(dV);
  p.big_ok = 1;
  return launch_gemm<A_MCONTIG, B_NCONTIG, E_BATCHED>(p, 1, st, "wino_gemm_up", g.NX);
}

int wfae_wino_gemm_wgrad(int variant, const float* Mt, const float* V, float* dw, int NB, int Chi, int Clo, int Hlo,
                         int Wlo, int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(Mt && V && dw, WFAE_ERR_NULL_POINTER, "wino_gemm_wgrad: null pointer");
  WinoGeom g;
  WFAE_REQUIRE(wino_geom(variant, NB, Chi, Clo, Hlo, Wlo, &g), WFAE_ERR_UNSUPPORTED,
               "wino_gemm_wgrad: needs Hlo, Wlo divisible by the tile, channels %% 4 == 0 and a tile count divisible by 4");
  const size_t slab = g.nU * sizeof(float);
  WFAE_REQUIRE(ws && ws_bytes >= 2 * slab, WFAE_ERR_WORKSPACE, "wino_gemm_wgrad: workspace %zu < %zu", ws_bytes, 2 * slab);
  hipStream_t st = (hipStream_t)stream;
  float* dU = (float*)ws;
  float* slabs = dU + g.nU;
  GemmP p = {};  // dU_xi (Clo x 4Chi) = Mt_xi (Clo x T) * V_xi^T (T x 4Chi)
  p.A = Mt; p.B = V; p.C = slabs;
  p.M = Clo; p.N = g.K4; p.K = (int)g.T;
  p.a_hw = p.K; p.a_img = 0; p.a_ld = p.K;
  p.b_hw = p.K; p.b_img = 0; p.b_ld = p.K;
  p.a_y = (long)Clo * g.T; p.b_y = (long)g.K4 * g.T;
  p.a_vec = aligned16(Mt); p.b_vec = aligned16(V); p.c_vec = aligned16(slabs);
  p.big_ok = 1;
  // split K so that NX * tiles * splits covers the chip a few times; a slab holds all dU_xi
  const long tiles = (long)cdiv(Clo, Clo >= 256 ? 256 : 128) * cdiv(g.K4, BN) * g.NX;
  const int stages = cdiv(p.K, BK);
  long want = (2048 + tiles - 1) / tiles;
  if (want < 1) want = 1;
  if (want > stages) want = stages;
  while (want > 1 && (size_t)want * slab > ws_bytes - slab) --want;
  p.k_per_split = cdiv(stages, (int)want) * BK;
  const int splits = cdiv(p.K, p.k_per_split);
  int rc = launch_gemm<A_KCONTIG, B_KCONTIG, E_SLAB>(p, splits, st, "wino_gemm_wgrad", g.NX);
  if (rc) return rc;
  rc = splitk_finish(slabs, dU, nullptr, (long)g.nU, g.K4, splits, 0, st);
  if (rc) return rc;
  return wino_weights_t(variant, dU, dw, Clo, Chi, accumulate ? 1 : 0, st);
}

// ---- the same three products on the bf16 matrix pipe with split operands (splitgemm.hip): planes = 3 — the exact fp32
// split, planes = 1 — the h plane alone = bf16-rounded operands (WFAE_PRECISION_BF16's arithmetic) ----
int wfae_wino_split_supported(int variant, int NB, int Chi, int Clo, int Hlo, int Wlo) {
  WinoGeom g;
  if (!wino_geom(variant, NB, Chi, Clo, Hlo, Wlo, &g)) return 0;
  return (g.K4 % 32 == 0 && Clo % 32 == 0 && g.T % 32 == 0) ? 1 : 0;
}

#define WFAE_WINO_SPLIT_GEOM(what)                                                                            \
  WinoGeom g;                                                                                                 \
  WFAE_REQUIRE(planes == 1 || planes == 3, WFAE_ERR_BAD_SHAPE, what ": planes must be 1 or 3");                 \
  WFAE_REQUIRE(wfae_wino_split_supported(variant, NB, Chi, Clo, Hlo, Wlo) && wino_geom(variant, NB, Chi, Clo, Hlo, Wlo, &g), \
               WFAE_ERR_UNSUPPORTED, what ": needs the Winograd geometry with 4 Chi, Clo and the tile count multiples of 32")

int wfae_wino_weights_split(int variant, const float* w, uint16_t* U3, uint16_t* Ut3, int planes, int Chi, int Clo,
                            wfae_stream_t stream) {
  WFAE_REQUIRE(w && U3 && Ut3, WFAE_ERR_NULL_POINTER, "wino_weights_split: null pointer");
  WFAE_REQUIRE((variant == 0 || variant == 1) && Chi > 0 && Clo > 0 && (planes == 1 || planes == 3), WFAE_ERR_BAD_SHAPE,
               "wino_weights_split: bad shape");
  return wino_weights_split(variant, w, U3, Ut3, planes, Clo, Chi, (hipStream_t)stream);
}

int wfae_wino_in_split(int variant, const float* hi, uint16_t* V3, int planes, int NB, int Chi, int Hlo, int Wlo,
                       wfae_stream_t stream) {
  WFAE_REQUIRE(hi && V3, WFAE_ERR_NULL_POINTER, "wino_in_split: null pointer");
  WFAE_WINO_TILE_CHECK("wino_in_split", Chi);
  WFAE_REQUIRE(planes == 1 || planes == 3, WFAE_ERR_BAD_SHAPE, "wino_in_split: planes must be 1 or 3");
  return wino_in_split(variant, hi, V3, planes, NB, Chi, Hlo, Wlo, (hipStream_t)stream);
}

int wfae_wino_out_t_split(int variant, const float* lo, uint16_t* Mt3, int planes, int NB, int Clo, int Hlo, int Wlo,
                          wfae_stream_t stream) {
  WFAE_REQUIRE(lo && Mt3, WFAE_ERR_NULL_POINTER, "wino_out_t_split: null pointer");
  WFAE_WINO_TILE_CHECK("wino_out_t_split", Clo);
  WFAE_REQUIRE(planes == 1 || planes == 3, WFAE_ERR_BAD_SHAPE, "wino_out_t_split: planes must be 1 or 3");
  return wino_out_t_split(variant, lo, Mt3, planes, NB, Clo, Hlo, Wlo, (hipStream_t)stream);
}

// bf16-stored tensors on the tensor side of the four transforms (the transform-domain operands / products are unchanged)
int wfae_wino_in_split_bf16(int variant, const uint16_t* hi, uint16_t* V3, int planes, int NB, int Chi, int Hlo, int Wlo,
                            wfae_stream_t stream) {
  WFAE_REQUIRE(hi && V3, WFAE_ERR_NULL_POINTER, "wino_in_split_bf16: null pointer");
  WFAE_WINO_TILE_CHECK("wino_in_split_bf16", Chi);
  WFAE_REQUIRE(planes == 1 || planes == 3, WFAE_ERR_BAD_SHAPE, "wino_in_split_bf16: planes must be 1 or 3");
  return wino_in_split(variant, hi, V3, planes, NB, Chi, Hlo, Wlo, (hipStream_t)stream);
}
int wfae_wino_out_t_split_bf16(int variant, const uint16_t* lo, uint16_t* Mt3, int planes, int NB, int Clo, int Hlo, int Wlo,
                               wfae_stream_t stream) {
  WFAE_REQUIRE(lo && Mt3, WFAE_ERR_NULL_POINTER, "wino_out_t_split_bf16: null pointer");
  WFAE_WINO_TILE_CHECK("wino_out_t_split_bf16", Clo);
  WFAE_REQUIRE(planes == 1 || planes == 3, WFAE_ERR_BAD_SHAPE, "wino_out_t_split_bf16: planes must be 1 or 3");
  return wino_out_t_split(variant, lo, Mt3, planes, NB, Clo, Hlo, Wlo, (hipStream_t)stream);
}
/* part null: plain transform; else also the BatchNorm sums of the ROUNDED result (part_capacity / splits_out as the fp32 forms) */
int wfae_wino_out_bf16(int variant, const float* M, uint16_t* lo, int NB, int Clo, int Hlo, int Wlo, double* part,
                       int64_t part_capacity, int* splits_out, wfae_stream_t stream) {
  WFAE_REQUIRE(M && lo && (part != nullptr) == (splits_out != nullptr), WFAE_ERR_NULL_POINTER, "wino_out_bf16: null pointer");
  WFAE_WINO_TILE_CHECK("wino_out_bf16", Clo);
  if (!part) return wino_out(variant, M, lo, NB, Clo, Hlo, Wlo, (hipStream_t)stream);
  const int splits = wino_out_stat_splits(variant, NB, Hlo, Wlo);
  WFAE_REQUIRE(part_capacity >= (int64_t)splits * Clo * 2, WFAE_ERR_WORKSPACE, "wino_out_bf16: part holds %lld doubles, needs %lld",
               (long long)part_capacity, (long long)splits * Clo * 2);
  *splits_out = splits;
  return wino_out_stats(variant, M, lo, NB, Clo, Hlo, Wlo, part, (hipStream_t)stream);
}
int wfae_wino_in_t_bf16(int variant, const float* dV, uint16_t* hi, int NB, int Chi, int Hlo, int Wlo, double* part,
                        int64_t part_capacity, int* splits_out, wfae_stream_t stream) {
  WFAE_REQUIRE(dV && hi && (part != nullptr) == (splits_out != nullptr), WFAE_ERR_NULL_POINTER, "wino_in_t_bf16: null pointer");
  WFAE_WINO_TILE_CHECK("wino_in_t_bf16", Chi);
  if (!part) return wino_in_t(variant, dV, hi, NB, Chi, Hlo, Wlo, (hipStream_t)stream);
  const int splits = wino_out_stat_splits(variant, NB, Hlo, Wlo);
  WFAE_REQUIRE(part_capacity >= (int64_t)splits * Chi * 2, WFAE_ERR_WORKSPACE, "wino_in_t_bf16: part holds %lld doubles, needs %lld",
               (long long)part_capacity, (long long)splits * Chi * 2);
  *splits_out = splits;
  return wino_in_t_stats(variant, dV, hi, NB, Chi, Hlo, Wlo, part, (hipStream_t)stream);
}

int wfae_wino_gemm_down_split(int variant, const uint16_t* U3, const uint16_t* V3, float* M, int planes, int NB, int Chi, int Clo,
                              int Hlo, int Wlo, wfae_stream_t stream) {
  WFAE_REQUIRE(U3 && V3 && M, WFAE_ERR_NULL_POINTER, "wino_gemm_down_split: null pointer");
  WFAE_WINO_SPLIT_GEOM("wino_gemm_down_split");
  // M_xi (Clo x T) = U_xi (Clo x 4Chi) * V_xi (4Chi x T)
  return split_gemm(0, planes, U3, V3, M, Clo, (int)g.T, g.K4, (long)g.nU, (long)g.nV, (long)Clo * g.K4, (long)g.K4 * g.T,
                    (long)Clo * g.T, g.NX, g.K4, 1, 0, (hipStream_t)stream, "wino_gemm_down_split");
}

int wfae_wino_gemm_up_split(int variant, const uint16_t* Ut3, const uint16_t* Mt3, float* dV, int planes, int NB, int Chi, int Clo,
                            int Hlo, int Wlo, wfae_stream_t stream) {
  WFAE_REQUIRE(Ut3 && Mt3 && dV, WFAE_ERR_NULL_POINTER, "wino_gemm_up_split: null pointer");
  WFAE_WINO_SPLIT_GEOM("wino_gemm_up_split");
  // dV_xi (4Chi x T) = U_xi^T (4Chi x Clo) * Mt_xi (Clo x T)
  return split_gemm(0, planes, Ut3, Mt3, dV, g.K4, (int)g.T, Clo, (long)g.nU, (long)g.nM, (long)Clo * g.K4, (long)Clo * g.T,
                    (long)g.K4 * g.T, g.NX, Clo, 1, 0, (hipStream_t)stream, "wino_gemm_up_split");
}

int wfae_wino_gemm_wgrad_split(int variant, const uint16_t* Mt3, const uint16_t* V3, float* dw, int planes, int NB, int Chi,
                               int Clo, int Hlo, int Wlo, int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(Mt3 && V3 && dw, WFAE_ERR_NULL_POINTER, "wino_gemm_wgrad_split: null pointer");
  WFAE_WINO_SPLIT_GEOM("wino_gemm_wgrad_split");
  const size_t slab = g.nU * sizeof(float);
  WFAE_REQUIRE(ws && ws_bytes >= 2 * slab, WFAE_ERR_WORKSPACE, "wino_gemm_wgrad_split: workspace %zu < %zu", ws_bytes, 2 * slab);
  hipStream_t st = (hipStream_t)stream;
  float* dU = (float*)ws;
  float* slabs = dU + g.nU;
  // dU_xi (Clo x 4Chi) = Mt_xi (Clo x T) * V_xi^T (T x 4Chi): both operands K(= T)-contiguous; split K so that the
  // 256 x 128 tiles cover the chip about four times
  const long tiles = (long)cdiv(Clo, 256) * cdiv(g.K4, 128) * g.NX;
  const int steps = (int)(g.T / 32);
  long want = (1024 + tiles - 1) / tiles;
  if (want < 1) want = 1;
  if (want > steps) want = steps;
  while (want > 1 && (size_t)want * slab > ws_bytes - slab) --want;
  const int k_per_split = cdiv(steps, (int)want) * 32;
  const int splits = cdiv(g.T, k_per_split);
  int rc = split_gemm(1, planes, Mt3, V3, slabs, Clo, g.K4, (int)g.T, (long)g.nM, (long)g.nV, (long)Clo * g.T, (long)g.K4 * g.T,
                      (long)Clo * g.K4, g.NX, k_per_split, splits, (long)g.nU, st, "wino_gemm_wgrad_split");
  if (rc) return rc;
  rc = splitk_finish(slabs, dU, nullptr, (long)g.nU, g.K4, splits, 0, st);
  if (rc) return rc;
  return wino_weights_t(variant, dU, dw, Clo, Chi, accumulate ? 1 : 0, st);
}

}  // extern "C"
