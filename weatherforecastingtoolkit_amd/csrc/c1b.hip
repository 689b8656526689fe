// c1b.hip — the Bottleneck's 1x1 convolutions (pipeline/models/ae_64x8x8_lin.py:15,19) forward and data gradient in the
// bf16-STORAGE mode ('medium' precision, BASELINE config 5): activations in HBM as bf16, weights as one bf16 plane, fp32
// accumulation.
//   Y[img][m][p] (bf16) = sum_k W[m][k] f(X[img][k][p]) (+ res[img][m][p])      m < M, k < K, p < HW
// gemm.hip serves the same products through its fp32 LDS images (8-byte global accesses, a bf16 -> fp32 -> bf16 round trip
// per operand value): 2.2 - 3.0 TB/s on tensors that are half the size (profiles/r03_v2_*).  Here a bf16 value never changes
// format between HBM and the matrix core:
//   * X is read as 16-byte pieces (8 bf16 of one k-row) and stored as it is into the k-row LDS image that
//     ds_read_b64_tr_b16 transposes into MFMA B fragments (the image of splitgemm.hip / c1gemm.hip); with the fused
//     BatchNorm + GELU prologue (f = GELU(x * bn_scale[k] + bn_shift[k])) the eight values are widened, activated in fp32 and
//     rounded once — the value gemm.hip's prologue feeds the matrix core;
//   * W (one bf16 plane, written once per step by c1gemm_split_weights) is staged in 64-byte rows read with ds_read_b128;
//   * one v_mfma_f32_32x32x16_bf16 per 32 x 32 x 16 tile product: the kernel is HBM-bound at every shape of the model, so
//     it is built for bytes in flight, not for the matrix pipe: 4 waves per block, 40 - 48 KiB of LDS, 3 - 4 blocks per CU;
//   * the result leaves through wave-private LDS slices as 16-byte rows of 8 bf16 (+ residual read the same way), with the
//     BatchNorm sums of the ROUNDED result reduced on the way (fp64 partial rows, the format of wfae_conv1x1_fwd_stats).
// Block tiles 128 x 128 (2 x 2 waves of 64 x 64), 64 x 256 and 32 x 256 (1 x 4 waves), K-step 32, two LDS stages,
// register-staged global loads one K-step ahead.
#include "common.h"

using namespace wfae;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct C1BP {
  const bf16_t* A;           // weight plane [M][K]
  const bf16_t* B;           // activations [NB][K][HW]
  bf16_t* C;                 // result [NB][M][HW]
  const bf16_t* res;         // residual [NB][M][HW] or null
  const float* pro_scale;    // PRO: folded BatchNorm scale / shift of the input channels [K]
  const float* pro_shift;
  double* part0;             // BatchNorm sums of the result: [rows][M], rows = WN * ntiles (one per 64-column wave tile); null = off
  double* part1;             // sums of squares
  int M, K, HW;
  long N;                    // NB * HW
  int mtiles;
};

constexpr int BNT = 256, BBK = 32;

__device__ __forceinline__ unsigned off_row(int r, int c) { return (unsigned)(r * 64 + ((c ^ ((r >> 2) & 3)) << 4)); }
__device__ __forceinline__ unsigned sw_tr(int k) { return (unsigned)(((k & 3) << 2) | ((k >> 2) & 3)); }

// inclusive prefix over 8 consecutive lanes (lanes 8r .. 8r + 7 of a 16-lane DPP row): lanes 7, 15, 23, ... end up with the
// sums of their group of eight
#define C1B_DPP_F64(v, CTRL)                                                                                       \
  __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true),                         \
                   __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true))
__device__ __forceinline__ double sum8(double v) {
  v += C1B_DPP_F64(v, 0x111);   // row_shr:1 (lanes shifted in from outside the row read 0)
  v += C1B_DPP_F64(v, 0x112);
  v += C1B_DPP_F64(v, 0x114);
  return v;
}

template <int TM, int WM, int WN, int PRO>
__global__ __launch_bounds__(BNT) void c1b_kernel(C1BP p) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int BM = 32 * TM * WM, BN = 64 * WN;
  constexpr int A_B = BM * 64, B_ROW_B = BN * 2, B_B = 32 * B_ROW_B, STAGE_B = A_B + B_B;
  constexpr int A_CH = BM * 4;                              // 16-byte chunks of the A tile per K-step
  constexpr int A_IT = A_CH >= BNT ? A_CH / BNT : 1;
  constexpr bool A_ALL = A_CH >= BNT;
  constexpr int CPR = BN / 8, RPP = BNT / CPR, B_IT = BBK / RPP;   // chunks per k-row, k-rows per pass, passes
  constexpr int SLICE_B = 8 * TM * 64 * 4;                  // one epilogue slice of a wave: 8 TM rows x 64 columns fp32
  static_assert(4 * SLICE_B <= 2 * STAGE_B, "epilogue slices fit the operand stages");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_B];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);   // one contiguous run of tiles per XCD
  const int m0 = (bid % p.mtiles) * BM;
  const int nt = bid / p.mtiles;
  const long n0 = (long)nt * BN;
  const int nsteps = p.K / BBK;

  // ---- loaders
  const int ac = t & 3, ar = t >> 2;
  const bf16_t* a_src[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) a_src[i] = p.A + (long)min(m0 + ar + 64 * i, p.M - 1) * p.K + ac * 8;
  const unsigned a_dst = off_row(ar, ac);     // row ar + 64 i: same swizzle (64 / 4 = 0 mod 4)
  const bool a_on = A_ALL || t < A_CH;
  const int cq = t % CPR, kq = t / CPR;
  const bf16_t* b_src;
  {
    long n = n0 + 8 * cq;
    if (n >= p.N) n = p.N - 8;   // clamped columns are computed and never stored
    const long img = n / p.HW;
    b_src = p.B + img * (long)p.K * p.HW + (n - img * p.HW) + (long)kq * p.HW;
  }
  const long b_row = (long)RPP * p.HW;
  int pro_k = kq;
  u32x4 ra[A_IT], rb[B_IT];
  float ps[B_IT], ph[B_IT];
  auto load_global = [&]() {
    if (a_on) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) ra[i] = *reinterpret_cast<const u32x4*>(a_src[i]);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      rb[i] = *reinterpret_cast<const u32x4*>(b_src + i * b_row);
      if constexpr (PRO) {
        ps[i] = p.pro_scale[pro_k + RPP * i];
        ph[i] = p.pro_shift[pro_k + RPP * i];
      }
    }
  };
  auto advance = [&](bool more) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) a_src[i] += more ? BBK : 0;
    b_src += more ? (long)BBK * p.HW : 0;
    pro_k += more ? BBK : 0;
  };
  auto act2 = [&](unsigned w, float s, float h) -> unsigned {   // bn_act_fwd_kernel<GELU>'s arithmetic on a bf16 pair
    return pack_bf16(gelu_f(fmaf(bf16_lo(w), s, h)), gelu_f(fmaf(bf16_hi(w), s, h)));
  };
  auto store_lds = [&](int buf) {
    unsigned char* s = smem + buf * STAGE_B;
    if (a_on) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) *reinterpret_cast<u32x4*>(s + a_dst + i * (64 * 64)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      u32x4 v = rb[i];
      if constexpr (PRO) {
        v.x = act2(v.x, ps[i], ph[i]);
        v.y = act2(v.y, ps[i], ph[i]);
        v.z = act2(v.z, ps[i], ph[i]);
        v.w = act2(v.w, ps[i], ph[i]);
      }
      const int k = kq + RPP * i;
      *reinterpret_cast<u32x4*>(s + A_B + (unsigned)(k * B_ROW_B) + (((unsigned)cq ^ sw_tr(k)) << 4)) = v;
    }
  };

  // ---- fragments (the maps of c1gemm.hip with one plane)
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

  auto compute = [&](int buf) {
    const unsigned char* s = smem + buf * STAGE_B;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[TM], fb[2];
      const unsigned a_c = (unsigned)(((2 * ks + lh) ^ a_x) << 4);
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(s + a_rd + i * (32 * 64) + a_c);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        s16x4 part[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int row = 16 * ks + 8 * (g >> 1) + 4 * hf + tq;
          const unsigned ch = (unsigned)(((wn0 + 32 * j) >> 3) + 2 * (g & 1) + (tp >> 1));
          const unsigned off = (unsigned)(row * B_ROW_B) + ((ch ^ sw_tr(row)) << 4) + 8u * (tp & 1);
          part[hf] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(s + A_B + off));
        }
        const s16x8 v = __builtin_shufflevector(part[0], part[1], 0, 1, 2, 3, 4, 5, 6, 7);
        fb[j] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  };

  // stage st is multiplied while K-step st + 1 sits in registers / moves into the other stage and K-step st + 2 is in flight
  load_global();
  advance(nsteps > 1);
  store_lds(0);
  load_global();   // K-step 1 (or 0 again when there is only one: stored to the idle stage, never read)
  advance(nsteps > 2);
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const int cur = st & 1;
    store_lds(cur ^ 1);   // K-step st + 1; in the last iteration a stale copy nobody reads
    load_global();        // K-step st + 2
    advance(st + 3 < nsteps);
    compute(cur);
    __syncthreads();
  }

  // ---- epilogue.  Accumulator register q of lane (r31, lh) is C[(q & 3) + 8 (q >> 2) + 4 lh][r31] of its 32 x 32 tile.
  // Slice gq = rows [8 gq, 8 gq + 8) of every 32-row tile of this wave: 8 TM rows x 64 columns, staged in the wave's own
  // 2 - 4 KiB and read back as 8 lanes x 8 columns per row (16-byte bf16 stores, 128 bytes per row segment).
  float* sw = reinterpret_cast<float*>(smem + wave * SLICE_B);
  const int er = lane >> 3, ec = (lane & 7) * 8;       // row of the pass, first column of this lane
  const long nn = n0 + wn0 + ec;
  const bool col_ok = nn < p.N;                         // N % 8 == 0: a chunk is inside or outside as a whole
  long c_base = 0;
  {
    const long n = col_ok ? nn : 0;
    const long img = n / p.HW;
    c_base = img * (long)p.M * p.HW + (n - img * p.HW);
  }
  const long prow = (long)nt * WN + (wave % WN);
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) sw[(i * 8 + r4 + 4 * lh) * 64 + 32 * j + r31] = acc[i][j][4 * gq + r4];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ps_ = 0; ps_ < TM; ++ps_) {
      const int lr = ps_ * 8 + er;                      // local row of the slice: tile ps_, row er of its 8
      const int m = m0 + wm0 + 32 * ps_ + 8 * gq + er;
      const bool ok = col_ok && m < p.M;
      const long off = c_base + (long)(m < p.M ? m : p.M - 1) * p.HW;
      float v[8];
      const float4 v0 = *reinterpret_cast<const float4*>(sw + lr * 64 + ec);
      const float4 v1 = *reinterpret_cast<const float4*>(sw + lr * 64 + ec + 4);
      v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
      if (p.res) {
        float rv[8];
        ldv(p.res + (ok ? off : 0), rv);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += rv[q];
      }
      const u32x4 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
      if (ok) *reinterpret_cast<u32x4*>(p.C + off) = o;
      if (p.part0) {   // sums of the ROUNDED values: fp32 sums of four, fp64 across the row (chan_reduce_kernel's arithmetic)
        const float z = ok ? 1.f : 0.f;
        const float a0 = bf16_lo(o.x), a1 = bf16_hi(o.x), a2 = bf16_lo(o.y), a3 = bf16_hi(o.y);
        const float b0 = bf16_lo(o.z), b1 = bf16_hi(o.z), b2 = bf16_lo(o.w), b3 = bf16_hi(o.w);
        double s1 = (double)(z * ((a0 + a1) + (a2 + a3))) + (double)(z * ((b0 + b1) + (b2 + b3)));
        double s2 = (double)(z * (fmaf(a0, a0, a1 * a1) + fmaf(a2, a2, a3 * a3))) +
                    (double)(z * (fmaf(b0, b0, b1 * b1) + fmaf(b2, b2, b3 * b3)));
        s1 = sum8(s1);
        s2 = sum8(s2);
        if ((lane & 7) == 7 && m < p.M) {
          p.part0[prow * p.M + m] = s1;
          p.part1[prow * p.M + m] = s2;
        }
      }
    }
  }
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// tile choice: 0 = 128 x 128, 1 = 64 x 256, 2 = 32 x 256; -1 = not served
inline int pick_tile(int M, int K, int HW) {
  if (M < 32 || M % 32 != 0 || K % BBK != 0 || K < BBK || HW % 8 != 0) return -1;
  if (M % 128 == 0) return 0;
  if (M % 64 == 0) return 1;
  return 2;
}

template <int PRO>
int launch_c1b(C1BP& p, int tile, hipStream_t st, const char* what) {
  const int bm = tile == 0 ? 128 : (tile == 1 ? 64 : 32), bn = tile == 0 ? 128 : 256;
  p.mtiles = cdiv(p.M, bm);
  const long tiles = (long)p.mtiles * cdiv(p.N, bn);
  WFAE_REQUIRE(tiles < (1l << 31), WFAE_ERR_BAD_SHAPE, "%s: grid too large", what);
  const dim3 grid((unsigned)tiles), block(BNT);
  if (tile == 0) hipLaunchKernelGGL((c1b_kernel<2, 2, 2, PRO>), grid, block, 0, st, p);
  else if (tile == 1) hipLaunchKernelGGL((c1b_kernel<2, 1, 4, PRO>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((c1b_kernel<1, 1, 4, PRO>), grid, block, 0, st, p);
  return check_launch(what);
}

// w [Cout][Cin] fp32 -> Wb [Cout][Cin] and Wtb [Cin][Cout] bf16 (round to nearest even)
__global__ __launch_bounds__(256) void c1b_weights_kernel(const float* __restrict__ w, bf16_t* __restrict__ Wb,
                                                          bf16_t* __restrict__ Wtb, int Cout, int Cin) {
  const long n = (long)Cout * Cin;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bf16_t h = (bf16_t)(pack_bf16(w[i], 0.f) & 0xffffu);
  Wb[i] = h;
  const long co = i / Cin, ci = i - co * Cin;
  Wtb[ci * Cout + co] = h;
}

}  // namespace

extern "C" {

int wfae_c1b_supported(int M, int K, int HW) { return pick_tile(M, K, HW) >= 0 ? 1 : 0; }

int wfae_c1b_stat_rows(int M, int K, int NB, int HW) {
  const int tile = pick_tile(M, K, HW);
  if (tile < 0) return 0;
  return tile == 0 ? 2 * cdiv((int64_t)NB * HW, 128) : 4 * cdiv((int64_t)NB * HW, 256);   // one row per 64-column wave tile
}

int wfae_c1b_weights(const float* w, uint16_t* Wb, uint16_t* Wtb, int Cout, int Cin, wfae_stream_t stream) {
  WFAE_REQUIRE(w && Wb && Wtb, WFAE_ERR_NULL_POINTER, "c1b_weights: null pointer");
  WFAE_REQUIRE(Cout > 0 && Cin > 0, WFAE_ERR_BAD_SHAPE, "c1b_weights: bad shape");
  const long n = (long)Cout * Cin;
  hipLaunchKernelGGL(c1b_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, Wb, Wtb, Cout,
                     Cin);
  return check_launch("c1b_weights");
}

int wfae_c1b_fwd(const uint16_t* Wb, const uint16_t* x, const float* pro_scale, const float* pro_shift, const uint16_t* res,
                 uint16_t* y, int NB, int K, int M, int HW, double* stat_part, int64_t stat_capacity, int* stat_rows,
                 wfae_stream_t stream) {
  WFAE_REQUIRE(Wb && x && y, WFAE_ERR_NULL_POINTER, "c1b_fwd: null pointer");
  WFAE_REQUIRE(NB > 0 && K > 0 && M > 0 && HW > 0 && (int64_t)NB * HW < (1ll << 31), WFAE_ERR_BAD_SHAPE, "c1b_fwd: bad shape");
  const int tile = pick_tile(M, K, HW);
  WFAE_REQUIRE(tile >= 0, WFAE_ERR_UNSUPPORTED, "c1b_fwd: needs M %% 32 == 0, K %% 32 == 0, HW %% 8 == 0 (M %d, K %d, HW %d)", M, K, HW);
  WFAE_REQUIRE(al16(Wb) && al16(x) && al16(y) && (!res || al16(res)), WFAE_ERR_UNSUPPORTED, "c1b_fwd: tensors must be 16-byte aligned");
  WFAE_REQUIRE((pro_scale != nullptr) == (pro_shift != nullptr) && (stat_part != nullptr) == (stat_rows != nullptr),
               WFAE_ERR_NULL_POINTER, "c1b_fwd: scale / shift and stat_part / stat_rows go together");
  C1BP p = {};
  p.A = Wb; p.B = x; p.C = y; p.res = res;
  p.pro_scale = pro_scale; p.pro_shift = pro_shift;
  p.M = M; p.K = K; p.HW = HW; p.N = (long)NB * HW;
  if (stat_part) {
    const int rows = wfae_c1b_stat_rows(M, K, NB, HW);
    WFAE_REQUIRE(stat_capacity >= 2 * (int64_t)rows * M, WFAE_ERR_WORKSPACE, "c1b_fwd: stat_part holds %lld doubles, needs %lld",
                 (long long)stat_capacity, (long long)(2 * (int64_t)rows * M));
    *stat_rows = rows;
    p.part0 = stat_part;
    p.part1 = stat_part + (long)rows * M;
  }
  return pro_scale ? launch_c1b<1>(p, tile, (hipStream_t)stream, "c1b_fwd") : launch_c1b<0>(p, tile, (hipStream_t)stream, "c1b_fwd");
}

}  // extern "C"
