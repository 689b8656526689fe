// splitgemm.hip — fp32-accurate batched GEMMs on the bf16 matrix pipe for the Winograd-domain products.
//
// v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (64 FLOP/clk/SIMD); the bf16 MFMA instructions at 16x that.  An fp32
// value is EXACTLY the sum of three bf16 values  x = h + m + l  (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 3 x 8
// significand bits, every subtraction exact), and a product of two bf16 values is exact in fp32, so
//     a b = ah bh + (ah bm + am bh) + (ah bl + al bh + am bm) + [am bl + al bm + al bl]
// with the bracket below 2^-23 |a b|: six bf16 MFMAs with fp32 accumulation reproduce the fp32 product to about one
// fp32 rounding (tests/test_kernels_gpu.py::test_split_gemm_*: error against fp64 within 1.3x of the fp32-MFMA kernel's)
// at 16/6 = 2.7x the fp32 MFMA rate.  The operand tensors of the Winograd GEMMs are written by this library's own
// transform kernels (wino.hip), which emit the three bf16 planes directly (6 bytes per element instead of 4): the GEMM
// itself does no conversion work.
//
// One kernel, C[y][m][n] (fp32) = sum_k A[y][m][k] B[y](k,n):
//   A: K-contiguous rows [M][K] (three planes `a_plane` elements apart),
//   B: BKIND 0 — [K][N], N contiguous (down: V, up: Mt), read from LDS through ds_read_b64_tr_b16;
//      BKIND 1 — [N][K], K contiguous (weight gradient: the contraction runs over the tiles), read like A.
// Block tile 256 x 128 x 32, 8 waves as 4 (M) x 2 (N), wave tile 64 x 64 = 4 x 4 MFMA tiles of 16 x 16 x 32, two LDS stages
// of 72 KiB, register-staged global loads one K-step ahead, one barrier per K-step.  Per K-step and wave: 96 MFMAs (1536
// cycles of the SIMD's matrix pipe) against 24 (36 with the transposed reads) LDS fragment reads, 9 global loads and 9 LDS
// stores.  LDS images (bank rule of MI355X_MICROARCH.md, LDS): off_row16 / off_tr16 below.
#include "common.h"
#include <stdlib.h>

using namespace wfae;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct SgP {
  const unsigned short* A;
  const unsigned short* B;
  float* C;
  long a_plane, b_plane;  // elements between the h / m / l planes
  long a_y, b_y, c_y;     // batch (blockIdx.y) strides, elements
  long c_split;           // split-K slab stride (blockIdx.z), elements
  int M, N, K;            // K % 32 == 0; BKIND 0: N % 8 == 0
  int lda, ldb, ldc;
  int k_per_split;        // multiple of 32
  int mtiles;
  int tiles, ny;          // work items: tile fastest, then batch (ny of them), then K-split
  int total;              // tiles * ny * splits
};

constexpr int SBM = 256, SBK = 32, SNT = 512;
constexpr int A_PLANE_B = SBM * 64;
// block columns SBN = 64 NTB: 128 (wave tile 64 x 64, the three-plane form: registers) or 256 (wave tile 64 x 128, the
// one-plane form: per K-step a block moves (256 + SBN) x 64 bytes from L2 into LDS for 256 x SBN x 32 MACs, and at one
// product per tile the 128-column block needs 23 TB/s of that traffic chip-wide to keep the matrix pipe fed — 256 columns
// need a third less)
// NP planes per operand: 3 = the exact fp32 split (six products per tile), 1 = the h plane alone = bf16-rounded operands
// with fp32 accumulation (torch's 'medium' matmul precision; one product per tile, operands 2 bytes per element)

// ---- the kernel, on v_mfma_f32_16x16x32_bf16.  Rounds 2's form of it ran v_mfma_f32_32x32x16_bf16 (two 16-deep k-slabs per
// K-step, same tiles and stages).  Under bf16 MFMA load the chip holds its clock by power, and the 16x16x32 shape needs
// less of it per FLOP (MI355X_MICROARCH.md, DVFS give-back (7)): measured on the four layer shapes of the model, same
// run, 28.03 -> 25.53 ms for the three products at an unchanged matrix-pipe share (0.59 -> 0.61) and a clock of 1.88
// instead of 1.74 GHz (profiles/r03_v7_pmc_split_gemm_mfma_shape.txt); step 213.4 -> 208.4 ms on that box.
// Wave tile 64 x 64 = 4 x 4 MFMA tiles of 16 x 16 (64 accumulator registers); a fragment spans the whole 32-deep
// K-step, so the two halves of an iteration are quadrants of the wave tile:
//     (A_lo, B_lo) -> (A_lo, B_hi) -> barrier -> (A_hi, B_hi) -> (A_hi, B_lo)
// and every quadrant shares one operand group (2 tiles x NP planes, 24 registers) with the one before it while the other
// group is read from LDS during the previous quadrant's MFMAs: 24 fragment reads per K-step as before, at most four
// groups live.  B_lo of the next K-step lands in the registers B_hi just left, so the two B groups swap roles every
// iteration (loop unrolled by two).  Row image for the 16-row operand reads: lane l reads row l & 15, 16-byte chunk
// l >> 4; ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32, and the chunk XOR
// g(r >> 2) with g = (0, 2, 3, 1) puts the 16 lanes of every group on 16 different bank quads ((r >> 2) & 3 is 2-way).
__device__ __forceinline__ unsigned off_row16(int r, int c) {
  return (unsigned)(r * 64 + ((c ^ ((0x78 >> (((r >> 2) & 3) << 1)) & 3)) << 4));
}

// k-row image for the transposed reads of this kernel: a 32-lane half reads rows {q, 8 + q} (q = 0..3) x 32 bytes, so
// the chunk XOR needs (k & 3) and bit 3 of k only — without bit 2 the two blocks of a fragment (rows 4 apart) differ
// by an immediate and four address registers serve the eight (tile, block) reads of a group
__device__ __forceinline__ unsigned off_tr16(int k, int ch) {
  return (unsigned)(k * 256 + ((ch ^ (((k & 3) << 2) | ((k >> 2) & 2))) << 4));
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BKIND, int NP = 3, int NTB = 2, bool PERSIST = false>
__global__ __launch_bounds__(SNT, 2) void sgemm3_kernel(SgP p) {
  static_assert(NP == 3 || NP == 1, "three exact planes or the h plane alone");
  static_assert(NTB == 2 || (NTB == 4 && NP == 1), "64 x 128 wave tiles (128 accumulators) only with one plane");
  constexpr int SBN = 64 * NTB, NBL = SBN / 128;   // NBL: 128-column (row) pieces of the B stage a thread loads
  constexpr int B_PLANE_B = SBN * 64;              // either image: SBN rows x 64 B, or NBL x (32 k-rows x 256 B)
  constexpr int A_STAGE_B = NP * A_PLANE_B;
  constexpr int STAGE_B = NP * (A_PLANE_B + B_PLANE_B);
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_B];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // XCD-aware placement; workgroups go round-robin over the 8 XCDs in launch order, each XCD with its own L2.
  // Classic launch (grid = tiles x batches x splits, one tile per workgroup): gridDim.x % 8 == 0: every XCD gets a
  // contiguous run of the tiles of one batch; otherwise (few tiles per batch / K-split: the weight gradients) whole (y, z)
  // groups are dealt to the XCDs, so that the tiles which share the group's operands run side by side under ONE L2.
  // PERSISTENT launch (round 3, three-plane operands; one workgroup per CU when every item has the same even number of
  // K-steps): items (tile, batch, K-split) in the linear order tile-fastest, XCD x owns the contiguous run [x W8, (x + 1) W8)
  // and its gridDim.x / 8 workgroups walk it together (item = x W8 + slot + (gridDim.x / 8) i).  The loader runs two K-steps
  // ahead ACROSS items: while a tile's last two K-steps are multiplied, the first two of the workgroup's next tile travel
  // HBM -> registers -> LDS, and the tile's epilogue is the only gap in the matrix pipe.  Same box, layer shapes of the
  // model (kbench totals): down 8.27 -> 7.95, up 8.85 -> 8.41 ms, weight gradients 9.25 -> 9.23; with one-plane operands
  // (short K loops, L2-bound) the static partition measured 4 % slower and is not used.
  int item, item_end, per;
  int c_bid = 0, c_by = 0, c_bz = 0;   // classic launch: the tile of this workgroup
  if constexpr (PERSIST) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int w8 = (p.total + 7) / 8;
    per = gridDim.x >> 3;
    item = xcd * w8 + slot;
    item_end = min(p.total, (xcd + 1) * w8);
    if (item >= item_end) return;   // whole workgroup: nobody reaches a barrier
  } else {
    c_bid = blockIdx.x; c_by = blockIdx.y; c_bz = blockIdx.z;
    if ((gridDim.x & 7) == 0) {
      c_bid = (c_bid & 7) * (gridDim.x >> 3) + (c_bid >> 3);
    } else {
      const unsigned nxt = gridDim.x, ngroups = gridDim.y * gridDim.z;
      const unsigned L = blockIdx.x + nxt * (blockIdx.y + gridDim.y * blockIdx.z);
      if (L < (ngroups & ~7u) * nxt) {
        const unsigned sl = L >> 3, G = (L & 7) + 8 * (sl / nxt);
        c_bid = (int)(sl % nxt);
        c_by = (int)(G % gridDim.y);
        c_bz = (int)(G / gridDim.y);
      }
    }
    item = 0; item_end = 1; per = 1;
  }

  const int ac = t & 3, ar = t >> 2;
  const unsigned a_dst = off_row16(ar, ac);
  const int bk = t >> 4, bch = t & 15;
  const unsigned b_dst = BKIND == 0 ? off_tr16(bk, bch) : a_dst;   // + h * 32 * 256 (k-row image per 128 columns) / + h * 128 * 64
  const unsigned b_step = BKIND == 0 ? 2u * (unsigned)(SBK * p.ldb) : 2u * SBK;

  // ---- loader cursor: the (item, K-step) the next global loads fetch — up to two K-steps and one item ahead of the multiply
  const char* __restrict__ Apl[NP];
  const char* __restrict__ Bpl[NP];
  unsigned a_off0, a_off1, b_off[NBL];
  int ld_next = item;   // the item the cursor moves to when it has fetched the last K-step of its current one
  int ld_k = 0, ld_nsteps = 0;
  auto tile_of = [&](int it, int& m0_, int& n0_, int& by_i, int& bz_i) {
    int bid;
    if constexpr (PERSIST) {
      const int r = it / p.tiles;
      bid = it - r * p.tiles;
      bz_i = r / p.ny;
      by_i = r - bz_i * p.ny;
    } else {
      bid = c_bid; by_i = c_by; bz_i = c_bz;
    }
    m0_ = (bid % p.mtiles) * SBM;
    n0_ = (bid / p.mtiles) * SBN;
  };
  auto set_cursor = [&](int it) {
    int m0_, n0_, by_i, bz_i;
    tile_of(it, m0_, n0_, by_i, bz_i);
    const int kb = bz_i * p.k_per_split;
    ld_nsteps = (min(p.K, kb + p.k_per_split) - kb) / SBK;
    ld_k = 0;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      Apl[pl] = reinterpret_cast<const char*>(p.A + (long)by_i * p.a_y + pl * p.a_plane);
      Bpl[pl] = reinterpret_cast<const char*>(p.B + (long)by_i * p.b_y + pl * p.b_plane);
    }
    // rows / columns beyond the matrix are clamped to valid ones (their products land where the epilogue does not store)
    a_off0 = 2u * (unsigned)(min(m0_ + ar, p.M - 1) * p.lda + kb + ac * 8);
    a_off1 = 2u * (unsigned)(min(m0_ + ar + 128, p.M - 1) * p.lda + kb + ac * 8);
#pragma unroll
    for (int h = 0; h < NBL; ++h) {
      if constexpr (BKIND == 0) {
        int n = n0_ + 128 * h + bch * 8;
        if (n >= p.N) n = p.N - 8;
        b_off[h] = 2u * (unsigned)((kb + bk) * p.ldb + n);
      } else {
        b_off[h] = 2u * (unsigned)(min(n0_ + ar + 128 * h, p.N - 1) * p.ldb + kb + ac * 8);
      }
    }
  };
  u32x4 ra[NP][2], rb[NP][NBL];
  auto load_global = [&]() {
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      ra[pl][0] = *reinterpret_cast<const u32x4*>(Apl[pl] + a_off0);
      ra[pl][1] = *reinterpret_cast<const u32x4*>(Apl[pl] + a_off1);
#pragma unroll
      for (int h = 0; h < NBL; ++h) rb[pl][h] = *reinterpret_cast<const u32x4*>(Bpl[pl] + b_off[h]);
    }
  };
  // after every load_global: one K-step on; past the item's last one the cursor moves to the workgroup's next item (a
  // uniform branch, once per item) or, when there is none, stays on the last K-step (re-loaded, stored to the idle stage,
  // never read)
  auto advance = [&]() {
    if (__builtin_expect(++ld_k < ld_nsteps, 1)) {
      a_off0 += 2u * SBK;
      a_off1 += 2u * SBK;
#pragma unroll
      for (int h = 0; h < NBL; ++h) b_off[h] += b_step;
    } else if (ld_next < item_end) {
      set_cursor(ld_next);
      ld_next += per;
    } else {
      ld_k = ld_nsteps - 1;
    }
  };
  auto store_lds = [&](int buf) {
    unsigned char* s = smem + buf * STAGE_B;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      *reinterpret_cast<u32x4*>(s + pl * A_PLANE_B + a_dst) = ra[pl][0];
      *reinterpret_cast<u32x4*>(s + pl * A_PLANE_B + a_dst + 128 * 64) = ra[pl][1];
#pragma unroll
      for (int h = 0; h < NBL; ++h) *reinterpret_cast<u32x4*>(s + A_STAGE_B + pl * B_PLANE_B + b_dst + h * 128 * 64) = rb[pl][h];
    }
  };

  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * (SBN / 2);
  const int r15 = lane & 15, g4 = lane >> 4, tq = r15 >> 2, tp = r15 & 3;
  const unsigned a_rd = off_row16(wm0 + r15, g4);   // + (32 half + 16 i) * 64: the swizzle only sees (r15 >> 2) & 3
  const unsigned b_rd_row = off_row16(wn0 + r15, g4);

  f32x4 acc[4][2 * NTB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2 * NTB; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;

  struct Grp {
    bf16x8 v[2][NP];
  };
  struct GrpB {
    bf16x8 v[NTB][NP];
  };
  auto read_a = [&](Grp& f, int buf, int half) {
    const unsigned char* s = smem + buf * STAGE_B;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        f.v[i][pl] = *reinterpret_cast<const bf16x8*>(s + pl * A_PLANE_B + a_rd + (32 * half + 16 * i) * 64);
  };
  auto read_b = [&](GrpB& f, int buf, int half) {
    const unsigned char* s = smem + buf * STAGE_B + A_STAGE_B;
    if constexpr (BKIND == 0) {
      // lane 4q+pp of the 16-lane group g4 addresses k-row q, columns 4pp..4pp+3 of a 4 (k) x 16 (n) block; the group's
      // fragment is k = 8 g4 .. 8 g4 + 7: two blocks
#pragma unroll
      for (int j = 0; j < NTB; ++j)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
          s16x4 part[2];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const int row = 8 * g4 + 4 * hf + tq;
            const int col = wn0 + 16 * NTB * half + 16 * j;   // first column of the tile inside the block
            const int ch = ((col & 127) >> 3) + (tp >> 1);
            const unsigned off = (unsigned)((col >> 7) * 32 * 256) + off_tr16(row, ch) + 8u * (tp & 1);
            part[hf] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3)))*)(s + pl * B_PLANE_B + off));
          }
          const s16x8 v = __builtin_shufflevector(part[0], part[1], 0, 1, 2, 3, 4, 5, 6, 7);
          f.v[j][pl] = __builtin_bit_cast(bf16x8, v);
        }
    } else {
#pragma unroll
      for (int j = 0; j < NTB; ++j)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
          f.v[j][pl] = *reinterpret_cast<const bf16x8*>(s + pl * B_PLANE_B + b_rd_row + (16 * NTB * half + 16 * j) * 64);
    }
  };
  auto quadrant = [&](const Grp& a, const GrpB& b, int ah, int bh) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NTB; ++j) {
        f32x4 c = acc[2 * ah + i][NTB * bh + j];   // smallest terms first
        if constexpr (NP == 3) {
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[i][2], b.v[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[i][0], b.v[j][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[i][1], b.v[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[i][1], b.v[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[i][0], b.v[j][1], c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[i][0], b.v[j][0], c, 0, 0, 0);
        acc[2 * ah + i][NTB * bh + j] = c;
      }
  };

  constexpr int NQ = 4 * (NP == 3 ? 6 : 1);            // MFMAs per quadrant
  constexpr int RB = 2 * NP * (BKIND == 0 ? 2 : 1);    // LDS reads of a B group
  constexpr int RA = 2 * NP;
  Grp A0, A1;
  GrpB B0, B1;
  // one K-step; on entry A0 = A_lo, bx = B_lo of K-step st (stage cur); on exit A0 = A_lo, by_ = B_lo of K-step st + 1
  auto iter = [&](int st, GrpB& bx, GrpB& by_) {
    const int cur = st & 1;
    read_b(by_, cur, 1);
    read_a(A1, cur, 1);
    store_lds(cur ^ 1);   // K-step st + 1 (the last iteration: K-step 0 of the next item, or a stale copy nobody reads)
    load_global();        // K-step st + 2 (the last two iterations: K-steps 0 and 1 of the next item)
    quadrant(A0, bx, 0, 0);
    quadrant(A0, by_, 0, 1);
    if constexpr (NP == 3) {
#pragma unroll
      for (int q = 0; q < 9; ++q) {   // 48 MFMAs, 12 or 18 LDS reads, 9 LDS writes, 9 global loads
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    read_a(A0, cur ^ 1, 0);
    quadrant(A1, by_, 1, 1);
    read_b(by_, cur ^ 1, 0);
    quadrant(A1, bx, 1, 0);
    if constexpr (NP == 3) {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x100, (RA + RB + 5) / 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NQ / 6, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue of the workgroup's first item
  set_cursor(item);
  ld_next = item + per;
  load_global();
  advance();
  store_lds(0);
  load_global();
  advance();
  __syncthreads();
  read_a(A0, 0, 0);
  read_b(B0, 0, 0);

  for (; item < item_end; item += per) {
    int m0, n0, by, bz;
    tile_of(item, m0, n0, by, bz);
    const int k_begin = bz * p.k_per_split;
    const int nsteps = (min(p.K, k_begin + p.k_per_split) - k_begin) / SBK;
    // (persistent launches have an even nsteps: K-step 0 of every item lies in stage 0 and B0 holds its B_lo)
    int st = 0;
    // the cursor moves BETWEEN the iterations: its once-per-item branch must not cut the body of an iteration, whose
    // memory instructions the scheduler spreads between the MFMAs (sched_group_barrier works inside one basic block)
    for (; st + 1 < nsteps; st += 2) {
      iter(st, B0, B1);
      advance();
      iter(st + 1, B1, B0);
      advance();
    }
    if (st < nsteps) {
      iter(st, B0, B1);
      advance();
    }

    // ---- epilogue: accumulator register q of lane (r15, g4) is C[4 g4 + q][r15] of its 16 x 16 tile
    float* __restrict__ Cb = p.C + (long)by * p.c_y + (long)bz * p.c_split;
    if (m0 + SBM <= p.M && n0 + SBN <= p.N) {   // whole tile (the usual case): no per-element bounds, one base + immediates
      float* __restrict__ c0 = Cb + (long)(m0 + wm0 + 4 * g4) * p.ldc + n0 + wn0 + r15;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < 2 * NTB; ++j) c0[(long)(16 * i + q) * p.ldc + 16 * j] = acc[i][j][q];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = m0 + wm0 + 16 * i + 4 * g4 + q;
#pragma unroll
          for (int j = 0; j < 2 * NTB; ++j) {
            const int col = n0 + wn0 + 16 * j + r15;
            if (row < p.M && col < p.N) Cb[(long)row * p.ldc + col] = acc[i][j][q];
          }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2 * NTB; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
  }
}

__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, unsigned short* __restrict__ o, long n, int planes) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned short h, m, l;
  split3(x[i], h, m, l);
  o[i] = h;
  if (planes == 3) {
    o[n + i] = m;
    o[2 * n + i] = l;
  }
}

template <int BKIND>
int launch_sgemm3(SgP& p, int planes, int batches, int splits, hipStream_t st, const char* what) {
  p.mtiles = cdiv(p.M, SBM);
  // one plane: 256-column blocks (round 3, same box: the Winograd products of a 'medium' step 13.5 -> 10.7 ms, step 105.4 -> 100.8)
  const bool wide = planes == 1 && p.N >= 256;
  p.tiles = p.mtiles * cdiv(p.N, wide ? 256 : 128);
  p.ny = batches;
  const long total = (long)p.tiles * batches * splits;
  WFAE_REQUIRE(total < (1l << 30), WFAE_ERR_BAD_SHAPE, "%s: grid too large", what);
  p.total = (int)total;
  // persistent (three planes) when every item has the same even number of K-steps (the stage parity and the role of the
  // two B register groups then repeat from item to item): one workgroup per CU; else one workgroup per tile
  const bool uniform = splits == 1 || p.K % p.k_per_split == 0;
  const int ksteps = (splits == 1 ? p.K : p.k_per_split) / SBK;
  static const int ncu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return (n + 7) / 8 * 8;
  }();
  WFAE_REQUIRE(p.tiles > 0 && batches <= 65535 && splits <= 65535, WFAE_ERR_BAD_SHAPE, "%s: grid too large", what);
  const dim3 classic((unsigned)p.tiles, batches, splits);
  // (An LDS-DMA form of the loaders — global_load_lds_dwordx4, no staging registers or ds_write — measured equal on the
  // Winograd shapes in round 2, 28.96 vs 28.57 ms, and was removed: the operand path is not what the waves wait for.)
  if (wide) hipLaunchKernelGGL((sgemm3_kernel<BKIND, 1, 4>), classic, dim3(SNT), 0, st, p);
  else if (planes == 1) hipLaunchKernelGGL((sgemm3_kernel<BKIND, 1>), classic, dim3(SNT), 0, st, p);
  else if (uniform && ksteps >= 2 && ksteps % 2 == 0 && p.total > ncu)
    hipLaunchKernelGGL((sgemm3_kernel<BKIND, 3, 2, true>), dim3(ncu), dim3(SNT), 0, st, p);
  else hipLaunchKernelGGL((sgemm3_kernel<BKIND, 3>), classic, dim3(SNT), 0, st, p);
  return check_launch(what);
}

}  // namespace

namespace wfae {

// C[y] (M x N) = A[y] (M x K, rows) * B[y]; kind 0: B is K x N (N contiguous), kind 1: B is N x K (K contiguous)
int split_gemm(int kind, int planes, const unsigned short* A, const unsigned short* B, float* C, int M, int N, int K,
               long a_plane, long b_plane, long a_y, long b_y, long c_y, int batches, int k_per_split, int splits,
               long c_split, hipStream_t st, const char* what) {
  WFAE_REQUIRE(planes == 1 || planes == 3, WFAE_ERR_BAD_SHAPE, "%s: planes must be 3 (exact fp32 split) or 1 (bf16 operands)", what);
  WFAE_REQUIRE(K % SBK == 0 && k_per_split % SBK == 0 && (kind == 1 || N % 8 == 0) && M > 0 && N >= 8, WFAE_ERR_UNSUPPORTED,
               "%s: the split GEMM needs K %% 32 == 0 and 16-byte operand rows", what);
  // the loaders address one plane of one batch with 32-bit BYTE offsets (2 * element index, one VGPR per operand stream)
  WFAE_REQUIRE((long)M * K < (1l << 31) - (1l << 16) && (long)K * N < (1l << 31) - (1l << 16), WFAE_ERR_BAD_SHAPE,
               "%s: one operand plane of one batch must stay below 2^31 elements (M %d, N %d, K %d)", what, M, N, K);
  SgP p = {};
  p.A = A; p.B = B; p.C = C;
  p.a_plane = a_plane; p.b_plane = b_plane;
  p.a_y = a_y; p.b_y = b_y; p.c_y = c_y; p.c_split = c_split;
  p.M = M; p.N = N; p.K = K;
  p.lda = K; p.ldb = kind == 0 ? N : K; p.ldc = N;
  p.k_per_split = k_per_split;
  return kind == 0 ? launch_sgemm3<0>(p, planes, batches, splits, st, what) : launch_sgemm3<1>(p, planes, batches, splits, st, what);
}

}  // namespace wfae

extern "C" {

int wfae_split_bf16x3(const float* x, uint16_t* out, int64_t n, int planes, wfae_stream_t stream) {
  WFAE_REQUIRE(x && out, WFAE_ERR_NULL_POINTER, "split_bf16x3: null pointer");
  WFAE_REQUIRE(n > 0 && n < (1ll << 39) && (planes == 1 || planes == 3), WFAE_ERR_BAD_SHAPE, "split_bf16x3: bad size / planes");
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, (long)n, planes);
  return check_launch("split_bf16x3");
}

int wfae_split_gemm(int b_kind, int planes, const uint16_t* A3, const uint16_t* B3, float* C, int M, int N, int K, int batches,
                    wfae_stream_t stream) {
  WFAE_REQUIRE(A3 && B3 && C, WFAE_ERR_NULL_POINTER, "split_gemm: null pointer");
  WFAE_REQUIRE((b_kind == 0 || b_kind == 1) && M > 0 && N > 0 && K > 0 && batches > 0, WFAE_ERR_BAD_SHAPE, "split_gemm: bad shape");
  WFAE_REQUIRE(((reinterpret_cast<uintptr_t>(A3) | reinterpret_cast<uintptr_t>(B3)) & 15) == 0, WFAE_ERR_BAD_SHAPE,
               "split_gemm: operands must be 16-byte aligned");
  const long na = (long)batches * M * K, nb = (long)batches * K * N;
  return split_gemm(b_kind, planes, A3, B3, C, M, N, K, na, nb, (long)M * K, (long)K * N, (long)M * N, batches, K, 1, 0,
                    (hipStream_t)stream, "split_gemm");
}

}  // extern "C"
