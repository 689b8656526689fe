// transformer.hip — the latent transformer of the `_tf` model variant
// (pipeline/models/ae_64x8x8_tf.py:77-80,107-109): 8 post-norm nn.TransformerEncoderLayer
// (d_model 64, 8 heads, ff 2048, ReLU, dropout 0.1) applied SEQ-FIRST to (B, 64 tokens, 64):
// attention runs across the batch dimension (sequence length S = B <= 64), the 64 tokens are
// the "batch".  FLOPs are negligible (~0.3 GFLOP/frame); the kernels favour simplicity:
// one wavefront per (token, head) for attention, one wavefront per row for LayerNorm.
// The four Linear layers reuse the MFMA GEMM (wfae_linear_*).
#include "common.h"

using namespace wfae;

namespace {

// counter-based RNG (splitmix64 finaliser): uniform [0,1) from (seed, index)
__device__ __forceinline__ float rng01(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// ---------------------------------------------------------------- LayerNorm
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                            const float* __restrict__ g, const float* __restrict__ b,
                                                            float* __restrict__ y, float* __restrict__ mean,
                                                            float* __restrict__ rstd, int rows, int E, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xp = x + (long)row * E;
  const float* rp = res ? res + (long)row * E : nullptr;
  float s = 0.f;
  for (int i = lane; i < E; i += 64) s += xp[i] + (rp ? rp[i] : 0.f);
  const float mu = wave_sum(s) / E;
  float v = 0.f;
  for (int i = lane; i < E; i += 64) {
    const float d = xp[i] + (rp ? rp[i] : 0.f) - mu;
    v += d * d;
  }
  const float rs = rsqrtf(wave_sum(v) / E + eps);
  for (int i = lane; i < E; i += 64) y[(long)row * E + i] = (xp[i] + (rp ? rp[i] : 0.f) - mu) * rs * g[i] + b[i];
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// dx (w.r.t. the normalised input h = x + res) and per-block partial dgamma / dbeta
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ res, const float* __restrict__ g,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ dx, float* __restrict__ part, int rows,
                                                            int E, int rows_per_block) {
  extern __shared__ float sm[];  // [4 waves][2][E]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* pg = sm + (wave * 2) * E;
  float* pb = pg + E;
  for (int i = lane; i < E; i += 64) { pg[i] = 0.f; pb[i] = 0.f; }
  const int r0 = blockIdx.x * rows_per_block;
  for (int r = r0 + wave; r < min(rows, r0 + rows_per_block); r += 4) {
    const float mu = mean[r], rs = rstd[r];
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < E; i += 64) {
      const float h = x[(long)r * E + i] + (res ? res[(long)r * E + i] : 0.f);
      const float xh = (h - mu) * rs;
      const float gd = g[i] * dy[(long)r * E + i];
      s1 += gd;
      s2 += gd * xh;
    }
    s1 = wave_sum(s1) / E;
    s2 = wave_sum(s2) / E;
    for (int i = lane; i < E; i += 64) {
      const float h = x[(long)r * E + i] + (res ? res[(long)r * E + i] : 0.f);
      const float xh = (h - mu) * rs;
      const float d = dy[(long)r * E + i];
      dx[(long)r * E + i] = rs * (g[i] * d - s1 - xh * s2);
      pg[i] += d * xh;
      pb[i] += d;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < E; i += 256) {
    float a = 0.f, c = 0.f;
    for (int w = 0; w < 4; ++w) { a += sm[(w * 2) * E + i]; c += sm[(w * 2 + 1) * E + i]; }
    part[((long)blockIdx.x * 2) * E + i] = a;
    part[((long)blockIdx.x * 2 + 1) * E + i] = c;
  }
}

// block = 64 columns x 16 row groups: the per-block partials are added 16-way in parallel, then combined through LDS
// in a fixed order
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(const float* __restrict__ part, int blocks, int E,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               int beta) {
  __shared__ float sa[16][64], sc[16][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + cl;
  float a = 0.f, c = 0.f;
  if (i < E)
    for (int b = rg; b < blocks; b += 16) { a += part[((long)b * 2) * E + i]; c += part[((long)b * 2 + 1) * E + i]; }
  sa[rg][cl] = a;
  sc[rg][cl] = c;
  __syncthreads();
  if (rg == 0 && i < E) {
    float ta = 0.f, tc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ta += sa[r][cl]; tc += sc[r][cl]; }
    dgamma[i] = (beta ? dgamma[i] : 0.f) + ta;
    dbeta[i] = (beta ? dbeta[i] : 0.f) + tc;
  }
}

// ---------------------------------------------------------------- attention
// qkv rows are (s, n) pairs: row = s*rs + n*rn (seq-first: rs = N, rn = 1; batch-first: rs = 1, rn = S),
// columns [q | k | v] each E = H*D wide.  One wavefront per (n, h); lane i < S owns query row i.  S <= 64,
// D in {8, 16, 64}.
template <int D>
__global__ __launch_bounds__(64) void mha_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                     float* __restrict__ probs, int S, int N, int H, float scale,
                                                     float p_drop, unsigned long long seed, long rs, long rn) {
  // rows padded to D + 4 words: 16-byte aligned, so a key / value row is read (all lanes the same address: an LDS
  // broadcast) as D / 4 ds_read_b128 instead of D ds_read_b32 — the loops below are LDS-instruction bound
  static_assert(D % 4 == 0, "head dim");
  __shared__ __attribute__((aligned(16))) float ks[64][D + 4], vs[64][D + 4];
  const int n = blockIdx.x / H, h = blockIdx.x % H;
  const int E = H * D, i = threadIdx.x;
  if (i < S) {
    const float4* row4 = reinterpret_cast<const float4*>(qkv + ((long)i * rs + (long)n * rn) * 3 * E + h * D);
#pragma unroll
    for (int d = 0; d < D / 4; ++d) {
      *reinterpret_cast<float4*>(&ks[i][4 * d]) = row4[E / 4 + d];
      *reinterpret_cast<float4*>(&vs[i][4 * d]) = row4[2 * E / 4 + d];
    }
  }
  __syncthreads();
  if (i >= S) return;
  float q[D];
  const float* row = qkv + ((long)i * rs + (long)n * rn) * 3 * E + h * D;
#pragma unroll
  for (int d = 0; d < D; ++d) q[d] = row[d] * scale;
  float* pr = probs + (((long)n * H + h) * S + i) * S;
  float mx = -INFINITY;
  for (int j = 0; j < S; ++j) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < D / 4; ++d) {
      const float4 k4 = *reinterpret_cast<const float4*>(&ks[j][4 * d]);
      s = fmaf(q[4 * d], k4.x, s);
      s = fmaf(q[4 * d + 1], k4.y, s);
      s = fmaf(q[4 * d + 2], k4.z, s);
      s = fmaf(q[4 * d + 3], k4.w, s);
    }
    pr[j] = s;
    mx = fmaxf(mx, s);
  }
  float den = 0.f;
  for (int j = 0; j < S; ++j) { const float e = expf(pr[j] - mx); pr[j] = e; den += e; }
  const float inv = 1.f / den;
  float o[D];
#pragma unroll
  for (int d = 0; d < D; ++d) o[d] = 0.f;
  const float keep = 1.f / (1.f - p_drop);
  for (int j = 0; j < S; ++j) {
    float pj = pr[j] * inv;
    pr[j] = pj;  // softmax probabilities (before dropout) saved for backward
    if (p_drop > 0.f) pj = rng01(seed, (((unsigned long long)n * H + h) * S + i) * S + j) >= p_drop ? pj * keep : 0.f;
#pragma unroll
    for (int d = 0; d < D / 4; ++d) {
      const float4 v4 = *reinterpret_cast<const float4*>(&vs[j][4 * d]);
      o[4 * d] = fmaf(pj, v4.x, o[4 * d]);
      o[4 * d + 1] = fmaf(pj, v4.y, o[4 * d + 1]);
      o[4 * d + 2] = fmaf(pj, v4.z, o[4 * d + 2]);
      o[4 * d + 3] = fmaf(pj, v4.w, o[4 * d + 3]);
    }
  }
#pragma unroll
  for (int d = 0; d < D; ++d) out[((long)i * rs + (long)n * rn) * E + h * D + d] = o[d];
}

template <int D>
__global__ __launch_bounds__(64) void mha_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ probs,
                                                     const float* __restrict__ dout, float* __restrict__ dqkv, int S,
                                                     int N, int H, float scale, float p_drop, unsigned long long seed,
                                                     long rs, long rn) {
  extern __shared__ float mha_lds[];
  float(*qs)[D + 1] = reinterpret_cast<float(*)[D + 1]>(mha_lds);
  float(*ks)[D + 1] = qs + 64;
  float(*vs)[D + 1] = ks + 64;
  float(*dos)[D + 1] = vs + 64;
  float(*dss)[65] = reinterpret_cast<float(*)[65]>(mha_lds + 4 * 64 * (D + 1));  // dS
  float(*pds)[65] = dss + 64;                                                      // dropped probabilities
  const int n = blockIdx.x / H, h = blockIdx.x % H;
  const int E = H * D, i = threadIdx.x;
  if (i < S) {
    const float* row = qkv + ((long)i * rs + (long)n * rn) * 3 * E + h * D;
    const float* dr = dout + ((long)i * rs + (long)n * rn) * E + h * D;
#pragma unroll
    for (int d = 0; d < D; ++d) { qs[i][d] = row[d]; ks[i][d] = row[E + d]; vs[i][d] = row[2 * E + d]; dos[i][d] = dr[d]; }
  }
  __syncthreads();
  const float keep = 1.f / (1.f - p_drop);
  if (i < S) {
    const float* pr = probs + (((long)n * H + h) * S + i) * S;
    // dP'_ij = dout_i . v_j (w.r.t. dropped probs), dP_ij = mask/keep * dP'_ij, dS = P (dP - sum_j P dP)
    float dot = 0.f;
    for (int j = 0; j < S; ++j) {
      float dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) dp = fmaf(dos[i][d], vs[j][d], dp);
      float m = 1.f;
      if (p_drop > 0.f) m = rng01(seed, (((unsigned long long)n * H + h) * S + i) * S + j) >= p_drop ? keep : 0.f;
      const float pj = pr[j];
      pds[i][j] = pj * m;
      dp *= m;
      dss[i][j] = dp;
      dot = fmaf(pj, dp, dot);
    }
    float dq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) dq[d] = 0.f;
    for (int j = 0; j < S; ++j) {
      const float ds = pr[j] * (dss[i][j] - dot);
      dss[i][j] = ds;
#pragma unroll
      for (int d = 0; d < D; ++d) dq[d] = fmaf(ds, ks[j][d], dq[d]);
    }
    float* o = dqkv + ((long)i * rs + (long)n * rn) * 3 * E + h * D;
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = dq[d] * scale;
  }
  __syncthreads();
  if (i < S) {
    const int j = i;  // this lane now owns key/value row j
    float dk[D], dv[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    for (int r = 0; r < S; ++r) {
      const float ds = dss[r][j], pd = pds[r][j];
#pragma unroll
      for (int d = 0; d < D; ++d) { dk[d] = fmaf(ds, qs[r][d], dk[d]); dv[d] = fmaf(pd, dos[r][d], dv[d]); }
    }
    float* o = dqkv + ((long)j * rs + (long)n * rn) * 3 * E + h * D;
#pragma unroll
    for (int d = 0; d < D; ++d) { o[E + d] = dk[d] * scale; o[2 * E + d] = dv[d]; }
  }
}

// ------------------------------------------------- relu / dropout element-wise
// mode 0: y = relu(x)           mode 1: dx = dy * (y > 0)
// mode 2: y = dropout(x)        mode 3: dx = dropout-mask * dy     (mask regenerated from seed)
template <int MODE>
__global__ __launch_bounds__(256) void tf_ew_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    float* __restrict__ out, long n, float p_drop,
                                                    unsigned long long seed) {
  const long stride = (long)gridDim.x * blockDim.x;
  const float keep = 1.f / (1.f - p_drop);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float v;
    if (MODE == 0) v = fmaxf(a[i], 0.f);
    else if (MODE == 1) v = b[i] > 0.f ? a[i] : 0.f;
    else v = rng01(seed, (unsigned long long)i) >= p_drop ? a[i] * keep : 0.f;
    out[i] = v;
  }
}

inline int ew_grid(long n) {
  long b = (n + 1023) / 1024;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

namespace {
template <int D>
int launch_mha_bwd(const float* qkv, const float* probs, const float* dout, float* dqkv, int S, int N, int H, float p_drop,
                   unsigned long long seed, long rs, long rn, hipStream_t st) {
  const size_t lds = (size_t)(4 * 64 * (D + 1) + 2 * 64 * 65) * sizeof(float);
  (void)hipFuncSetAttribute((const void*)mha_bwd_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((mha_bwd_kernel<D>), dim3(N * H), dim3(64), lds, st, qkv, probs, dout, dqkv, S, N, H,
                     1.0f / sqrtf((float)D), p_drop, seed, rs, rn);
  return check_launch("mha_bwd");
}
}  // namespace

extern "C" {

int wfae_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                       float* mean, float* rstd, int rows, int E, float eps, wfae_stream_t stream) {
  WFAE_REQUIRE(x && gamma && beta && y && mean && rstd, WFAE_ERR_NULL_POINTER, "layernorm_fwd: null pointer");
  WFAE_REQUIRE(rows > 0 && E > 0, WFAE_ERR_BAD_SHAPE, "layernorm_fwd: bad shape");
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, res, gamma, beta,
                     y, mean, rstd, rows, E, eps);
  return check_launch("layernorm_fwd");
}

int wfae_layernorm_bwd(const float* dy, const float* x, const float* res, const float* gamma, const float* mean,
                       const float* rstd, float* dx, float* dgamma, float* dbeta, int rows, int E, int accumulate,
                       void* ws, size_t ws_bytes, wfae_stream_t stream) {
  WFAE_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta, WFAE_ERR_NULL_POINTER,
               "layernorm_bwd: null pointer");
  WFAE_REQUIRE(rows > 0 && E > 0 && E <= 2048, WFAE_ERR_BAD_SHAPE, "layernorm_bwd: bad shape");
  // 8 rows per block (2 per wave): the row loop is a chain of dependent wave reductions, so rows per wave, not bytes,
  // set the time — 64 rows per block left 2048-row problems on 32 blocks (120 us per call in the ViT step)
  int blocks = cdiv(rows, 8);
  if (blocks > 1024) blocks = 1024;
  const int rpb = cdiv(rows, blocks);
  blocks = cdiv(rows, rpb);
  const size_t need = (size_t)blocks * 2 * E * sizeof(float);
  WFAE_REQUIRE(ws && ws_bytes >= need, WFAE_ERR_WORKSPACE, "layernorm_bwd: workspace");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), (size_t)8 * E * sizeof(float), st, dy, x, res,
                     gamma, mean, rstd, dx, (float*)ws, rows, E, rpb);
  int rc = check_launch("layernorm_bwd");
  if (rc) return rc;
  hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(cdiv(E, 64)), dim3(1024), 0, st, (const float*)ws, blocks, E, dgamma,
                     dbeta, accumulate);
  return check_launch("layernorm_bwd_reduce");
}


int wfae_mha_fwd(const float* qkv, float* out, float* probs, int S, int N, int H, int D, int batch_first, float p_drop,
                 uint64_t seed, wfae_stream_t stream) {
  WFAE_REQUIRE(qkv && out && probs, WFAE_ERR_NULL_POINTER, "mha_fwd: null pointer");
  WFAE_REQUIRE(S > 0 && S <= 64 && N > 0 && H > 0, WFAE_ERR_BAD_SHAPE, "mha_fwd: sequence length must be 1..64");
  WFAE_REQUIRE(p_drop >= 0.f && p_drop < 1.f, WFAE_ERR_BAD_SHAPE, "mha_fwd: dropout probability");
  // the kernels read K / V rows as float4: rows are 3*H*D floats long, so the base pointer decides the alignment
  WFAE_REQUIRE((reinterpret_cast<uintptr_t>(qkv) & 15) == 0 && (3 * H * D) % 4 == 0, WFAE_ERR_BAD_SHAPE,
               "mha_fwd: qkv must be 16-byte aligned (got %p; pass a contiguous tensor, not a sliced view)", (const void*)qkv);
  hipStream_t st = (hipStream_t)stream;
  const float scale = 1.0f / sqrtf((float)D);
  const long rs = batch_first ? 1 : N, rn = batch_first ? S : 1;
  const unsigned long long sd = (unsigned long long)seed;
  if (D == 8)
    hipLaunchKernelGGL((mha_fwd_kernel<8>), dim3(N * H), dim3(64), 0, st, qkv, out, probs, S, N, H, scale, p_drop, sd, rs, rn);
  else if (D == 16)
    hipLaunchKernelGGL((mha_fwd_kernel<16>), dim3(N * H), dim3(64), 0, st, qkv, out, probs, S, N, H, scale, p_drop, sd, rs, rn);
  else if (D == 64)
    hipLaunchKernelGGL((mha_fwd_kernel<64>), dim3(N * H), dim3(64), 0, st, qkv, out, probs, S, N, H, scale, p_drop, sd, rs, rn);
  else
    return fail(WFAE_ERR_UNSUPPORTED, "mha_fwd: head dim %d (8, 16 and 64 built)", D);
  return check_launch("mha_fwd");
}

int wfae_mha_bwd(const float* qkv, const float* probs, const float* dout, float* dqkv, int S, int N, int H, int D,
                 int batch_first, float p_drop, uint64_t seed, wfae_stream_t stream) {
  WFAE_REQUIRE(qkv && probs && dout && dqkv, WFAE_ERR_NULL_POINTER, "mha_bwd: null pointer");
  WFAE_REQUIRE(S > 0 && S <= 64 && N > 0 && H > 0, WFAE_ERR_BAD_SHAPE, "mha_bwd: sequence length must be 1..64");
  WFAE_REQUIRE(p_drop >= 0.f && p_drop < 1.f, WFAE_ERR_BAD_SHAPE, "mha_bwd: dropout probability");
  hipStream_t st = (hipStream_t)stream;
  const long rs = batch_first ? 1 : N, rn = batch_first ? S : 1;
  const unsigned long long sd = (unsigned long long)seed;
  if (D == 8) return launch_mha_bwd<8>(qkv, probs, dout, dqkv, S, N, H, p_drop, sd, rs, rn, st);
  if (D == 16) return launch_mha_bwd<16>(qkv, probs, dout, dqkv, S, N, H, p_drop, sd, rs, rn, st);
  if (D == 64) return launch_mha_bwd<64>(qkv, probs, dout, dqkv, S, N, H, p_drop, sd, rs, rn, st);
  return fail(WFAE_ERR_UNSUPPORTED, "mha_bwd: head dim %d (8, 16 and 64 built)", D);
}

int wfae_mha_seqfirst_fwd(const float* qkv, float* out, float* probs, int S, int N, int H, int D, float p_drop,
                          uint64_t seed, wfae_stream_t stream) {
  return wfae_mha_fwd(qkv, out, probs, S, N, H, D, 0, p_drop, seed, stream);
}

int wfae_mha_seqfirst_bwd(const float* qkv, const float* probs, const float* dout, float* dqkv, int S, int N, int H,
                          int D, float p_drop, uint64_t seed, wfae_stream_t stream) {
  return wfae_mha_bwd(qkv, probs, dout, dqkv, S, N, H, D, 0, p_drop, seed, stream);
}

int wfae_relu_fwd(const float* x, float* y, int64_t n, wfae_stream_t stream) {
  WFAE_REQUIRE(x && y, WFAE_ERR_NULL_POINTER, "relu_fwd: null pointer");
  if (n <= 0) return WFAE_OK;
  hipLaunchKernelGGL((tf_ew_kernel<0>), dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr, y,
                     (long)n, 0.f, 0ull);
  return check_launch("relu_fwd");
}

int wfae_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, wfae_stream_t stream) {
  WFAE_REQUIRE(dy && y && dx, WFAE_ERR_NULL_POINTER, "relu_bwd: null pointer");
  if (n <= 0) return WFAE_OK;
  hipLaunchKernelGGL((tf_ew_kernel<1>), dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, (long)n, 0.f, 0ull);
  return check_launch("relu_bwd");
}

int wfae_dropout(const float* x, float* y, int64_t n, float p_drop, uint64_t seed, wfae_stream_t stream) {
  WFAE_REQUIRE(x && y, WFAE_ERR_NULL_POINTER, "dropout: null pointer");
  WFAE_REQUIRE(p_drop >= 0.f && p_drop < 1.f, WFAE_ERR_BAD_SHAPE, "dropout: probability");
  if (n <= 0) return WFAE_OK;
  hipLaunchKernelGGL((tf_ew_kernel<2>), dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr, y,
                     (long)n, p_drop, (unsigned long long)seed);
  return check_launch("dropout");
}

}  // extern "C"
