/*
 * wfae.h — C ABI of libwfae.so: hand-written HIP/gfx950 kernels for the
 * conv-autoencoder training hot path of Autobot37/weatherforecastingtoolkit.
 *
 * The reference has NO native/FFI boundary (it is 100 % Python, every kernel
 * is a stock ATen op reached through torch.nn — SURVEY.md §0.1, §8b), so this
 * ABI is defined by the build.  Each entry point names the reference call
 * site whose ATen op it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - all tensors are dense fp32, NCHW, caller-owned DEVICE pointers
 *     (torch: tensor.data_ptr()); the library never allocates, frees or
 *     retains device memory;
 *   - `ws`/`ws_bytes` is caller-provided scratch (query with
 *     wfae_workspace_bytes); contents are undefined after the call;
 *   - `stream` is a hipStream_t passed as void* (torch:
 *     torch.cuda.current_stream().cuda_stream); every launch goes to it, no
 *     implicit device synchronisation;
 *   - return 0 on success, a negative wfae_status otherwise; never throws or
 *     aborts; wfae_last_error_string() describes the calling thread's last
 *     failure;
 *   - re-entrant and thread-safe: no mutable global state except the
 *     matmul-precision mode below (one atomic int, set once at start-up);
 *   - `accumulate` != 0 means "out += result" (gradient accumulation),
 *     0 means "out = result".
 */
#ifndef WFAE_H
#define WFAE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* wfae_stream_t;

typedef enum {
  WFAE_OK = 0,
  WFAE_ERR_BAD_SHAPE = -1,
  WFAE_ERR_NULL_POINTER = -2,
  WFAE_ERR_WORKSPACE = -3,
  WFAE_ERR_LAUNCH = -4,
  WFAE_ERR_UNSUPPORTED = -5
} wfae_status;

/* ABI history (wfae_version() = 100 * major + minor; a caller built against another minor must re-check the entry
 * points named here):
 *   100  round 1.
 *   101  wfae_conv1x1_fwd_stats / wfae_conv1x1_fwd_bnact / wfae_bn_stats_from_rows: `stat_part` changed from float* to
 *        double* and `stat_capacity` counts DOUBLES (the epilogue sums are reduced in fp64).  A caller built against 100
 *        would hand over a buffer of half the byte size: check wfae_version() >= 101 before using them.
 *   102  round 3: wfae_c1gemm_* (1x1 convolutions on the bf16 matrix pipe with exact split operands, fused BatchNorm
 *        backward epilogues), the bf16 ACTIVATION STORAGE entry points (*_bf16, *_bf16in, *_bf16out, wfae_convert_*),
 *        wfae_c1b_*, wfae_c1w_*, wfae_g3b_* added; nothing removed.
 *   103  round 4: REMOVED wfae_c1gemm_* (six entry points) and wfae_conv1x1_bwd_data_bnred / _bndx (the fused
 *        BatchNorm-backward epilogues of the tile-synchronous GEMMs: parity-green, never faster — the record is
 *        profiles/r03_kbench_c1_fused_bn_backward.txt); wfae_g3b_fwd no longer serves 8 channels per group;
 *        wfae_bn_act_bwd_from_rows now finishes the partial rows of wfae_c1r_bnred.  ADDED
 *        wfae_c1r_* (register-direct 1x1 convolutions: fp32 tensors, C <= 256 stages and the C >= 512 widening products;
 *        wfae_c1r_bnred / wfae_c1r_bndx: the BatchNorm backward of the layer in front in the data gradient's epilogue) and
 *        wfae_c1rb_* (the same on bf16-stored tensors, every stage). */
int wfae_version(void);
const char* wfae_last_error_string(void);
/* upper bound of scratch bytes any single call needs for a problem whose
 * largest weight tensor has `max_weight_elems` elements. */
size_t wfae_workspace_bytes(int64_t max_weight_elems);

/* ---- arithmetic mode of the MFMA GEMM family ------------------------------
 * counterpart of torch.set_float32_matmul_precision(...), which the reference
 * calls once at start-up (experiments/ae_v2/train.py:270, ae_v2_2/train.py:223)
 * and of BASELINE config 5 ("bf16").  WFAE_PRECISION_FP32 (default): operands
 * and accumulation in fp32 (v_mfma_f32_32x32x2_f32).  WFAE_PRECISION_BF16:
 * tensors stay fp32 in HBM; every GEMM-family kernel (1x1 / linear / 4x4 /
 * Winograd / flat-shift convolutions, all three roles) rounds its two operands
 * to bf16 (RNE) on the way from LDS to the matrix core and accumulates in fp32
 * (v_mfma_f32_32x32x16_bf16) — torch's 'medium'.  Process-wide; read at launch. */
enum { WFAE_PRECISION_FP32 = 0, WFAE_PRECISION_BF16 = 1 };
int wfae_set_matmul_precision(int mode);
int wfae_get_matmul_precision(void);

/* fp32 GEMMs on the bf16 matrix pipe ("split" operands, see the Winograd section below): at WFAE_PRECISION_FP32 the
 * MFMA-bound GEMMs carry every fp32 operand exactly as three bf16 values and multiply with six
 * v_mfma_f32_32x32x16_bf16 per fp32 product — fp32 accuracy at up to 16/6 of the fp32 instruction's rate.  On by
 * default (environment WFAE_SPLIT_GEMM=0 turns it off at load); 0 keeps every GEMM on v_mfma_f32_32x32x2_f32.
 * Process-wide; read at launch. */
int wfae_set_split_gemm(int on);
int wfae_get_split_gemm(void);

/* ---- 1x1 convolution as an fp32-MFMA GEMM on NCHW ------------------------
 * replaces nn.Conv2d(C, C/4, 1) / (C/4, C, 1) in Bottleneck
 * (pipeline/models/ae_64x8x8_lin.py:15,19) and the latent projections
 * enc[4] / dec[0] (:69,79).
 * y[n,co,p] = sum_ci w[co,ci] x[n,ci,p] (+ bias[co]) (+ res[n*res_img_stride + co*HW + p])
 * res_img_stride = Cout*HW for the Bottleneck residual (:22), 0 for the
 * broadcast pos_emb add (:91). */
int wfae_conv1x1_fwd(const float* x, const float* w, const float* bias, const float* res,
                     int64_t res_img_stride, float* y, int NB, int Cin, int Cout, int HW,
                     wfae_stream_t stream);
/* Same convolution with the BatchNorm statistics of y (the next layer of Bottleneck is always
 * BatchNorm2d, ae_64x8x8_lin.py:14-19) reduced in the epilogue instead of by a second pass over y:
 * stat_part (capacity stat_capacity DOUBLES) receives *stat_rows partial rows, sum[rows][Cout] followed by
 * sumsq[rows][Cout] (one row per 64-pixel wave tile, reduced in fp64 from fp32 sums of four — the arithmetic of the
 * separate statistics pass, so its results are reproduced to fp64 rounding); wfae_bn_stats_from_rows finishes them.  When the
 * shape is not served by the vector epilogue *stat_rows is 0: y is still complete, run
 * wfae_bn_stats_train on it.  stat_rows is a HOST pointer. */
int wfae_conv1x1_fwd_stats(const float* x, const float* w, const float* bias, const float* res,
                           int64_t res_img_stride, float* y, int NB, int Cin, int Cout, int HW,
                           double* stat_part, int64_t stat_capacity, int* stat_rows, wfae_stream_t stream);
/* BatchNorm-apply + GELU fused into the 1x1 GEMM's operand loader (SURVEY.md 2.2 K3/K7/K8 "fused prologue"):
 * y = conv1x1(gelu(x * bn_scale[c] + bn_shift[c]), w) (+bias)(+res) and, for the backward pass,
 * dw (+)= dy * gelu(x * bn_scale + bn_shift)^T — the activated tensor of the reference's BN -> GELU -> Conv1x1 chain
 * (pipeline/models/ae_64x8x8_lin.py:14-15) is rebuilt between the global load and the LDS store and never exists in
 * HBM.  bn_scale / bn_shift [Cin] are the folded vectors wfae_bn_stats_train / wfae_bn_fold_eval produce; results are
 * bit-identical to wfae_bn_act_fwd followed by wfae_conv1x1_fwd / wfae_conv1x1_bwd_weight (in either matmul precision).
 * HW % 4 == 0 (>= 16 for the weight gradient), channel counts % 4 == 0 and 16-byte aligned tensors only
 * (WFAE_ERR_UNSUPPORTED otherwise: run the two-kernel form). */
int wfae_conv1x1_fwd_bnact(const float* x, const float* bn_scale, const float* bn_shift, const float* w,
                           const float* bias, const float* res, int64_t res_img_stride, float* y, int NB, int Cin,
                           int Cout, int HW, double* stat_part, int64_t stat_capacity, int* stat_rows,
                           wfae_stream_t stream);   /* stat_part / stat_rows: as wfae_conv1x1_fwd_stats, or both null */
int wfae_conv1x1_bwd_weight_bnact(const float* dy, const float* x, const float* bn_scale, const float* bn_shift,
                                  float* dw, int NB, int Cin, int Cout, int HW, int accumulate, void* ws,
                                  size_t ws_bytes, wfae_stream_t stream);
int wfae_conv1x1_bwd_data(const float* dy, const float* w, float* dx, int NB, int Cin, int Cout,
                          int HW, wfae_stream_t stream);
int wfae_conv1x1_bwd_weight(const float* dy, const float* x, float* dw, int NB, int Cin, int Cout,
                            int HW, int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream);

/* ---- The Bottleneck's 1x1 convolutions at the HBM-bound stages, "register-direct" (csrc/c1r.hip, ABI 103).
 * Same products as wfae_conv1x1_fwd[_stats|_bnact] / wfae_conv1x1_bwd_data (pipeline/models/ae_64x8x8_lin.py:15,19):
 *   y[n,m,p] = sum_k A[m,k] f(x[n,k,p]) (+ res[n,m,p]),   A[m,k] = w[m * w_sm + k * w_sk]
 * (forward: M = Cout, K = Cin, w_sm = Cin, w_sk = 1; data gradient: M = Cin, K = Cout, w_sm = 1, w_sk = Cin — the same
 * (Cout, Cin) weight tensor either way).  fp32 tensors; every operand value is carried exactly as three bf16 values (six
 * v_mfma_f32_16x16x32_bf16 products per fp32 product, fp32 accumulation: the accuracy of the fp32 path, needs the split
 * GEMMs on).  The activation goes HBM -> registers -> matrix core with no LDS staging (16-byte loads of four pixels of eight
 * channel rows per lane; the four pixel components feed four MFMAs), results leave as 16-byte stores; the weights are split
 * by the kernel itself into an LDS image.  Served: (M, K) = (32, 128), (64, 256), (128, 32), (256, 64) — the C = 128 and
 * C = 256 stages — with HW % 64 == 0 (wfae_c1r_supported).
 *   pro_scale / pro_shift [K] non-null: f = GELU(x * scale[k] + shift[k]) (wfae_conv1x1_fwd_bnact's prologue), else identity;
 *   res non-null: residual add (M > K only);
 *   stat_part non-null: BatchNorm sums of y, *stat_rows = wfae_c1r_stat_rows(...) partial rows, sum[rows][M] then
 *   sumsq[rows][M] in fp64 (one row per wave of the persistent grid; finish with wfae_bn_stats_from_rows); capacity
 *   2 * rows * M doubles.  stat_rows is a HOST pointer. */
int wfae_c1r_supported(int M, int K, int HW);
int wfae_c1r_stat_rows(int M, int K, int NB, int HW);
int wfae_c1r_fwd(const float* w, int64_t w_sm, int64_t w_sk, const float* x, const float* pro_scale, const float* pro_shift,
                 const float* res, float* y, int NB, int K, int M, int HW, double* stat_part, int64_t stat_capacity,
                 int* stat_rows, wfae_stream_t stream);
/* The widening DATA GRADIENT of the C <= 256 stages with the reductions of the BatchNorm + GELU backward of the layer in front
 * taken in its epilogue (csrc/c1r.hip, ABI 103): da (NB,M,HW) = A dt as wfae_c1r_fwd, and part = sum dU [rows][M] then sum dU xhat
 * [rows][M] in fp64, dU = da * gelu'(x * bn_scale + bn_shift), xhat = (x - save_mean) * save_invstd, x (NB,M,HW) the BatchNorm
 * input — phase 1 of wfae_bn_act_bwd without its pass over (da, x); da may be NULL (the sums alone).  Finish with wfae_bn_act_bwd_from_rows (dgamma, dbeta and the
 * coefficients at the head of ws), then wfae_bn_act_bwd(phases = 2, same ws).  rows <= wfae_c1r_stat_rows(M, K, NB, HW). */
int wfae_c1r_bnred_supported(int M, int K, int HW);
int wfae_c1r_bnred(const float* w, int64_t w_sm, int64_t w_sk, const float* dt, const float* x, const float* bn_scale,
                   const float* bn_shift, const float* save_mean, const float* save_invstd, float* da, int NB, int K, int M, int HW,
                   double* part, int64_t part_capacity, int* part_rows, wfae_stream_t stream);
int wfae_bn_act_bwd_from_rows(const double* part, int rows, int C, float* dgamma, float* dbeta, int accumulate, void* ws,
                              size_t ws_bytes, wfae_stream_t stream);
/* The second pass of that backward in the epilogue of the SAME data gradient, computed again from dt (csrc/c1r.hip, ABI 103):
 *   dx (NB,M,HW) = gamma * save_invstd * (dU - coef[2c] / n - xhat * coef[2c+1] / n) + res,   n = NB * HW,
 * dU, xhat as above with da = A dt rebuilt in registers; coef = the two sums per channel that wfae_bn_act_bwd_from_rows leaves at
 * the head of its ws (training = 0: both taken as zero); res (the gradient arriving over the skip connection) may be null.  With
 * wfae_c1r_bnred(da = NULL) — the sums alone, nothing stored — the layer's backward moves dt twice and x twice, res and dx once,
 * where the da-storing sequence (wfae_c1r_bnred + wfae_bn_act_bwd phases = 2) also writes and re-reads da.  Replaces
 * wfae_bn_act_bwd(phases = 2) for the shapes of wfae_c1r_bnred_supported. */
int wfae_c1r_bndx(const float* w, int64_t w_sm, int64_t w_sk, const float* dt, const float* x, const float* gamma,
                  const float* bn_scale, const float* bn_shift, const float* save_mean, const float* save_invstd, const float* coef,
                  const float* res, float* dx, int NB, int K, int M, int HW, int training, wfae_stream_t stream);
/* The same register-direct product on bf16-STORED tensors (csrc/c1rb.hip, ABI 103; needs WFAE_PRECISION_BF16): bf16 pieces of
 * eight pixels per lane go HBM -> registers -> v_mfma_f32_16x16x32_bf16 with four byte-permutes per fragment and no other
 * arithmetic, the weight as one bf16 plane rounded by the kernel itself, fp32 accumulation, one rounding of the result.  Serves
 * every Bottleneck product (M, K) = (C/4, C) and (C, C/4), C = 128 .. 1024, with HW % 128 == 0 (wfae_c1rb_supported); other
 * shapes: wfae_c1b_fwd.  Arguments as wfae_c1r_fwd with uint16_t tensors; the BatchNorm sums are those of the ROUNDED result. */
int wfae_c1rb_supported(int M, int K, int HW);
int wfae_c1rb_stat_rows(int M, int K, int NB, int HW);
int wfae_c1rb_fwd(const float* w, int64_t w_sm, int64_t w_sk, const uint16_t* x, const float* pro_scale, const float* pro_shift,
                  const uint16_t* res, uint16_t* y, int NB, int K, int M, int HW, double* stat_part, int64_t stat_capacity,
                  int* stat_rows, wfae_stream_t stream);
/* Weight gradients of the Bottleneck's 1x1 convolutions with both operands read as K-contiguous rows (csrc/c1w.hip, ABI
 * 102): dw (Cout,Cin) (+)= dy (NB,Cout,HW) . x'^T, x' = x or (bn_scale / bn_shift non-null) gelu(x * bn_scale[c] +
 * bn_shift[c]) rebuilt in the loader (wfae_conv1x1_bwd_weight_bnact's prologue).  fp32 tensors: exact three-plane bf16
 * split at the LDS store, six bf16 MFMA products per fp32 product (needs the split GEMMs on); _bf16: bf16-stored tensors,
 * one product (needs WFAE_PRECISION_BF16).  HW % 32 == 0, channel counts % 32 == 0 (wfae_c1w_supported); ws holds the
 * split-K slabs: up to NB * ceil(HW / 256) slabs of Cout * Cin floats (fewer when they would not fit). */
int wfae_c1w_supported(int Cin, int Cout, int HW);
int wfae_c1w_bwd_weight(const float* dy, const float* x, const float* bn_scale, const float* bn_shift, float* dw, int NB, int Cin,
                        int Cout, int HW, int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream);
int wfae_c1w_bwd_weight_bf16(const uint16_t* dy, const uint16_t* x, const float* bn_scale, const float* bn_shift, float* dw, int NB,
                             int Cin, int Cout, int HW, int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream);
/* ---- nn.Linear (to_latent / from_latent, ae_64x8x8_lin.py:74-75,92,98) ---
 * y[b,o] = sum_i x[b,i] w[o,i] + bias[o] */
int wfae_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int In,
                    int Out, void* ws, size_t ws_bytes, wfae_stream_t stream);
int wfae_linear_bwd_data(const float* dy, const float* w, float* dx, int B, int In, int Out,
                         wfae_stream_t stream);
int wfae_linear_bwd_weight(const float* dy, const float* x, float* dw, int B, int In, int Out,
                           int accumulate, wfae_stream_t stream);
/* the same with the batch dimension split across workgroups (deterministic slab reduce): for layers whose
 * Out x In tile grid is much smaller than the chip (transformer blocks) */
int wfae_linear_bwd_data_splitk(const float* dy, const float* w, float* dx, int B, int In, int Out, void* ws,
                                size_t ws_bytes, wfae_stream_t stream);
int wfae_linear_bwd_weight_splitk(const float* dy, const float* x, float* dw, int B, int In, int Out, int accumulate,
                                  void* ws, size_t ws_bytes, wfae_stream_t stream);

/* ---- 4x4 stride-2 pad-1 convolution pair ---------------------------------
 * "hi" = the 2H x 2W side with Chi channels, "lo" = the H x W side with Clo
 * channels.  Both nn.Conv2d(Chi, Clo, 4, 2, 1) weights (Clo,Chi,4,4)
 * (EncBlock.down[0], ae_64x8x8_lin.py:31) and nn.ConvTranspose2d(Clo, Chi, 4,
 * 2, 1) weights (Clo,Chi,4,4) (DecBlock.up[0], :42) are laid out
 * [lo][hi][ky][kx], so three kernels serve six operations:
 *   down : lo[n,l,oy,ox] = sum_{h,ky,kx} w[l,h,ky,kx] hi[n,h,2oy-1+ky,2ox-1+kx]
 *          = Conv2d forward            = ConvTranspose2d backward-data
 *   up   : hi[n,h,iy,ix] = sum_{l,ky,kx: iy=2oy-1+ky} w[l,h,ky,kx] lo[n,l,oy,ox]
 *          = ConvTranspose2d forward   = Conv2d backward-data
 *   wgrad: dw[l,h,ky,kx] = sum_{n,oy,ox} lo[n,l,oy,ox] hi[n,h,2oy-1+ky,2ox-1+kx]
 *          = Conv2d backward-weight (lo=dy, hi=x) = ConvTranspose2d
 *            backward-weight (lo=x, hi=dy)
 * Hlo,Wlo are the lo-side spatial dims. */
int wfae_conv4x4s2_down(const float* hi, const float* w, float* lo, int NB, int Chi, int Clo,
                        int Hlo, int Wlo, wfae_stream_t stream);
int wfae_conv4x4s2_up(const float* lo, const float* w, float* hi, int NB, int Chi, int Clo, int Hlo,
                      int Wlo, void* ws, size_t ws_bytes, wfae_stream_t stream);
int wfae_conv4x4s2_wgrad(const float* lo, const float* hi, float* dw, int NB, int Chi, int Clo,
                         int Hlo, int Wlo, int accumulate, void* ws, size_t ws_bytes,
                         wfae_stream_t stream);

/* ---- direct (im2col-free, LDS-tiled VALU) convolution ---------------------
 * grouped 3x3 pad 1 of Bottleneck (ae_64x8x8_lin.py:17), the 128->1 3x3 output
 * conv (:84) and the 1->256 4x4 s2 input conv (:31 with in_ch=1).
 * w is (Cout, Cin/groups, KS, KS).  (KS, stride) in {(3,1), (4,2), (4,1)}; (4,1) is the fourth conv of the
 * PatchGAN discriminator (losses/model.py:137).  bwd_data supports stride 1 only (H, W = input dims;
 * stride-2 data gradients go through wfae_conv4x4s2_up). */
int wfae_dconv_fwd(const float* x, const float* w, const float* bias, float* y, int NB, int Cin,
                   int Cout, int H, int W, int KS, int stride, int pad, int groups,
                   wfae_stream_t stream);
int wfae_dconv_bwd_data(const float* dy, const float* w, float* dx, int NB, int Cin, int Cout,
                        int H, int W, int KS, int pad, int groups, wfae_stream_t stream);
int wfae_dconv_bwd_weight(const float* dy, const float* x, float* dw, int NB, int Cin, int Cout,
                          int H, int W, int KS, int stride, int pad, int groups, int accumulate,
                          void* ws, size_t ws_bytes, wfae_stream_t stream);

/* grouped 3x3 pad-1 stride-1 convolution with Cin == Cout == C and C/groups in {4,8,16,32}
 * (the Bottleneck middle conv, ae_64x8x8_lin.py:17), register-blocked kernels.
 * fwd: transposed=0 -> y = conv(x, w); transposed=1 -> data gradient dx = conv^T(dy, w) (pass dy as x).
 * bwd_weight: dw[(g,oc),ci,ky,kx] as a per-group fp32-MFMA implicit GEMM over the pixels. */
int wfae_gconv3x3_fwd(const float* x, const float* w, float* y, int NB, int C, int H, int W, int groups,
                      int transposed, void* ws, size_t ws_bytes, wfae_stream_t stream);
int wfae_gconv3x3_bwd_weight(const float* dy, const float* x, float* dw, int NB, int C, int H, int W,
                             int groups, int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream);

/* ---- BatchNorm2d (+GELU) --------------------------------------------------
 * nn.BatchNorm2d(eps=1e-5, momentum=0.1) followed by nn.GELU()
 * (ae_64x8x8_lin.py:14,16,18,32,43).
 * bn_stats_train: per-channel batch mean / biased variance over (N,H,W)
 *   (fp64 accumulation), writes save_mean, save_invstd, the folded
 *   scale = gamma*invstd and shift = beta - mean*scale, and updates
 *   running_mean/var (unbiased var, momentum) in place.
 * bn_fold_eval: scale/shift from running statistics (module.eval()).
 * bn_act_fwd: y = act(x*scale[c] + shift[c]); act 0 = identity, 1 = exact GELU, 2 = LeakyReLU(0.2).
 * bn_act_bwd phases: 1 = per-channel reductions (+ dgamma/dbeta and the coefficients kept at the head of ws),
 *   2 = the dx kernel (reads those coefficients: same ws, no other call in between), 3 = both.
 * bn_act_bwd: given dy = dL/dy, x and the saved statistics, computes
 *   dgamma, dbeta and dx (+ res, the residual-branch gradient of
 *   Bottleneck, :22).  training=0 uses the eval-mode formula. */
int wfae_bn_stats_train(const float* x, int NB, int C, int HW, const float* gamma,
                        const float* beta, float eps, float momentum, float* running_mean,
                        float* running_var, float* save_mean, float* save_invstd, float* scale,
                        float* shift, void* ws, size_t ws_bytes, wfae_stream_t stream);
/* wfae_bn_stats_train's second half on the partial rows of wfae_conv1x1_fwd_stats (same outputs) */
int wfae_bn_stats_from_rows(const double* stat_part, int rows, int NB, int C, int HW, const float* gamma,
                            const float* beta, float eps, float momentum, float* running_mean,
                            float* running_var, float* save_mean, float* save_invstd, float* scale,
                            float* shift, void* ws, size_t ws_bytes, wfae_stream_t stream);
int wfae_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, float* save_mean, float* save_invstd,
                      float* scale, float* shift, int C, wfae_stream_t stream);
int wfae_bn_act_fwd(const float* x, const float* scale, const float* shift, float* y, int NB, int C,
                    int HW, int act, wfae_stream_t stream);
/* wfae_bn_act_fwd that also reduces the BatchNorm sums of ITS OUTPUT y (the first Bottleneck after a Down/Up unit
 * normalises y again): same fp64 partial layout as wfae_wino_out_stats, needs part_capacity >= 2 * C * NB * ceil(HW / 4096)
 * doubles.  wfae_bn_stats_from_parts = the second half of wfae_bn_stats_train (mean / invstd / folded scale, shift,
 * running-statistics update) on such partial sums; the sums are accumulated in fp64 like the separate pass, so the
 * statistics agree with it to fp64 rounding. */
int wfae_bn_act_fwd_stats(const float* x, const float* scale, const float* shift, float* y, int NB, int C, int HW, int act,
                          double* part, int64_t part_capacity, int* splits_out, wfae_stream_t stream);
int wfae_bn_stats_from_parts(const double* part, int splits, int NB, int C, int HW, const float* gamma, const float* beta,
                             float eps, float momentum, float* running_mean, float* running_var, float* save_mean,
                             float* save_invstd, float* scale, float* shift, wfae_stream_t stream);
int wfae_bn_act_bwd(const float* dy, const float* x, const float* gamma, const float* scale,
                    const float* shift, const float* save_mean, const float* save_invstd,
                    const float* res, float* dx, float* dgamma, float* dbeta, int NB, int C, int HW,
                    int act, int training, int accumulate, int phases, void* ws, size_t ws_bytes,
                    wfae_stream_t stream);

/* ---- bf16 ACTIVATION STORAGE (ABI 102; BASELINE config 5: the reference's bf16 regime, experiments/ae_v2_2/train.py:223
 * + Lightning precision).  In this mode the activation tensors of the convolution stacks — and their gradients — live in
 * HBM as bf16 bit patterns (torch.bfloat16, passed as uint16_t*), halving the traffic of the HBM-bound kernels; parameters,
 * parameter gradients, BatchNorm statistics, the latent vectors, the loss and every accumulation stay fp32, and each kernel
 * widens its inputs to fp32 registers and rounds results to nearest even on the way out.  A `_bf16` entry point is its
 * fp32 namesake with the activation pointers retyped; every lane still moves 16 bytes per access (8 elements instead
 * of 4), so HW % 8 == 0 takes the vector path (other sizes: the scalar path, as HW % 4 != 0 does for fp32). */
int wfae_convert_f32_to_bf16(const float* src, uint16_t* dst, int64_t n, wfae_stream_t stream);
int wfae_convert_bf16_to_f32(const uint16_t* src, float* dst, int64_t n, wfae_stream_t stream);
int wfae_bn_stats_train_bf16(const uint16_t* x, int NB, int C, int HW, const float* gamma, const float* beta, float eps,
                             float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                             float* scale, float* shift, void* ws, size_t ws_bytes, wfae_stream_t stream);
int wfae_bn_act_fwd_bf16(const uint16_t* x, const float* scale, const float* shift, uint16_t* y, int NB, int C, int HW, int act,
                         wfae_stream_t stream);
/* the sums are those of the ROUNDED values written to y (what the next BatchNorm reads) */
int wfae_bn_act_fwd_stats_bf16(const uint16_t* x, const float* scale, const float* shift, uint16_t* y, int NB, int C, int HW,
                               int act, double* part, int64_t part_capacity, int* splits_out, wfae_stream_t stream);
/* 1x1 convolutions on bf16-stored activations (x, res, y / dy, dx bf16; weights, bias, dw, BatchNorm vectors fp32; needs
 * WFAE_PRECISION_BF16; Cin % 4 == 0, HW % 4 == 0).  One entry point per role covers the plain and the fused forms:
 * bn_scale / bn_shift non-null = the BatchNorm + GELU prologue of wfae_conv1x1_fwd_bnact / wfae_conv1x1_bwd_weight_bnact,
 * stat_part / stat_rows non-null = the BatchNorm sums of wfae_conv1x1_fwd_stats (sums of the ROUNDED results). */
int wfae_conv1x1_fwd_bf16(const uint16_t* x, const float* bn_scale, const float* bn_shift, const float* w, const float* bias,
                          const uint16_t* res, int64_t res_img_stride, uint16_t* y, int NB, int Cin, int Cout, int HW,
                          double* stat_part, int64_t stat_capacity, int* stat_rows, wfae_stream_t stream);
int wfae_conv1x1_bwd_data_bf16(const uint16_t* dy, const float* w, uint16_t* dx, int NB, int Cin, int Cout, int HW,
                               wfae_stream_t stream);
int wfae_conv1x1_bwd_weight_bf16(const uint16_t* dy, const uint16_t* x, const float* bn_scale, const float* bn_shift, float* dw,
                                 int NB, int Cin, int Cout, int HW, int accumulate, void* ws, size_t ws_bytes,
                                 wfae_stream_t stream);
/* grouped 3x3 convolution of the Bottleneck on bf16-stored activations (weights / dw fp32) */
int wfae_gconv3x3_fwd_bf16(const uint16_t* x, const float* w, uint16_t* y, int NB, int C, int H, int W, int groups,
                           int transposed, void* ws, size_t ws_bytes, wfae_stream_t stream);
int wfae_gconv3x3_bwd_weight_bf16(const uint16_t* dy, const uint16_t* x, float* dw, int NB, int C, int H, int W, int groups,
                                  int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream);
/* the two full-resolution convolutions with ONE channel on one side keep that side fp32:
 *   dconv_fwd_bf16out:      Conv2d(1, C, 4, 2, 1) forward (encoder first layer): x fp32 (NB,1,H,W) -> y bf16 (NB,C,H/2,W/2)
 *   dconv_fwd_bf16in:       Conv2d(Cin, Cout, 3, 1, 1) forward (the output convolution): x bf16 -> y fp32
 *   dconv_bwd_data_bf16out: data gradient of Conv2d(Cin, 1, 3, 1, 1): dy fp32 (NB,1,H,W) -> dx bf16 (NB,Cin,H,W)
 *   c1_wgrad_bf16:          their weight gradients; flip 0: big = dy bf16 (NB,C,H/2,W/2), small = x fp32 (NB,1,H,W) -> dw
 *                           (C,1,4,4); flip 1: big = x bf16 (NB,C,H,W), small = dy fp32 (NB,1,H,W) -> dw (1,C,3,3) */
int wfae_dconv_fwd_bf16out(const float* x, const float* w, const float* bias, uint16_t* y, int NB, int Cout, int H, int W,
                           wfae_stream_t stream);
int wfae_dconv_fwd_bf16in(const uint16_t* x, const float* w, const float* bias, float* y, int NB, int Cin, int Cout, int H,
                          int W, wfae_stream_t stream);
int wfae_dconv_bwd_data_bf16out(const float* dy, const float* w, uint16_t* dx, int NB, int Cin, int H, int W,
                                wfae_stream_t stream);
int wfae_c1_wgrad_bf16(int flip, const uint16_t* big, const float* small, float* dw, int NB, int C, int H, int W, int accumulate,
                       void* ws, size_t ws_bytes, wfae_stream_t stream);
/* Winograd transforms with the TENSOR side stored as bf16 (operands / products of the transform domain unchanged);
 * wino_out_bf16 / wino_in_t_bf16: part null = plain transform, else also the BatchNorm sums of the rounded result */
int wfae_wino_in_split_bf16(int variant, const uint16_t* hi, uint16_t* V3, int planes, int NB, int Chi, int Hlo, int Wlo,
                            wfae_stream_t stream);
int wfae_wino_out_t_split_bf16(int variant, const uint16_t* lo, uint16_t* Mt3, int planes, int NB, int Clo, int Hlo, int Wlo,
                               wfae_stream_t stream);
int wfae_wino_out_bf16(int variant, const float* M, uint16_t* lo, int NB, int Clo, int Hlo, int Wlo, double* part,
                       int64_t part_capacity, int* splits_out, wfae_stream_t stream);
int wfae_wino_in_t_bf16(int variant, const float* dV, uint16_t* hi, int NB, int Chi, int Hlo, int Wlo, double* part,
                        int64_t part_capacity, int* splits_out, wfae_stream_t stream);
/* The Bottleneck's 1x1 convolutions on bf16-stored activations without a format change between HBM and the matrix core
 * (csrc/c1b.hip): y (bf16) = W f(x) (+ res), W one bf16 plane.  c1b_weights: w (Cout,Cin) fp32 -> Wb [Cout][Cin] and Wtb
 * [Cin][Cout] bf16 (the forward passes Wb with (M, K) = (Cout, Cin), the data gradient Wtb with (M, K) = (Cin, Cout)).
 * c1b_fwd: pro_scale / pro_shift [K] non-null = the BatchNorm + GELU prologue; stat_part non-null = BatchNorm sums of the
 * rounded y as *stat_rows rows, sum[rows][M] then sumsq[rows][M], rows = wfae_c1b_stat_rows(M, K, NB, HW), finish with
 * wfae_bn_stats_from_rows.  Served: M % 32 == 0, K % 32 == 0, HW % 8 == 0 (wfae_c1b_supported), WFAE_PRECISION_BF16 arithmetic. */
int wfae_c1b_supported(int M, int K, int HW);
int wfae_c1b_stat_rows(int M, int K, int NB, int HW);
int wfae_c1b_weights(const float* w, uint16_t* Wb, uint16_t* Wtb, int Cout, int Cin, wfae_stream_t stream);
int wfae_c1b_fwd(const uint16_t* Wb, const uint16_t* x, const float* pro_scale, const float* pro_shift, const uint16_t* res,
                 uint16_t* y, int NB, int K, int M, int HW, double* stat_part, int64_t stat_capacity, int* stat_rows,
                 wfae_stream_t stream);
/* The Bottleneck's grouped 3x3 convolution on bf16-stored activations as an implicit GEMM on the bf16 matrix pipe
 * (csrc/g3b.hip; reference pipeline/models/ae_64x8x8_lin.py:17): same arguments and result as wfae_gconv3x3_fwd_bf16 (which
 * routes here when wfae_g3b_supported says 1), transposed = 1 gives the data gradient.  Served: (channels per group, W) =
 * (4,384), (8,192), (16,96), (32,48), (32,24) with H a multiple of the strip height; workspace >= C/16 * 3 * 2 * 1024 bytes
 * (C/32 * 3 * 3 * 2 * 1024 at 32 channels per group) for the weight fragments.  WFAE_PRECISION_BF16 arithmetic. */
int wfae_g3b_supported(int C, int H, int W, int groups);
int wfae_g3b_fwd_bf16(const uint16_t* x, const float* w, uint16_t* y, int NB, int C, int H, int W, int groups, int transposed,
                      void* ws, size_t ws_bytes, wfae_stream_t stream);
/* weight gradient of the same convolution on the same kernel family (both tensors staged as they lie in memory, the kx shift
 * of a tap taken on the dy fragment in registers): dw (C, C/groups, 3, 3) fp32, += when accumulate; workspace >=
 * NB * (H / strip height) * C * (C/groups) * 9 * 4 bytes (one partial per block, added in fixed order) */
int wfae_g3b_bwd_weight_bf16(const uint16_t* dy, const uint16_t* x, float* dw, int NB, int C, int H, int W, int groups,
                             int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream);
/* the same kernels on fp32 tensors: every value split exactly into three bf16 planes at the LDS store, six products per fp32
 * product (the arithmetic of the split GEMMs: fp32 accuracy).  wfae_g3b_f32_supported(…, wgrad): 1 when the shape is served at
 * the current mode (fp32 precision with the split switch on; weight gradient at W <= 96 only: three planes of full-width rows exceed
 * the LDS at W = 384 and allow one block per CU at W = 192, which measured slower than wfae_gconv3x3_bwd_weight's kernel). */
int wfae_g3b_f32_supported(int C, int H, int W, int groups, int wgrad);
/* forward (transposed = 0) / data gradient (1) on fp32 tensors: (channels per group, W) = (16, 96) is what
 * wfae_g3b_f32_supported(…, 0) reports (faster than wfae_gconv3x3_fwd's kernel there); (8, 192) is accepted as well but measured
 * slower.  Same arguments as wfae_gconv3x3_fwd; workspace >= 3 x the bf16 form's */
int wfae_g3b_fwd(const float* x, const float* w, float* y, int NB, int C, int H, int W, int groups, int transposed, void* ws,
                 size_t ws_bytes, wfae_stream_t stream);
int wfae_g3b_bwd_weight(const float* dy, const float* x, float* dw, int NB, int C, int H, int W, int groups, int accumulate,
                        void* ws, size_t ws_bytes, wfae_stream_t stream);
int wfae_bn_act_bwd_bf16(const uint16_t* dy, const uint16_t* x, const float* gamma, const float* scale, const float* shift,
                         const float* save_mean, const float* save_invstd, const uint16_t* res, uint16_t* dx, float* dgamma,
                         float* dbeta, int NB, int C, int HW, int act, int training, int accumulate, int phases, void* ws,
                         size_t ws_bytes, wfae_stream_t stream);

/* ---- element-wise / reductions -------------------------------------------- */
int wfae_gelu_fwd(const float* x, float* y, int64_t n, wfae_stream_t stream);   /* nn.GELU */
int wfae_gelu_bwd(const float* dy, const float* x, float* dx, int64_t n, wfae_stream_t stream);
int wfae_sigmoid_fwd(const float* x, float* y, int64_t n, wfae_stream_t stream); /* :86,102 */
int wfae_sigmoid_bwd(const float* dy, const float* y, float* dx, int64_t n, wfae_stream_t stream);
int wfae_add(const float* a, const float* b, float* out, int64_t n, wfae_stream_t stream);
/* out[c] (+)= sum_{o,i} x[o,c,i]: conv-bias / pos_emb / linear-bias gradients */
int wfae_reduce_sum(const float* x, int outer, int C, int inner, float* out, int accumulate,
                    void* ws, size_t ws_bytes, wfae_stream_t stream);

/* ---- Winograd forms of the three 4x4 stride-2 operations above (same tensors, same results up to fp32
 * rounding).  The convolution is split into its four input-parity phases (each a 2x2 stride-1 convolution) and
 * every phase goes through F(MxM, 2x2) on N x N input tiles, N = M + 1:
 *   variant 0: F(2x2,2x2), 9 GEMMs,  18   FLOP per (lo channel, hi channel, lo pixel) instead of 32;
 *   variant 1: F(4x4,2x2), 25 GEMMs, 12.5 FLOP; interpolation points {0, +-1, +-2}.
 * xi = N*N transform positions, T = NB*Hlo*Wlo/M^2 tiles.  The pieces are separate entry points so that a
 * training step transforms every tensor once and reuses it (the weight gradient shares both transformed
 * operands with the forward / data gradient of the same layer):
 *   wino_weights:    w (Clo,Chi,4,4)       -> U  [xi][Clo][4Chi]   (G g G^T per phase)
 *   wino_in:         hi (NB,Chi,2Hlo,2Wlo) -> V  [xi][4Chi][T]     (B^T d B of the 2N x 2N patches at stride 2M)
 *   wino_out_t:      lo (NB,Clo,Hlo,Wlo)   -> Mt [xi][Clo][T]      (adjoint of the output transform)
 *   wino_gemm_down:  M  = U * V    [xi][Clo][T]   then  wino_out:  lo = A^T M A           (= wfae_conv4x4s2_down)
 *   wino_gemm_up:    dV = U^T * Mt [xi][4Chi][T]  then  wino_in_t: hi = overlap-add B dV B^T (= wfae_conv4x4s2_up)
 *   wino_gemm_wgrad: dw = G^T (Mt * V^T) G,  ws >= 2 * 4*|U| bytes                        (= wfae_conv4x4s2_wgrad)
 * with Mt = wino_out_t(lo-side tensor), V = wino_in(hi-side tensor).  Supported when Hlo, Wlo are multiples of
 * M, Chi % 4 == 0, Clo % 4 == 0 and T % 4 == 0; wfae_wino_sizes returns WFAE_ERR_UNSUPPORTED otherwise and
 * {T, |U|, |V|, |M|} (element counts) in out4 when it is. */
int wfae_wino_sizes(int variant, int NB, int Chi, int Clo, int Hlo, int Wlo, int64_t* out4);
int wfae_wino_weights(int variant, const float* w, float* U, int Chi, int Clo, wfae_stream_t stream);
int wfae_wino_in(int variant, const float* hi, float* V, int NB, int Chi, int Hlo, int Wlo, wfae_stream_t stream);
int wfae_wino_in_t(int variant, const float* dV, float* hi, int NB, int Chi, int Hlo, int Wlo, wfae_stream_t stream);
int wfae_wino_out(int variant, const float* M, float* lo, int NB, int Clo, int Hlo, int Wlo, wfae_stream_t stream);
/* wfae_wino_out that also reduces the BatchNorm sums of its result (the BatchNorm of EncBlock.down / DecBlock.up follows
 * the convolution directly, reference ae_64x8x8_lin.py:31-32,42-43): fp64 partial sums part[(split * Clo + c) * 2 +
 * {0 = sum, 1 = sum of squares}], *splits_out rows; finish them with wfae_bn_stats_from_parts. */
int wfae_wino_out_stats(int variant, const float* M, float* lo, int NB, int Clo, int Hlo, int Wlo, double* part,
                        int64_t part_capacity, int* splits_out, wfae_stream_t stream);
/* the same for the adjoint input transform, whose result is the output of a ConvTranspose2d (DecBlock.up, :42-43) */
int wfae_wino_in_t_stats(int variant, const float* dV, float* hi, int NB, int Chi, int Hlo, int Wlo, double* part,
                         int64_t part_capacity, int* splits_out, wfae_stream_t stream);
int wfae_wino_out_t(int variant, const float* lo, float* Mt, int NB, int Clo, int Hlo, int Wlo, wfae_stream_t stream);
int wfae_wino_gemm_down(int variant, const float* U, const float* V, float* M, int NB, int Chi, int Clo, int Hlo, int Wlo,
                        wfae_stream_t stream);
int wfae_wino_gemm_up(int variant, const float* U, const float* Mt, float* dV, int NB, int Chi, int Clo, int Hlo, int Wlo,
                      wfae_stream_t stream);
int wfae_wino_gemm_wgrad(int variant, const float* Mt, const float* V, float* dw, int NB, int Chi, int Clo, int Hlo,
                         int Wlo, int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream);

/* ---- The three Winograd-domain products on the bf16 matrix pipe at fp32 accuracy ("split" operands).
 * v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate; v_mfma_f32_32x32x16_bf16 at 16x that.  An fp32 value is
 * exactly h + m + l with three bf16 values (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)) and bf16 x bf16 is exact
 * in fp32, so the six products ah bh, ah bm, am bh, ah bl, al bh, am bm accumulated in fp32 carry the fp32 product to
 * 2^-23 relative — the error of the fp32 MFMA path (tests/test_kernels_gpu.py::test_split_gemm_matches_fp32_accuracy) at
 * 16/6 of its rate.  A split operand is three planes of bf16 bit patterns (uint16_t), the h plane first, each plane
 * the fp32 layout's element count long; the *_split transforms write them directly (6 bytes per element instead of 4),
 * the GEMM outputs stay fp32 and go through wino_out / wino_in_t / the G^T dU G reduction unchanged.
 *   wino_weights_split: U3 [3][xi][Clo][4Chi] and Ut3 [3][xi][4Chi][Clo] (the up product contracts over Clo)
 *   wino_in_split / wino_out_t_split: V3 [3][xi][4Chi][T], Mt3 [3][xi][Clo][T]
 * Range: the split is exact for |x| in [2^-110, 3.3e38) (below, the l plane loses bits to bf16's underflow — the absolute
 * error stays under 2^-133; above, bf16(x) rounds to infinity); NaN and infinities propagate as NaN.
 * `planes` = 3: the exact split above.  `planes` = 1: only the h plane is written / read — bf16-rounded operands with
 * fp32 accumulation, i.e. WFAE_PRECISION_BF16's arithmetic with 2-byte operand storage for the matrix work (one MFMA
 * product per tile; the GEMMs turn HBM-bound).  Buffers hold `planes` planes.
 * wfae_wino_split_supported: 1 when wfae_wino_sizes accepts the geometry and 4 Chi, Clo and T are multiples of 32.
 * wfae_split_bf16x3 / wfae_split_gemm: the conversion and the batched product on their own (C[y] = A[y] B[y], A [M][K];
 * b_kind 0: B [K][N], b_kind 1: B [N][K]; K % 32 == 0, N % 8 == 0 for b_kind 0; planes batches*M*K resp. batches*K*N
 * elements apart). */
int wfae_wino_split_supported(int variant, int NB, int Chi, int Clo, int Hlo, int Wlo);
int wfae_wino_weights_split(int variant, const float* w, uint16_t* U3, uint16_t* Ut3, int planes, int Chi, int Clo,
                            wfae_stream_t stream);
int wfae_wino_in_split(int variant, const float* hi, uint16_t* V3, int planes, int NB, int Chi, int Hlo, int Wlo,
                       wfae_stream_t stream);
int wfae_wino_out_t_split(int variant, const float* lo, uint16_t* Mt3, int planes, int NB, int Clo, int Hlo, int Wlo,
                          wfae_stream_t stream);
int wfae_wino_gemm_down_split(int variant, const uint16_t* U3, const uint16_t* V3, float* M, int planes, int NB, int Chi, int Clo,
                              int Hlo, int Wlo, wfae_stream_t stream);
int wfae_wino_gemm_up_split(int variant, const uint16_t* Ut3, const uint16_t* Mt3, float* dV, int planes, int NB, int Chi, int Clo,
                            int Hlo, int Wlo, wfae_stream_t stream);
int wfae_wino_gemm_wgrad_split(int variant, const uint16_t* Mt3, const uint16_t* V3, float* dw, int planes, int NB, int Chi,
                               int Clo, int Hlo, int Wlo, int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream);
int wfae_split_bf16x3(const float* x, uint16_t* out, int64_t n, int planes, wfae_stream_t stream);
int wfae_split_gemm(int b_kind, int planes, const uint16_t* A3, const uint16_t* B3, float* C, int M, int N, int K, int batches,
                    wfae_stream_t stream);

/* ---- 4x4 stride-1 convolution on the MFMA GEMM (PatchGAN layer 4: Conv2d(256, 512, 4, stride=1, padding=1,
 * bias=False), pipeline/models/autoencoderkl/losses/model.py:137).
 * fwd, transposed = 0: x (NB,Cin,H,W) -> y (NB,Cout,H+2pad-3,W+2pad-3);
 * fwd, transposed = 1: the data gradient — x is dy (NB,Cout,H,W) -> y = dx (NB,Cin,H+3-2pad,W+3-2pad).
 * bwd_weight: dy (NB,Cout,H+2pad-3,W+2pad-3), x (NB,Cin,H,W) -> dw (Cout,Cin,4,4).
 * The channel count of the operand that is read must be a multiple of 16 (else WFAE_ERR_UNSUPPORTED; use
 * wfae_dconv_*).  Workspace: padded copy of the input + padded-width output + packed weights
 * (+ split-K slabs for bwd_weight). */
int wfae_conv4x4s1_fwd(const float* x, const float* w, float* y, int NB, int Cin, int Cout, int H, int W, int pad,
                       int transposed, void* ws, size_t ws_bytes, wfae_stream_t stream);
int wfae_conv4x4s1_bwd_weight(const float* dy, const float* x, float* dw, int NB, int Cin, int Cout, int H, int W,
                              int pad, int accumulate, void* ws, size_t ws_bytes, wfae_stream_t stream);

/* ---- PatchGAN discriminator pieces (pipeline/models/autoencoderkl/losses/model.py:100-150;
 * hinge loss contperceptual.py:19-23; generator term -mean(D(fake)) experiments/ae_v2/train.py:78).
 * leaky_relu: nn.LeakyReLU(0.2).  BatchNorm + LeakyReLU is wfae_bn_act_* with act = 2.
 * pad2d: zero-pad every plane by `pad` pixels (crop = 1: the inverse crop, its gradient) — the final
 *   Conv2d(512, 1, kernel_size=1, padding=1) of the reference is pad + 1x1 conv.
 * mean: out[0] = weight * mean(x) (hinge = 0) or weight * mean(relu(1 + sign*x)) (hinge = 1).
 * scale: y = x * scale * (scale_dev ? scale_dev[0] : 1)  (adaptive weight, gradient clipping). */
int wfae_leaky_relu_fwd(const float* x, float* y, int64_t n, wfae_stream_t stream);
int wfae_leaky_relu_bwd(const float* dy, const float* x, float* dx, int64_t n, wfae_stream_t stream);
int wfae_pad2d(const float* x, float* y, int64_t planes, int H, int W, int pad, int crop, wfae_stream_t stream);
int wfae_mean_fwd(const float* x, float* out, int64_t n, int hinge, float sign, float weight, void* ws,
                  size_t ws_bytes, wfae_stream_t stream);
int wfae_mean_bwd(const float* x, const float* gout, float* dx, int64_t n, int hinge, float sign, float weight,
                  wfae_stream_t stream);
int wfae_scale(const float* x, const float* scale_dev, float scale, float* y, int64_t n, wfae_stream_t stream);
/* gradient clipping by global norm (Lightning clip_gradients -> torch.nn.utils.clip_grad_norm_,
 * experiments/ae_v2_2/train.py:140,155): sumsq_parts are wfae_sumsq results of the gradient buffers;
 * total = pre_scale * sqrt(sum) (pre_scale = 1/world when the buffers hold rank-summed gradients);
 * out2[0] = min(1, max_norm / (total + 1e-6)), out2[1] = total; apply with wfae_scale(scale_dev = out2). */
int wfae_clip_coef(const double* sumsq_parts, int n_parts, float max_norm, float pre_scale, float* out2,
                   wfae_stream_t stream);
/* Loss.calculate_adaptive_weight (experiments/ae_v2_2/train.py:46-52, ae_v2/train.py:46-52):
 * out[0] = clamp(disc_weight * sqrt(sumsq_rec) / (sqrt(sumsq_disc) + 1e-4), 0, 1e4). */
int wfae_adaptive_weight(const double* sumsq_rec, const double* sumsq_disc, float disc_weight, float* out,
                         wfae_stream_t stream);

/* ---- Path-B latent linear forecaster (reference
 * experiments/v1_experiments/pretrained_ae_linear_sevir/train.py:67 `nn.Linear(T_in*C, T_out*C)` per latent pixel,
 * :73-83 training_step).  v is the latent sequence (B,T,C,H*W).
 * latent_diff_pack: X (B*HW, Tin*C)[t*C+c] = v[b,t,c,p] - v[b,Tin-1,c,p]  (the reference's `inp - inp_t` followed
 *   by permute(0,3,4,1,2).reshape(b,h,w,Tin*C), :77-81), Y (B*HW, Tout*C) = the same for the target frames (:78).
 *   The linear layer itself is wfae_linear_fwd / _bwd_weight on X.
 * latent_unpack_add: out (B,Tout,C,HW) = pred (B*HW, Tout*C) re-laid out + last input frame (`pred + inp_t`, :87).
 * mse: loss = mean((pred - target)^2) (F.mse_loss, :82), dpred = gloss * 2 (pred - target) / n. */
int wfae_latent_diff_pack(const float* v, float* X, float* Y, int B, int T, int Tin, int C, int HW,
                          wfae_stream_t stream);
int wfae_latent_unpack_add(const float* pred, const float* v, float* out, int B, int T, int Tin, int C, int HW,
                           wfae_stream_t stream);
int wfae_mse_fwd(const float* pred, const float* target, float* loss, int64_t n, void* ws, size_t ws_bytes,
                 wfae_stream_t stream);
int wfae_mse_bwd(const float* pred, const float* target, const float* gloss, float* dpred, int64_t n,
                 wfae_stream_t stream);

/* ---- sigmoid + L1 loss (ae_64x8x8_lin.py:102 + experiments/ae_v2/train.py:55)
 * recon = sigmoid(h); loss[0] = weight * mean |recon - x|  (fp64 accumulation).
 * bwd: dh = gloss[0] * weight * sign(recon-x) * recon*(1-recon) / n */
int wfae_sigmoid_l1_fwd(const float* h, const float* x, float* recon, float* loss, float weight,
                        int64_t n, void* ws, size_t ws_bytes, wfae_stream_t stream);
int wfae_sigmoid_l1_bwd(const float* recon, const float* x, const float* gloss, float weight,
                        float* dh, int64_t n, wfae_stream_t stream);
/* plain L1 (F.l1_loss) on an existing reconstruction: loss and dL/drecon */
int wfae_l1_fwd(const float* recon, const float* x, float* loss, float weight, int64_t n, void* ws,
                size_t ws_bytes, wfae_stream_t stream);
int wfae_l1_bwd(const float* recon, const float* x, const float* gloss, float weight, float* drecon,
                int64_t n, wfae_stream_t stream);

/* ---- SSIM / PSNR (pytorch_msssim.ssim at experiments/ae_v2/train.py:62;
 * torchmetrics SSIM/PSNR at pipeline/metrics.py:71-84).  11-tap Gaussian
 * sigma 1.5, valid window, K=(0.01,0.03), data_range 1.
 * ssim_fwd: out[0] = mean over images of mean over the (H-10)x(W-10) map.
 * ssim_bwd: dy = gout[0] * d ssim / d y  (gradient w.r.t. the second image).
 * psnr: out[0] = mean_n 10 log10(range_n^2 / mse_n), range_n = max-min of
 * target n; clamp01 != 0 clamps both inputs to [0,1] first (metrics.py:92-93). */
int wfae_ssim_fwd(const float* x, const float* y, float* out, int NB, int H, int W, int clamp01,
                  void* ws, size_t ws_bytes, wfae_stream_t stream);
int wfae_ssim_bwd(const float* x, const float* y, const float* gout, float* dy, int NB, int H, int W,
                  void* ws, size_t ws_bytes, wfae_stream_t stream);
int wfae_psnr(const float* pred, const float* target, float* out, int NB, int HW, int clamp01,
              void* ws, size_t ws_bytes, wfae_stream_t stream);

/* ---- latent transformer of the `_tf` variant (pipeline/models/ae_64x8x8_tf.py:77-80,107-109):
 * nn.TransformerEncoderLayer(d_model=64, nhead=8, dim_feedforward=2048, dropout=0.1), post-norm, ReLU,
 * batch_first=False.  The Linear layers use wfae_linear_*.
 * layernorm: y = LN(x + res) (res may be null); mean/rstd per row are saved for backward; bwd returns the
 *   gradient w.r.t. (x + res) plus dgamma/dbeta.
 * mha_seqfirst: qkv rows are (s, n) pairs (row = s*N + n) with columns [q | k | v], E = H*D each; attention over
 *   the S axis per (n, head), softmax(q k^T / sqrt(D)), dropout on the probabilities from a counter-based
 *   generator keyed by `seed` (the same seed regenerates the mask in backward).  S <= 64, D in {8, 16}.
 * dropout: y = x * mask / (1 - p), mask from (seed, element index); applying it to dy with the same seed is
 *   the backward. */
int wfae_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                       float* mean, float* rstd, int rows, int E, float eps, wfae_stream_t stream);
int wfae_layernorm_bwd(const float* dy, const float* x, const float* res, const float* gamma, const float* mean,
                       const float* rstd, float* dx, float* dgamma, float* dbeta, int rows, int E, int accumulate,
                       void* ws, size_t ws_bytes, wfae_stream_t stream);
/* mha: the same with a layout switch — batch_first = 0: row = s*N + n (above); batch_first = 1: row = n*S + s, i.e. qkv
 * is (N batch elements, S tokens, 3E) as nn.TransformerEncoderLayer(batch_first=True) of AE_ViT_2048 has it
 * (pipeline/models/ae_vit.py:104-109).  S <= 64, D in {8, 16, 64}. */
int wfae_mha_fwd(const float* qkv, float* out, float* probs, int S, int N, int H, int D, int batch_first, float p_drop,
                 uint64_t seed, wfae_stream_t stream);
int wfae_mha_bwd(const float* qkv, const float* probs, const float* dout, float* dqkv, int S, int N, int H, int D,
                 int batch_first, float p_drop, uint64_t seed, wfae_stream_t stream);
int wfae_mha_seqfirst_fwd(const float* qkv, float* out, float* probs, int S, int N, int H, int D, float p_drop,
                          uint64_t seed, wfae_stream_t stream);
int wfae_mha_seqfirst_bwd(const float* qkv, const float* probs, const float* dout, float* dqkv, int S, int N, int H,
                          int D, float p_drop, uint64_t seed, wfae_stream_t stream);
int wfae_relu_fwd(const float* x, float* y, int64_t n, wfae_stream_t stream);
int wfae_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, wfae_stream_t stream);
int wfae_dropout(const float* x, float* y, int64_t n, float p_drop, uint64_t seed, wfae_stream_t stream);

/* ---- Path-B token autoencoder AE_ViT_2048 (pipeline/models/ae_vit.py:84-162), pieces beyond the kernels above.
 * patchify / unpatchify: image (B,C,Hp*P,Wp*P) <-> rows (B*Hp*Wp, C*P*P) — the permutation that turns
 *   Conv2d(C, E, P, P) (:99) and ConvTranspose2d(E, C, P, P) (:128) into linear GEMMs; unpatchify adds bias[c].
 * add_bcast: out[o][i] = x[o][i] + p[i] (positional tokens, :142,158).
 * sq_attn: single-query attention of GlobalCrossEncode (:22-42): q (B, H*D) already projected, kv rows (b, l)
 *   with columns [k | v] (H*D each, as `.view(B, L, 2, nh, dh)` lays them out), softmax(q.k / sqrt(D)) over
 *   the L <= 64 tokens, out (B, H*D), probs (B, H, L) saved for backward. */
int wfae_patchify(const float* img, float* rows, int B, int C, int Hp, int Wp, int P, wfae_stream_t stream);
int wfae_unpatchify(const float* rows, const float* bias, float* img, int B, int C, int Hp, int Wp, int P,
                    wfae_stream_t stream);
int wfae_add_bcast(const float* x, const float* p, float* out, int64_t outer, int64_t inner, wfae_stream_t stream);
/* strided row copy: dst[r][dst_off + c] = src[(r / row_div) * src_ld + src_off + c], c < cols, r < rows; dst rows are
 * dst_ld floats long; zero_fill: the other columns of each dst row are written as 0.  The layout-only steps of
 * AE_ViT_2048.forward (pipeline/models/ae_vit.py:135 `query_vec.expand`, :86 the value slice of a fused k/v projection,
 * :89 one token copied to all positions) and their gradients. */
int wfae_copy_rows(const float* src, float* dst, int64_t rows, int cols, int64_t src_ld, int src_off, int row_div,
                   int dst_ld, int dst_off, int zero_fill, wfae_stream_t stream);
/* sum_mid: out[a][b] = sum_m x[a][m][b] (gradient of a row broadcast over the token axis) */
int wfae_sum_mid(const float* x, float* out, int64_t A, int M, int64_t Bn, wfae_stream_t stream);
int wfae_sq_attn_fwd(const float* q, const float* kv, float* out, float* probs, int B, int L, int H, int D,
                     wfae_stream_t stream);
int wfae_sq_attn_bwd(const float* q, const float* kv, const float* probs, const float* dout, float* dq, float* dkv, int B,
                     int L, int H, int D, wfae_stream_t stream);

/* ---- AdamW (torch.optim.AdamW via pipeline/helpers.py:63-74) ---------------
 * p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
 * p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps),  g pre-multiplied by grad_scale
 * (1/world_size after a sum all-reduce). */
int wfae_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
               float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2,
               float grad_scale, wfae_stream_t stream);
/* out[0] = sum x^2 (fp64 accumulation, written as fp64): grad-norm tracking
 * (pipeline/helpers.py:250-256) */
int wfae_sumsq(const float* x, int64_t n, double* out, void* ws, size_t ws_bytes,
               wfae_stream_t stream);

/* ---- loader contract on device (pipeline/datasets/sevire/sevir.py:749-794,
 * 98-139): uint8 VIL (NB,H,W,T) 'NHWT' -> fp32 (NB,T,H,W) 'NTHW' scaled by
 * 1/255. */
int wfae_vil_u8_to_f32(const uint8_t* src, float* dst, int NB, int H, int W, int T, float scale,
                       wfae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* WFAE_H */
