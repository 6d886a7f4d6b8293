/*
 * tfimm_hip.h -- C ABI of libtfimm_hip.so, the MI355X (gfx950) forward-path kernels that
 * stand in for the TensorFlow ops tfimm's model code calls.
 *
 * The reference (martinsbruveris/tensorflow-image-models) has no native layer: every
 * arithmetic op is a tf.* / tf.keras.layers.* call inside Python model code. Each entry
 * point below therefore replaces a *TF op call site* (cited per function, paths relative
 * to the reference checkout) rather than an existing FFI symbol.  INTEGRATION.md shows the
 * ctypes stub a tfimm maintainer would add.
 *
 * Conventions
 *   - plain C: pointers + ints, no C++/torch types.  All pointers are DEVICE pointers
 *     unless a name ends in _host.  The caller owns every buffer.
 *   - activations: bf16 (uint16 storage), NHWC / (rows, channels) row-major.
 *   - weights are pre-packed by the host side (see tfimm/engine/pack.py):
 *       GEMM/conv weights  Wt[N][ldw]  bf16, K contiguous ("B transposed"),
 *       bias / norm params  fp32.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     null stream).  Return value: 0 on success, otherwise a negative TFIMM_E* code or a
 *     positive hipError_t; tfimm_hip_last_error() returns a static message for the last
 *     failure on the calling thread.  Nothing throws or aborts across the boundary.
 */
#ifndef TFIMM_HIP_H
#define TFIMM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: tfimm_tha_desc grew (proj_dev), tfimm_hip_mlp_fused / tfimm_hip_plan_* / tfimm_hip_ref_* added (round 3) */
/* 4: tfimm_gemm_desc grew (a2 ...: a second A operand, the shortcut convolution folded into a block's last GEMM; round 6) */
#define TFIMM_HIP_ABI_VERSION 4

#if defined(__GNUC__)
#define TFIMM_API __attribute__((visibility("default")))
#else
#define TFIMM_API
#endif

#define TFIMM_EINVAL (-1)   /* bad descriptor (shape/alignment/flag)            */
#define TFIMM_EUNSUP (-2)   /* valid request this build has no kernel for       */

/* activation codes (reference tfimm/layers/factory.py:6-13 act_layer_factory) */
enum {
  TFIMM_ACT_NONE = 0,     /* "linear"                                   */
  TFIMM_ACT_RELU = 1,
  TFIMM_ACT_GELU = 2,     /* exact erf GELU (keras gelu approximate=False) */
  TFIMM_ACT_SWISH = 3,    /* x * sigmoid(x)                             */
  TFIMM_ACT_SIGMOID = 4,
  TFIMM_ACT_RELU6 = 5,
  TFIMM_ACT_TANH = 6
};

/* A-operand addressing modes of tfimm_hip_gemm */
enum {
  TFIMM_A_DENSE = 0,      /* A[m][k] = a[m*lda + k]                                   */
  TFIMM_A_CONV = 1,       /* implicit-GEMM gather from NHWC, Cin % 8 == 0             */
  TFIMM_A_CONV_C4 = 2     /* implicit-GEMM gather from NHWC with Cin == 4 (padded RGB) */
};

TFIMM_API int tfimm_hip_abi_version(void);
TFIMM_API const char* tfimm_hip_last_error(void);
/* Fills name[len] with e.g. "gfx950:sramecc+:xnack-"; returns CU count or <0. */
TFIMM_API int tfimm_hip_device_info(int device, char* name, int len);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_gemm: out[M][N] = epilogue( A[M][K] . Wt[N][K]^T )            (MFMA, bf16->fp32)
 *
 *   epilogue(v)[m][n] = v + bias[n]                      (bias may be NULL)
 *                       -> act(.)  unless act_after_res
 *                       -> + residual[rm][n]             (residual may be NULL;
 *                                                         rm = res_mod ? m % res_mod : m)
 *                       -> act(.)  if act_after_res
 *   stored to out[om][n], om = remap_in ? (m / remap_in) * remap_out + m % remap_in + remap_off : m,
 *   as bf16 (out_f32 == 0) or fp32.
 *
 * Replaces, in the reference:
 *   tf.keras.layers.Dense                  vit.py:155,169  swin.py:167,197  transformers.py:209-212
 *   Conv2D 1x1 / kxk (+ZeroPadding2D) + BatchNormalization(inference, folded) + Activation
 *                                          resnet.py:220-258,315-330,505-512  efficientnet_blocks.py:18-63
 *   PatchEmbeddings conv                   layers/transformers.py:164-165
 *   residual adds                          vit.py:228,234  resnet.py:289  efficientnet_blocks.py:451
 *
 * mode TFIMM_A_CONV:  m = (b*OH + oy)*OW + ox,  k = (ky*KW + kx)*Cin + ci  (Cin % 8 != 0 takes a slow element-load path),
 *     A[m][k] = x[b][oy*stride - pad_t + ky][ox*stride - pad_l + kx][ci]  (0 outside the image)
 * mode TFIMM_A_CONV_C4: Cin == 4, k = (ky*KWp + kx)*4 + ci with KWp = KW rounded up to even
 *     (the packed weight has zero columns for kx >= KW).
 * a_scale (dense mode only, optional): A[m][k] is multiplied by a_scale[(m / rows_per_image)*K + k]
 *     before the product -- the SqueezeExcite gate  x * sigmoid(...)  of
 *     efficientnet_blocks.py:241-248 folded into the projection conv.
 * ------------------------------------------------------------------------------------- */
typedef struct tfimm_gemm_desc {
  const void* a;          /* bf16 */
  const void* wt;         /* bf16 [N][ldw], zero padded to ldw >= K, ldw % 8 == 0 */
  const float* bias;      /* [N] or NULL */
  const void* residual;   /* bf16 [.][ldr] or NULL */
  void* out;              /* bf16 or fp32 [.][ldc] */
  const float* a_scale;   /* fp32 [M/rows_per_image][K] or NULL */
  int32_t M, N, K;
  int32_t lda, ldw, ldr, ldc;
  int32_t out_f32;
  int32_t act, act_after_res;
  int32_t res_mod;
  int32_t remap_in, remap_out, remap_off;
  int32_t mode;
  int32_t B, H, W, Cin, KH, KW, stride, pad_t, pad_l, OH, OW;
  int32_t rows_per_image;
  int32_t tile_hint;      /* 0 = the library's cost model; otherwise a kernel-table index measured by the caller
                             (tfimm/engine/tune.py): 1..6 register-staged tiles, 11..16 one-tile LDS-DMA kernels,
                             21..29 persistent LDS-DMA kernels (28 = 256x256 deep ring, 29 = 256x32).  A hint the
                             problem cannot use (alignment, flavour not built) falls back to the cost model. */
  int32_t stride_w;       /* TFIMM_A_CONV only: horizontal stride if it differs from `stride` (0 = same).
                             Lets a stride-2 RGB stem / patch embedding run on the pixel-PAIR view of a
                             zero-padded 4-channel image ([B][Hp][Wp/2][8], see tfimm_hip_cast_input_pad):
                             vertical stride s, horizontal stride s/2, kernel width ceil(KW/2). */
  int32_t pix_pitch;      /* TFIMM_A_CONV only: elements between consecutive pixels of x if it differs from Cin
                             (0 = Cin).  With `a` pointing at channel c0 of a [B][H][W][C] tensor, Cin = w and
                             pix_pitch = C the convolution reads the channel slice [c0, c0 + w) -- one group of a
                             grouped convolution (resnet.py:229-236); ldc / the `out` pointer place its w output
                             channels the same way. */
  const float* ln_stats;  /* dense mode, no residual: fp32 [M][2] = (mean, rstd) of every row of `a` (tfimm_hip_row_stats).
                             With ln_c1 it folds a LayerNormalization over the K axis INTO this layer
                             (layers/factory.py:42-50 in front of a Dense, vit.py:226-233, swin.py:243-327):
                             out = act(rstd_m * (a . wt^T - mean_m * c1[n]) + bias[n]), where the caller has already
                             multiplied gamma into wt (W' = gamma * W, then rounded to bf16), c1[n] = sum_k W'[k][n] of the
                             ROUNDED weights and bias = beta . W + b.  The normalised tensor is never written. */
  const void* ln_c1;      /* bf16 [N][2][8]: c1[n] as the three-term bf16 split (ca, cb, cc), laid out as the MFMA fragment
                             pair {ca,cb,cc,ca,cb,cc,ca,cb} {cc,0,0,0,0,0,0,0} (tfimm/engine/pack.py: pack_ln_c1) */
  /* ---- ABI v4: a SECOND A operand whose product accumulates into the same output tile (mode TFIMM_A_DENSE or TFIMM_A_CONV for
   *      the first operand -- the 1x1 conv3 of a bottleneck, the 3x3 conv2 of a basic block -- no residual):
   *      out = act( a . wt[:, 0:K]^T + a2' . wt[:, Kp:Kp+K2]^T + bias ),   Kp = K rounded up to 64.
   * It is the shortcut convolution of a residual block folded into the block's last 1x1 convolution (resnet.py:282-290
   * `x = self.conv3(x) ... shortcut = self.downsample(shortcut) ... x += shortcut`, downsample_conv resnet.py:315-330: a 1x1
   * convolution of stride s + BatchNorm): instead of writing the shortcut tensor and reading it back as `residual`, its K2
   * input channels are more k-tiles of the same GEMM (the two folded BatchNorm shifts are added on the host).  a2' is the
   * strided row view of a2: output row m = (b, oy, ox) of an [a2_OH][a2_OW] image reads pixel (b, oy * a2_stride,
   * ox * a2_stride) of the [a2_H][a2_W][lda2] tensor a2 (a2_stride = 1: row m itself).  NULL = no second operand. */
  const void* a2;         /* bf16 [B * a2_H * a2_W][lda2] */
  int32_t K2, lda2;       /* channels of the second operand (multiple of 8) and its pixel pitch in elements */
  int32_t a2_stride, a2_H, a2_W, a2_OH, a2_OW;
  int32_t a2_window;      /* 0 / 1: the 1x1 view above.  w > 1: a w x w window of taps -- output row (b, oy, ox) reads the pixels
                             (b, oy * a2_stride + dy, ox * a2_stride + dx), dy, dx < w, as w * w further operands of K2 channels each
                             (taps in (dy, dx) order, each padded to whole 64-wide k-tiles in wt): the average-pool shortcut of
                             ResNet-D, AveragePooling2D(2, 2) + 1x1 convolution (resnet.py:295-312) = a 2x2 / stride-2 convolution whose
                             taps are the 1x1 kernel / 4.  The window must lie inside the image for every output pixel. */
} tfimm_gemm_desc;

TFIMM_API int tfimm_hip_gemm(const tfimm_gemm_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_row_stats: stats[r] = (mean, 1 / sqrt(var + eps)) over the d channels of row r (fp32 two-pass, population
 * variance -- tf.keras.layers.LayerNormalization's statistics), for a LayerNorm that is folded into the following
 * tfimm_hip_gemm (ln_stats / ln_c1).  x: bf16 rows of x_stride elements.
 * ------------------------------------------------------------------------------------- */
TFIMM_API int tfimm_hip_row_stats(const void* x, float* stats, int64_t rows, int d, int64_t x_stride, float eps, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_cast_input: float32 / bf16 NHWC image batch -> bf16 NHWC with channels padded
 * to c_out (zeros).  in_dtype: 0 = fp32, 1 = bf16.  Replaces Keras' implicit input cast
 * (model(x) accepts float arrays, tests/models/test_factory.py:47-49).
 * ------------------------------------------------------------------------------------- */
TFIMM_API int tfimm_hip_cast_input(const void* in, int in_dtype, void* out, int64_t n_pixels,
                         int c_in, int c_out, void* stream);

/* tfimm_hip_cast_input_pad: as tfimm_hip_cast_input for c_in <= 4 -> 4 stored channels, but writes the
 * image into the interior of a zero border: out[B][H + pad_t + pad_b][W + pad_l + pad_r][4].  This is
 * the ZeroPadding2D / "same" padding in front of a stem convolution (resnet.py:505, layers/conv.py:61)
 * done once while the input is converted, so the convolution itself needs no bounds checks and its
 * operand tiles can be fetched by LDS-DMA. */
TFIMM_API int tfimm_hip_cast_input_pad(const void* in, int in_dtype, void* out, int B, int H, int W, int c_in,
                             int pad_t, int pad_b, int pad_l, int pad_r, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_stem_conv_pool: the ResNet stem in one kernel -- ZeroPadding2D(3) + Conv2D 7x7 stride 2 (BN folded into
 * weights and bias) + ReLU + ZeroPadding2D(1) + MaxPool2D 3x3 stride 2 (resnet.py:505-512, 538-540; forward_features
 * :572-576).  x: the zero-bordered 4-channel image tfimm_hip_cast_input_pad / tfimm_hip_preprocess_input_pad wrote,
 * viewed as pixel pairs [batch][Hp][Wp2][8] bf16 (Wp2 = padded width / 2 <= 116, Hp >= 2 (OH - 1) + 7).
 * wt: [64][ldw] bf16 with k = ky * 32 + (kx / 2) * 8 + (kx % 2) * 4 + c (kx = 7 and c = 3 are zero), the layout of the
 * unfused TFIMM_A_CONV call on the pair view; bias: 64 floats.  OH x OW (OW <= 112): size of the convolution output,
 * which exists only in LDS; out: [batch][PH][PW][64] bf16, PH = (OH - 1) / 2 + 1, PW = (OW - 1) / 2 + 1.
 * ------------------------------------------------------------------------------------- */
typedef struct {
  const void* x;
  const void* wt;
  const float* bias;
  void* out;
  int32_t batch, Hp, Wp2, OH, OW, ldw;
  /* in_dtype 0: x is the padded pair view described above.  1 / 2: x is the caller's own [batch][H][W][3] image in
   * bf16 / float32 and the kernel applies the border (pad_t rows above, pad_l columns left, zeros to Hp x 2 Wp2), the
   * zero 4th channel and the bf16 rounding of tfimm_hip_cast_input_pad itself -- that launch and its 8 bytes per
   * pixel of HBM traffic disappear.  Hp, Wp2 describe the same (virtual) padded geometry in every mode. */
  int32_t in_dtype, H, W, pad_t, pad_l;
} tfimm_stem_desc;

TFIMM_API int tfimm_hip_stem_conv_pool(const tfimm_stem_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_preprocess_input / _pad: the two conversions above for a uint8 image with values in [0, 255], with the
 * model's preprocessing applied on the way: out = bf16(((float)v / 255 - mean[c]) / std[c]) -- the three float32
 * operations of create_preprocessing (models/factory.py:165-167; mean/std: utils/constants.py:3-6, tiled to c_in by
 * the caller as in factory.py:153-163), so the result is bit-identical to preprocessing in float32 on the host
 * followed by tfimm_hip_cast_input.  mean, std: HOST arrays of c_in floats (copied into the launch),
 * c_in <= TFIMM_PREPROCESS_MAX_CHANNELS (<= 4 for _pad); padded channels and the border are 0.
 * ------------------------------------------------------------------------------------- */
#define TFIMM_PREPROCESS_MAX_CHANNELS 8
TFIMM_API int tfimm_hip_preprocess_input(const void* in, void* out, int64_t n_pixels, int c_in, int c_out,
                               const float* mean, const float* std, void* stream);
TFIMM_API int tfimm_hip_preprocess_input_pad(const void* in, void* out, int B, int H, int W, int c_in,
                                   int pad_t, int pad_b, int pad_l, int pad_r,
                                   const float* mean, const float* std, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_layernorm: y[r][:] = (x[r][:] - mean) * rsqrt(var + eps) * gamma + beta,
 * population variance, fp32 statistics.  x row r starts at x + r*x_stride (elements),
 * y row r at y + r*y_stride.  Replaces tf.keras.layers.LayerNormalization
 * (layers/factory.py:42-50; call sites vit.py:222,231,452  swin.py:295,322,504).
 * ------------------------------------------------------------------------------------- */
TFIMM_API int tfimm_hip_layernorm(const void* x, void* y, const float* gamma, const float* beta,
                        int64_t rows, int d, int64_t x_stride, int64_t y_stride, float eps,
                        void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_attention: fused softmax(scale * Q K^T [+ bias] [+ mask]) V per (sequence, head),
 * reading q/k/v straight out of the packed QKV projection  qkv[row][3][heads][hd]  and
 * writing out[row][heads*hd].  fp32 softmax, bf16 P.  Head dims: global attention 1..128 (vit_huge_patch14: 80),
 * windows 1..64; anything else returns TFIMM_EUNSUP.
 *
 * window == 0: global attention, sequence s = image, token t -> row s*n_tokens + t.
 *     Replaces vit.py:156-167 (reshape/transpose, scale*matmul, softmax, matmul, merge).
 * window  > 0: Swin (shifted-)window attention on a res_h x res_w token grid, sequence =
 *     (image, window); token (ty,tx) of window (wy,wx) is grid position
 *     ((wy*window+ty+shift) % res_h, (wx*window+tx+shift) % res_w) -- i.e. tf.roll(-shift),
 *     window_partition, window_reverse and tf.roll(+shift) (swin.py:72-108,295-318) are
 *     folded into the load/store index map.  rel_bias[heads][n][n] (fp32, n = window^2) is
 *     the gathered relative_position_bias (swin.py:175-184); when shift > 0 the -100 mask of
 *     swin.py:249-273 is recomputed from region ids.
 * ------------------------------------------------------------------------------------- */
typedef struct tfimm_attn_desc {
  const void* qkv;        /* bf16 [rows][3*heads*hd] */
  void* out;              /* bf16 [rows][heads*hd]   */
  const float* rel_bias;  /* fp32 [heads][n][n] or NULL */
  int32_t batch;          /* images */
  int32_t n_tokens;       /* tokens per image (global) / res_h*res_w (window) */
  int32_t heads, hd;
  float scale;
  int32_t window, shift, res_h, res_w;
  const float* bias_log2; /* optional (window > 0): rel_bias with the shift mask already added and everything
                             multiplied by log2(e), one tile per window kind:
                             [kinds][heads][n][ceil64(n)], kinds = 1 (shift == 0) or 4 (shift > 0; kind =
                             2 * (last window row) + (last window column) -- the only four distinct
                             patterns of the mask of swin.py:249-273).  Built once on the host
                             (tfimm/engine/pack.py: swin_bias_tiles); when NULL the kernel combines
                             rel_bias and the mask itself for every workgroup. */
} tfimm_attn_desc;

TFIMM_API int tfimm_hip_attention(const tfimm_attn_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_talking_heads_attention: CaiT self-attention with the two head-mixing Dense layers
 * around the softmax (TalkingHeadAttention.call, cait.py:233-262):
 *     s[b,h,i,j]  = scale * q[b,i,h,:] . k[b,j,h,:]
 *     a[b,h',i,j] = softmax_j( sum_h s[b,h,i,j] * proj_l_w[h][h'] + proj_l_b[h'] )
 *     w[b,g,i,j]  = sum_h' a[b,h',i,j] * proj_w_w[h'][g] + proj_w_b[g]
 *     out[b,i,g,:] = sum_j w[b,g,i,j] * v[b,j,g,:]
 * qkv packed as for tfimm_hip_attention (bf16 [rows][3*heads*hd], rows = batch * n_tokens);
 * proj_*_w / proj_*_b are the Keras kernels [heads_in][heads_out] and biases in fp32 and are HOST
 * pointers (heads <= 16): the library copies them into the kernel's argument segment at launch, so
 * they are read with scalar loads -- they are layer weights, known on the host when the plan is built.  MFMA kernel for hd in {32, 48} (every
 * CaiT configuration has hd = 48) and heads in {1, 2, 3, 4, 6, 8, 16}; any other shape takes a plain
 * fp32 kernel (one workgroup per query row) as long as 2 * heads * n_tokens floats fit in LDS.
 * ------------------------------------------------------------------------------------- */
typedef struct tfimm_tha_desc {
  const void* qkv;
  void* out;                /* bf16 [rows][heads*hd] */
  const float* proj_l_w;    /* HOST fp32 [heads][heads] */
  const float* proj_l_b;    /* HOST fp32 [heads] */
  const float* proj_w_w;
  const float* proj_w_b;
  int32_t batch, n_tokens, heads, hd;
  float scale;
  const float* proj_dev;    /* optional DEVICE fp32 [proj_l_w | proj_l_b | proj_w_w | proj_w_b] (2 * (heads^2 + heads) values): the
                               MFMA kernel then takes the mixing layers from there instead of its argument segment (what
                               plans do; results are the same either way) */
} tfimm_tha_desc;

TFIMM_API int tfimm_hip_talking_heads_attention(const tfimm_tha_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_class_attention: the class token of every image attends to all of its tokens
 * (ClassAttention.call, cait.py:118-146 between the q/k/v and proj layers):
 *     out[b, h*hd + d] = sum_j softmax_j( q[b, h, :] . k[b, j, h, :] ) * v[b, j, h, d]
 * q: bf16, one row per image with row stride ldq (ALREADY scaled: the caller folds
 * (D/H)^-0.5 into the q layer); kv: bf16 [B*n_tokens][ldkv] holding k in columns
 * [0, heads*hd) and v in [heads*hd, 2*heads*hd); out: bf16, one row per image, stride ldo.
 * ------------------------------------------------------------------------------------- */
TFIMM_API int tfimm_hip_class_attention(const void* q, const void* kv, void* out, int B, int n_tokens, int heads,
                              int hd, int ldq, int ldkv, int ldo, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_copy_rows: dst[b][dst_row0 + r][:] = src[b][r][:] for r < src_rows -- one input of
 * a tf.concat along the token axis (cait.py:425-426: class token in front of the patch tokens).
 * bf16 rows of d elements (16-byte vectors when d % 8 == 0 and both buffers are 16-byte aligned).
 * ------------------------------------------------------------------------------------- */
TFIMM_API int tfimm_hip_copy_rows(const void* src, void* dst, int B, int src_rows, int dst_rows, int dst_row0,
                        int d, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_maxpool: k x k / stride max pool with symmetric zero padding `pad`, NHWC bf16.
 * Padding contributes ZEROS (the reference pads with ZeroPadding2D and pools VALID,
 * resnet.py:538-540), not -inf.
 * ------------------------------------------------------------------------------------- */
TFIMM_API int tfimm_hip_maxpool(const void* x, void* y, int B, int H, int W, int C, int k, int stride,
                      int pad, int OH, int OW, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_mean_rows: y[b][c] = mean_r x[b][r][c], r < R.  out_f32: 0 bf16, 1 fp32.
 * GlobalAveragePooling2D / 1D (layers/classifier.py:35, swin.py:506) and the SE squeeze
 * (efficientnet_blocks.py:242).
 * ------------------------------------------------------------------------------------- */
TFIMM_API int tfimm_hip_mean_rows(const void* x, void* y, int B, int R, int C, int out_f32, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_bcast_rows: dst[(b*dst_rows_per_image + t)][:] = src[t][:], t < n_rows, b < B.
 * Writes the (cls [, dist]) token rows, already summed with their pos_embed rows on the
 * host (vit.py:427-434 tf.repeat / tf.concat / + pos_embed).
 * ------------------------------------------------------------------------------------- */
TFIMM_API int tfimm_hip_bcast_rows(const void* src, void* dst, int B, int n_rows, int d,
                         int dst_rows_per_image, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_dwconv: depthwise k x k conv, stride s, explicit (pad_t, pad_l) zero padding,
 * NHWC bf16, folded-BN scale already in w, + bias, + activation.  w: fp32 [k*k][C].
 * Optionally accumulates per-(image, channel) sums of the OUTPUT into sum_out -- the SE squeeze fused into the
 * producer: int64 [B][C] FIXED-POINT accumulators in units of 2^-20 (zeroed by the caller; read them with
 * tfimm_hip_se_gate(sums_fixed = 1)).  Integer adds commute, so the sums -- and everything computed from them -- are
 * bit-identical from launch to launch whatever order the workgroups arrive in (float atomics were not), and from batch
 * size to batch size: a thread converts the partial sum of every finished output row (four pixels of one channel, rounded
 * to 2^-16; partials beyond +-32768 saturate) before it adds, so the way a launch splits an image into row segments
 * does not show in the result.
 * Replaces DepthwiseConv2D + BatchNormalization + Activation
 * (efficientnet_blocks.py:350-352,443-445; convnext.py:191-197 with act none).
 * ------------------------------------------------------------------------------------- */
TFIMM_API int tfimm_hip_dwconv(const void* x, const float* w, const float* bias, void* y, void* sum_out,
                     int B, int H, int W, int C, int k, int stride, int pad_t, int pad_l,
                     int OH, int OW, int act, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_se_gate: gate[b][c] = gate_act( W2 . act( W1 . mean[b] + b1 ) + b2 )[c]
 * with mean[b][c] = sums[b][c] * inv_count.  sums: fp32 [B][C] (sums_fixed = 0: e.g. the means of tfimm_hip_mean_rows
 * with inv_count = 1) or the int64 fixed-point accumulators of tfimm_hip_dwconv / tfimm_hip_expand_dwconv (sums_fixed = 1).
 * w1: fp32 [rd][C] (reduce conv, transposed), w2: fp32 [rd][C]
 * (expand conv as Keras stores it) -- both are walked along C by consecutive threads.
 * SqueezeExcite.call (efficientnet_blocks.py:241-248) / SEModule.call (layers/attention.py:66-74)
 * minus the final multiply, which is fused into the consumer (a_scale of tfimm_hip_gemm)
 * or done by tfimm_hip_scale_channels.
 * ------------------------------------------------------------------------------------- */
TFIMM_API int tfimm_hip_se_gate(const void* sums, int sums_fixed, float inv_count, const float* w1, const float* b1,
                      const float* w2, const float* b2, float* gate, int B, int C, int rd,
                      int act, int gate_act, void* stream);

/* y[b][r][c] = x[b][r][c] * gate[b][c] (+ residual, then relu if act_after != 0) */
TFIMM_API int tfimm_hip_scale_channels(const void* x, const float* gate, const void* residual, void* y,
                             int B, int R, int C, int act_after, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_patch_merge_ln: Swin PatchMerging front half (swin.py:352-359): gather the 2x2
 * neighbourhood in the order (0,0),(1,0),(0,1),(1,1) -> 4C channels, LayerNorm(4C).
 * x: bf16 [B][H*W][C]; y: bf16 [B][(H/2)*(W/2)][4C].
 * ------------------------------------------------------------------------------------- */
TFIMM_API int tfimm_hip_patch_merge_ln(const void* x, void* y, const float* gamma, const float* beta,
                             int B, int H, int W, int C, float eps, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_conv_chain: the tail of a ResNet bottleneck block in ONE launch,
 *     mid = act1( conv_{KH x KW, stride, pad}(x) * W1 + b1 )            [M][C1]   never written to memory
 *     out = act2( mid . W2^T + b2 + residual )                          [M][N2]
 * i.e. pad2 / conv2 / bn2 / act2 / conv3 / bn3 / += shortcut / act3 of Bottleneck.call (resnet.py:273-290), BatchNorm
 * folded into (W, b) by the host.  x: bf16 NHWC [B][H][W][Cin], Cin % 64 == 0; w1: bf16 [C1][ldw1 = KH*KW*Cin] in
 * (ky, kx, ci) order; w2: bf16 [N2][ldw2] with its K (= C1) axis PERMUTED: within every 16 channels the two middle
 * quads are swapped (k-slot 16t + s holds channel 16t + {0..3, 8..11, 4..7, 12..15}[s]) -- the order in which a wave's
 * GEMM-1 accumulators become the register operand of GEMM 2 (tfimm/engine/pack.py chain_k_order); residual (may be
 * NULL) / out: bf16 rows of ldr / ldc elements (multiples of 8); M = B*OH*OW.  Built for the ResNet stage-1 shape --
 * 3x3 / stride 1 / pad 1, Cin = C1 = 64, W <= 63, N2 in {256, 512} (csrc/gemm_chain_kernel.h) -- and the stage-2 shape --
 * Cin = C1 = 128, W <= 31, N2 in {256, 512}, no shortcut convolution (csrc/conv_strip.hip); anything else returns TFIMM_EUNSUP and the caller
 * runs the two convolutions as two tfimm_hip_gemm launches (same arithmetic: fp32 accumulation, the intermediate
 * rounded to bf16 once; only the summation order inside GEMM 2's 16-wide k-steps differs).
 * ------------------------------------------------------------------------------------- */
typedef struct tfimm_chain_desc {
  const void* x;
  const void* w1;
  const float* b1;
  const void* w2;
  const float* b2;
  const void* residual;
  void* out;
  int32_t B, H, W, Cin, KH, KW, stride, pad_t, pad_l, OH, OW;
  int32_t C1, N2;
  int32_t ldw1, ldw2, ldr, ldc;
  int32_t act1, act2;       /* TFIMM_ACT_*; act2 is applied after the residual add */
  const void* ds_x;         /* optional: the shortcut is a 1x1 convolution + BN of the block INPUT (first block of a stage,
                               resnet.py:315-330): bf16 [M][ds_cin], ds_cin = 64.  Its product accumulates into the second
                               GEMM's accumulators -- out = act2(mid . W2^T + ds_x . Wds^T + b2), the caller adding the
                               shortcut's folded-BN shift to b2 -- so neither that launch nor its tensor exists.
                               residual must be NULL, activations relu. */
  const void* ds_w;         /* its weights (BN scale folded) as MFMA fragments: bf16 [N2/32][4][64][8], element
                               [blk][t][lane][e] = Wds[k = 16 t + 8 (lane >> 5) + e][n = 32 blk + (lane & 31)] */
  int32_t ds_cin;
} tfimm_chain_desc;

TFIMM_API int tfimm_hip_conv_chain(const tfimm_chain_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_expand_dwconv: the front half of an inverted-residual block in one launch --
 *   conv_pw (1x1, Cin -> C) + bn1 + act1, then conv_dw (k x k depthwise, stride) + bn2 + act2, optionally the
 *   squeeze sums of the result (efficientnet_blocks.py:438-445, InvertedResidual.call up to `se`).
 * The expanded tensor exists only in LDS, rounded to bf16 exactly where the two-launch path (tfimm_hip_gemm +
 * tfimm_hip_dwconv) rounds the tensor it stores; depthwise accumulation is fp32.
 *   x    bf16 [B][H][W][Cin], Cin % 8 == 0, Cin <= 32
 *   w1   expand weights (BN scale folded) as MFMA fragments: bf16 [Cpad/32][2][64][8], element
 *        [cc][ks][lane][j] = W1[k = 16 ks + 8 (lane >> 5) + j][c = 32 cc + (lane & 31)], zero for k >= Cin, c >= C
 *   b1, b2  fp32 [Cpad] folded BN shifts;  wdw  fp32 [k*k][Cpad] depthwise taps (BN scale folded), zero padded
 *   y    bf16 [B][OH][OW][C];  sum_out  int64 [B][C] fixed-point squeeze sums as in tfimm_hip_dwconv (zeroed by the caller) or NULL
 * k in {3, 5}, stride in {1, 2}, explicit top / left zero padding of the EXPANDED tensor (bottom / right follow from
 * OH, OW); at stride 1 at most 512 expanded channels (their bias is staged in LDS once); one image of the output below 2 GiB.
 * Anything else returns TFIMM_EUNSUP and the caller runs the two launches.
 * ------------------------------------------------------------------------------------- */
typedef struct tfimm_expand_dw_desc {
  const void* x;
  const void* w1;
  const float* b1;
  const float* wdw;
  const float* b2;
  void* y;
  void* sum_out;
  int32_t B, H, W, Cin, C, Cpad, k, stride, pad_t, pad_l, OH, OW;
  int32_t act1, act2;
  int32_t stem;           /* 1: the "expansion" is the network's 3 x 3 / stride 2 RGB stem convolution (conv_stem + bn1 + act,
                             efficientnet.py:300-302) in front of the first block's depthwise layer: x is the zero-bordered
                             4-channel image [B][img_h][img_w][4] of tfimm_hip_cast_input_pad (the convolution itself pads
                             nothing), Cin = 4, H x W the convolution's OUTPUT size, k = 3 / stride = 1 the depthwise layer,
                             and w1 is bf16 [Cpad/32][3][64][8] with element [cc][ks][lane][j] = W[tap][c][32 cc + (lane & 31)],
                             tap = 4 ks + 2 (lane >> 5) + j / 4, c = j % 4 (zero for tap >= 9, c >= 3) */
  int32_t img_h, img_w;
} tfimm_expand_dw_desc;

TFIMM_API int tfimm_hip_expand_dwconv(const tfimm_expand_dw_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_attention_probs: probs[b][h][i][j] = softmax_j(scale * q[b,i,h,:] . k[b,j,h,:]) in fp32 --
 * the attention map ViTMultiHeadAttention.call returns as features["attn"] when return_features=True
 * (vit.py:160-163; ViT.forward_features stores it as "block_<j>/attn", vit.py:447-450).  qkv: bf16
 * [B*n_tokens][3*heads*hd] as tfimm_hip_attention reads it.  Feature path only: the plain forward never
 * materialises the map.
 * ------------------------------------------------------------------------------------- */
TFIMM_API int tfimm_hip_attention_probs(const void* qkv, void* probs, int B, int n_tokens, int heads, int hd,
                              float scale, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_group_norm: GroupNormalization over NHWC (layers/norm.py:37-165, group_normalize): per (image, group)
 * mean / population variance over (H, W, C/groups), y = x*inv + (beta - mean*inv), inv = rsqrt(var+eps)*gamma
 * (tf.nn.batch_normalization), then act, then (+ residual, act_after_res) when residual != NULL -- the
 * norm + activation + shortcut add of a ResNet block whose norm_layer is "group_norm" (resnet.py:269-290).
 * x / residual / y: bf16 [B][rows][C]; gamma / beta: fp32 [C]; stats_ws: int64 [B][groups][2] scratch (zeroed here): sum and
 * sum of squares in 2^-20 fixed point, so the statistics do not depend on the order the workgroups add them in.
 * ------------------------------------------------------------------------------------- */
TFIMM_API int tfimm_hip_group_norm(const void* x, const float* gamma, const float* beta, const void* residual, void* y,
                         void* stats_ws, int B, int rows, int C, int groups, float eps, int act,
                         int act_after_res, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_blur_pool: BlurPool2D(kernel_size=3, stride) (layers/blurpool.py:5-66): tf.pad(REFLECT) by
 * p = (3 + stride) / 2 - 1, then the depthwise [1 2 1] x [1 2 1] / 16 filter at `stride`, VALID.
 * x: bf16 [B][H][W][C] -> y: bf16 [B][OH][OW][C], OH = (H + 2p - 3) / stride + 1.
 * ------------------------------------------------------------------------------------- */
TFIMM_API int tfimm_hip_blur_pool(const void* x, void* y, int B, int H, int W, int C, int stride, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_avg_pool: AveragePooling2D(pool_size=k, strides=stride, padding="same") (resnet.py:299-301):
 * OH = ceil(H / stride); windows clipped at the border average over their VALID elements only.
 * ------------------------------------------------------------------------------------- */
TFIMM_API int tfimm_hip_avg_pool(const void* x, void* y, int B, int H, int W, int C, int k, int stride, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_eca_gate: gate[b][c] = gate_act( sum_t w[t] * mean[b][c + t - (k-1)/2] ), zero padded over the
 * channel axis, mean = sums * inv_count -- EcaModule.call between the channel mean and the final multiply
 * (ZeroPadding1D + Conv1D(1, k, no bias) + gate, layers/attention.py:110-126).  All fp32; w: [k].
 * ------------------------------------------------------------------------------------- */
TFIMM_API int tfimm_hip_eca_gate(const float* sums, float inv_count, const float* w, float* gate, int B, int C, int k,
                       int gate_act, void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_grouped_conv3x3: ZeroPadding2D(1) + Conv2D(3x3, stride, groups) (+ folded BatchNorm + activation) for
 * groups of at most 32 channels with equal input and output width -- ResNeXt's conv2 (resnet.py:229-241).  Runs as
 * 32-channel super-groups: 32 / (C / groups) times the useful multiply-accumulates instead of `groups` times for the
 * dense block-diagonal expansion.  x / y: bf16 NHWC, C % 32 == 0; wfrag: the kernel packed per MFMA lane by
 * tfimm/engine/pack.py pack_grouped3x3 ([C / 32][18][64] 16-byte fragments); bias: fp32 [C].
 * ------------------------------------------------------------------------------------- */
TFIMM_API int tfimm_hip_grouped_conv3x3(const void* x, const void* wfrag, const float* bias, void* y, int B, int H, int W,
                              int C, int stride, int act, void* stream);

/* hipMemsetAsync on the library's own HIP runtime (zeroing accumulation buffers such as the
 * dwconv sum_out) -- avoids a second runtime instance being loaded by the host language. */
TFIMM_API int tfimm_hip_memset_async(void* dst, int value, size_t bytes, void* stream);

/* Elementwise y = act(x + bias?)  -- used by paths with no producer to fuse into. */
TFIMM_API int tfimm_hip_bias_act(const void* x, const float* bias, void* y, int64_t rows, int C, int act,
                       void* stream);

/* ---------------------------------------------------------------------------------------
 * tfimm_hip_mlp_fused: out = residual + fc2( act( fc1( LayerNormalization(x) ) ) ) in ONE launch -- norm2 / mlp.fc1 / GELU /
 * mlp.fc2 / shortcut add of a Swin block (swin.py:322-325, layers/transformers.py:208-214), norm / fc1 / GELU / fc2 (/ LayerScale,
 * folded into w2 / b2) / shortcut add of a ConvNeXt block (convnext.py:226-232).  The hidden tensor is never written: the bf16-packed
 * accumulators of the first GEMM are the register operand of the second (csrc/mlp.hip).  Built for C = 128, hidden = 512.
 *   x, residual, out  bf16 [M][C] (residual may alias x)
 *   w1   bf16 [hidden][C], LayerNorm gamma folded in (W1' = gamma * W1, then rounded);  b1 fp32 [hidden] = beta . W1 + b1;  the
 *        kernel normalises the rows itself: (x - mean) * rstd (two-pass statistics, eps), rounded to bf16, is what W1' multiplies
 *   w2   bf16 [C][hidden] with its K axis in tfimm/engine/pack.py chain_k_order (within every 16 channels the two middle quads
 *        swapped);  b2 fp32 [C]
 * Anything else returns TFIMM_EUNSUP and the caller runs tfimm_hip_row_stats + two tfimm_hip_gemm launches.
 * ------------------------------------------------------------------------------------- */
typedef struct tfimm_mlp_desc {
  const void* x;
  const void* w1;
  const float* b1;
  const void* w2;
  const float* b2;
  const void* residual;
  void* out;
  int64_t M;
  int32_t C, hidden, act;
  float eps;
} tfimm_mlp_desc;
TFIMM_API int tfimm_hip_mlp_fused(const tfimm_mlp_desc* d, void* stream);

/* =======================================================================================
 * FLOAT32 VERIFICATION PATH (csrc/ref32.hip; selected by TFIMM_PRECISION=fp32, tfimm/engine/precision.py)
 *
 * The reference is float32 end to end and pins values at 1e-3 relative to the maximum (tests/test_timm.py:71).  The
 * entry points below run the SAME layer program as the bf16 kernels above -- same host-side lowering and weight
 * transformations, minus the cross-layer fusions -- with float32 activations, float32 GEMM weights (Wt[N][ldw] float)
 * and float32 accumulation, so that the engine's arithmetic can be held to the reference's own bar.  Plain kernels
 * (one thread / wave per output), 20-50x slower than the product path: a checker, not a fallback -- nothing selects
 * them unless the caller asks for fp32.  Every signature is the bf16 entry point's with `bf16` tensors replaced by
 * `float` (descriptor fields keep their meaning; tile_hint / ln_* / bias_log2 are ignored or refused).
 * tfimm_hip_se_gate and tfimm_hip_eca_gate are float32 already and serve both paths.
 * ======================================================================================= */
TFIMM_API int tfimm_hip_ref_gemm(const tfimm_gemm_desc* d, void* stream);   /* TFIMM_A_DENSE and TFIMM_A_CONV (any Cin) */
/* in_dtype: 0 float32, 1 bf16, 2 uint8 with out = ((float)v / 255 - mean[c]) / std[c] (models/factory.py:165-167);
 * mean / std: HOST arrays of c_in floats for uint8, NULL otherwise.  out: float32 [n_pixels][c_out], channels >= c_in zero. */
TFIMM_API int tfimm_hip_ref_cast_input(const void* in, int in_dtype, void* out, int64_t n_pixels, int c_in, int c_out,
                                       const float* mean, const float* std, void* stream);
TFIMM_API int tfimm_hip_ref_layernorm(const void* x, void* y, const float* gamma, const float* beta, int64_t rows, int d,
                                      int64_t x_stride, int64_t y_stride, float eps, void* stream);
TFIMM_API int tfimm_hip_ref_patch_merge_ln(const void* x, void* y, const float* gamma, const float* beta, int B, int H, int W,
                                           int C, float eps, void* stream);
TFIMM_API int tfimm_hip_ref_copy_rows(const void* src, void* dst, int B, int src_rows, int dst_rows, int dst_row0, int d,
                                      void* stream);
TFIMM_API int tfimm_hip_ref_bcast_rows(const void* src, void* dst, int B, int n_rows, int d, int dst_rows_per_image,
                                       void* stream);
TFIMM_API int tfimm_hip_ref_mean_rows(const void* x, void* y, int B, int R, int C, int out_f32, void* stream);
TFIMM_API int tfimm_hip_ref_scale_channels(const void* x, const float* gate, const void* residual, void* y, int B, int R,
                                           int C, int act_after, void* stream);
TFIMM_API int tfimm_hip_ref_maxpool(const void* x, void* y, int B, int H, int W, int C, int k, int stride, int pad, int OH,
                                    int OW, void* stream);
TFIMM_API int tfimm_hip_ref_avg_pool(const void* x, void* y, int B, int H, int W, int C, int k, int stride, void* stream);
TFIMM_API int tfimm_hip_ref_blur_pool(const void* x, void* y, int B, int H, int W, int C, int stride, void* stream);
/* sum_out must be NULL: on this path the SqueezeExcite mean is a tfimm_hip_ref_mean_rows launch */
TFIMM_API int tfimm_hip_ref_dwconv(const void* x, const float* w, const float* bias, void* y, void* sum_out, int B, int H,
                                   int W, int C, int k, int stride, int pad_t, int pad_l, int OH, int OW, int act,
                                   void* stream);
TFIMM_API int tfimm_hip_ref_group_norm(const void* x, const float* gamma, const float* beta, const void* residual, void* y,
                                       void* stats_ws, int B, int rows, int C, int groups, float eps, int act,
                                       int act_after_res, void* stream);
TFIMM_API int tfimm_hip_ref_attention(const tfimm_attn_desc* d, void* stream);
TFIMM_API int tfimm_hip_ref_attention_probs(const void* qkv, void* probs, int B, int n_tokens, int heads, int hd, float scale,
                                            void* stream);
TFIMM_API int tfimm_hip_ref_talking_heads_attention(const tfimm_tha_desc* d, void* stream);
TFIMM_API int tfimm_hip_ref_class_attention(const void* q, const void* kv, void* out, int B, int n_tokens, int heads, int hd,
                                            int ldq, int ldkv, int ldo, void* stream);

/* =======================================================================================
 * PROGRAM-LEVEL ENTRY POINTS (csrc/plan.hip): a whole forward behind three calls, for hosts without Python.
 *
 * Lowering a model configuration to the call sequence above (weight packing, buffer plan, tile selection) is host logic
 * (tfimm/engine/graph.py).  `Plan.export()` serialises the FINISHED plan of one (model, batch size) -- every call with its
 * arguments, the packed constants, the slab sizes, the named outputs -- into a self-contained blob; these functions run it
 * by calling the very same op-level entry points with the very same arguments (bit-identical results).  Conventions as
 * everywhere else: the caller owns the device memory (ONE workspace of tfimm_plan_info.workspace_bytes, 256-byte aligned),
 * every launch is asynchronous on `stream` and capturable into a hipGraph, nothing allocates on the device.
 *
 *   tfimm_hip_plan_query    parse the blob: workspace size, batch, input geometry
 *   tfimm_hip_plan_create   upload the constants into `workspace` (synchronises `stream` once; the blob may be freed
 *                           afterwards) and resolve every pointer of the call list
 *   tfimm_hip_plan_forward  input: [batch][in_h][in_w][in_c] NHWC, in_dtype 0 = float32, 1 = bf16 (device pointer)
 *   tfimm_hip_plan_output   where a named result lies ("logits", or a feature name of a plan exported with features):
 *                           device pointer into the workspace, rows (= batch * rows per image), columns, dtype (0 bf16, 1 f32)
 * A plan object is not re-entrant (one forward at a time); distinct plans are independent.
 * ======================================================================================= */
typedef void* tfimm_plan_t;
typedef struct tfimm_plan_info {
  uint64_t workspace_bytes;
  int32_t batch, in_h, in_w, in_c;
  int32_t n_calls, n_outputs;
} tfimm_plan_info;
TFIMM_API int tfimm_hip_plan_query(const void* blob, size_t bytes, tfimm_plan_info* info);
TFIMM_API int tfimm_hip_plan_create(const void* blob, size_t bytes, void* workspace, void* stream, tfimm_plan_t* plan);
TFIMM_API int tfimm_hip_plan_forward(tfimm_plan_t plan, const void* input, int in_dtype, void* stream);
TFIMM_API int tfimm_hip_plan_output(tfimm_plan_t plan, const char* name, void** ptr, int64_t* rows, int64_t* cols, int* dtype);
TFIMM_API int tfimm_hip_plan_destroy(tfimm_plan_t plan);

#ifdef __cplusplus
}
#endif
#endif /* TFIMM_HIP_H */
