// Small streaming kernels outside the GEMM / attention families (gfx950):
//   tfimm_hip_attention_probs  softmax(scale * Q K^T) materialised in fp32 -- the "block_j/attn" entries of
//                              ViT's feature dictionary (vit.py:160-163); never on the plain forward path
//   tfimm_hip_group_norm       GroupNormalization over NHWC (layers/norm.py:37-165) + activation + residual
//   tfimm_hip_blur_pool        BlurPool2D: reflect padding + binomial 3x3 depthwise filter (layers/blurpool.py:5-66)
//   tfimm_hip_avg_pool         AveragePooling2D(padding="same"): clipped border windows (resnet.py:299-301)
//   tfimm_hip_eca_gate         EcaModule gate: Conv1D over the channel axis of the channel means + sigmoid
//                              (layers/attention.py:105-126), fp32 throughout
#include "common.h"

// =====================================================================================================================
// attention probabilities
// =====================================================================================================================
// One wave per (image, head, query row): lanes own keys j = lane, lane + 64, ...; the fp32 output row doubles as the
// score buffer between the three passes (scores + max, exp + sum, normalise).
__global__ __launch_bounds__(256) void attn_probs_kernel(const bf16_t* __restrict__ qkv, float* __restrict__ out, int n,
                                                         int heads, int hd, float scale, int64_t total_rows) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= total_rows) return;
  const int i = (int)(row % n);
  const int64_t bh = row / n;
  const int h = (int)(bh % heads);
  const int64_t b = bh / heads;
  const int D = heads * hd, ld = 3 * D;
  const bf16_t* q = qkv + ((b * n + i) * (int64_t)ld + h * hd);
  const bf16_t* kbase = qkv + (b * n * (int64_t)ld + D + h * hd);
  float* o = out + row * n;
  float mx = -__builtin_inff();
  const bool vec = (hd & 7) == 0 && (D & 7) == 0;
  for (int j = lane; j < n; j += 64) {
    const bf16_t* k = kbase + (int64_t)j * ld;
    float acc = 0.f;
    if (vec) {
      for (int d = 0; d < hd; d += 8) {
        float qf[8], kf[8];
        unpack8(*(const uint4*)(q + d), qf);
        unpack8(*(const uint4*)(k + d), kf);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf(qf[e], kf[e], acc);
      }
    } else {
      for (int d = 0; d < hd; ++d) acc = fmaf(bf2f(q[d]), bf2f(k[d]), acc);
    }
    acc *= scale;                       // vit.py:160: the scale multiplies the product
    o[j] = acc;
    mx = fmaxf(mx, acc);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) {
    const float e = __expf(o[j] - mx);
    o[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int j = lane; j < n; j += 64) o[j] *= inv;
}

extern "C" int tfimm_hip_attention_probs(const void* qkv, void* probs, int B, int n_tokens, int heads, int hd, float scale,
                                         void* stream) {
  if (!qkv || !probs) TFIMM_FAIL(TFIMM_EINVAL, "attention_probs: null pointer");
  if (B <= 0 || n_tokens <= 0 || heads <= 0 || hd <= 0) TFIMM_FAIL(TFIMM_EINVAL, "attention_probs: bad shape");
  const int64_t rows = (int64_t)B * heads * n_tokens;
  const int64_t blocks = (rows + 3) / 4;
  if (blocks > 0x7fffffffLL) TFIMM_FAIL(TFIMM_EINVAL, "attention_probs: grid too large");
  TFIMM_LAUNCH(attn_probs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv,
               (float*)probs, n_tokens, heads, hd, scale, rows);
  return 0;
}

// =====================================================================================================================
// group normalisation
// =====================================================================================================================
// Pass 1: per (image, group) sum and sum of squares.  A workgroup takes a run of rows of ONE image; a thread keeps one
// channel vector (V = 8 or 1 channels) in registers across its rows, the per-channel totals meet in LDS, the G group
// totals leave with one global atomic pair per group and workgroup.
template <int V>
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x, tfimm_sq_t* __restrict__ stats, int rows,
                                                       int C, int G, int rows_per_block) {
  // totals in 64-bit fixed point (common.h, squeeze sums): integer adds commute, so the statistics -- and the network's
  // output -- do not depend on the order in which threads and workgroups arrive
  extern __shared__ __attribute__((aligned(16))) tfimm_sq_t gn_lds[];      // [C][2]
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  for (int c = threadIdx.x; c < 2 * C; c += 256) gn_lds[c] = 0;
  __syncthreads();
  const int nvec = C / V;
  const int nvp = min(nvec, 256);                 // vectors covered per pass of the workgroup
  const int rstep = 256 / nvp;                    // rows covered per pass
  const int v_in = threadIdx.x % nvp, r_in = threadIdx.x / nvp;
  const bf16_t* xb = x + (int64_t)b * rows * C;
  if (r_in < rstep) {
    for (int v = v_in; v < nvec; v += nvp) {
      float s[V], q[V];
#pragma unroll
      for (int e = 0; e < V; ++e) s[e] = q[e] = 0.f;
      for (int r = r0 + r_in; r < r1; r += rstep) {
        float f[V];
        if constexpr (V == 8) {
          unpack8(*(const uint4*)(xb + (int64_t)r * C + v * 8), f);
        } else {
          f[0] = bf2f(xb[(int64_t)r * C + v]);
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
          s[e] += f[e];
          q[e] = fmaf(f[e], f[e], q[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < V; ++e) {
        sq_add(&gn_lds[2 * (v * V + e)], sq_from_float(s[e]));
        sq_add(&gn_lds[2 * (v * V + e) + 1], sq_from_float(q[e]));
      }
    }
  }
  __syncthreads();
  const int S = C / G;
  for (int g = threadIdx.x; g < G; g += 256) {
    tfimm_sq_t s = 0, q = 0;
    for (int c = g * S; c < (g + 1) * S; ++c) {
      s += gn_lds[2 * c];
      q += gn_lds[2 * c + 1];
    }
    sq_add(&stats[((int64_t)b * G + g) * 2], s);
    sq_add(&stats[((int64_t)b * G + g) * 2 + 1], q);
  }
}

// Pass 2: y = act_after(act((x - mean) * rsqrt(var + eps) * gamma + beta) + residual)
template <int V>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, const tfimm_sq_t* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const bf16_t* __restrict__ residual, bf16_t* __restrict__ y,
                                                       int rows, int C, int G, float eps, int act, int act_after,
                                                       int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float gn_ms[];       // [G][2]: mean, rstd
  const int b = blockIdx.y;
  const int S = C / G;
  const float inv_n = 1.f / ((float)rows * (float)S);
  for (int g = threadIdx.x; g < G; g += 256) {
    const double s = (double)stats[((int64_t)b * G + g) * 2] * (1.0 / 1048576.0);
    const double q = (double)stats[((int64_t)b * G + g) * 2 + 1] * (1.0 / 1048576.0);
    const float mean = (float)(s * inv_n);
    const float var = fmaxf((float)(q * inv_n - (s * inv_n) * (s * inv_n)), 0.f);
    gn_ms[2 * g] = mean;
    gn_ms[2 * g + 1] = rsqrtf(var + eps);
  }
  __syncthreads();
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  const int nvec = C / V;
  const int64_t base = (int64_t)b * rows * C;
  const int64_t n_items = (int64_t)(r1 - r0) * nvec;
  for (int64_t it = threadIdx.x; it < n_items; it += 256) {
    const int r = r0 + (int)(it / nvec), v = (int)(it % nvec);
    const int64_t off = base + (int64_t)r * C + v * V;
    float f[V], rs[V];
    if constexpr (V == 8) {
      unpack8(*(const uint4*)(x + off), f);
      if (residual) unpack8(*(const uint4*)(residual + off), rs);
    } else {
      f[0] = bf2f(x[off]);
      if (residual) rs[0] = bf2f(residual[off]);
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int c = v * V + e;
      const int g = c / S;
      const float inv = gn_ms[2 * g + 1] * gamma[c];                   // tf.nn.batch_normalization (layers/norm.py:104)
      float o = f[e] * inv + (beta[c] - gn_ms[2 * g] * inv);
      o = apply_act(o, act);
      if (residual) o = apply_act(o + rs[e], act_after);
      f[e] = o;
    }
    if constexpr (V == 8) {
      *(uint4*)(y + off) = pack8(f);
    } else {
      y[off] = (bf16_t)f2bf(f[0]);
    }
  }
}

extern "C" int tfimm_hip_group_norm(const void* x, const float* gamma, const float* beta, const void* residual, void* y,
                                    void* stats_ws_v, int B, int rows, int C, int groups, float eps, int act,
                                    int act_after_res, void* stream) {
  tfimm_sq_t* stats_ws = reinterpret_cast<tfimm_sq_t*>(stats_ws_v);
  if (!x || !gamma || !beta || !y || !stats_ws) TFIMM_FAIL(TFIMM_EINVAL, "group_norm: null pointer");
  if (B <= 0 || rows <= 0 || C <= 0 || groups <= 0 || C % groups) TFIMM_FAIL(TFIMM_EINVAL, "group_norm: bad shape");
  if (B > 65535) TFIMM_FAIL(TFIMM_EUNSUP, "group_norm: batch %d > 65535", B);
  if ((size_t)C * 16 > 64 * 1024) TFIMM_FAIL(TFIMM_EUNSUP, "group_norm: %d channels", C);
  hipStream_t st = (hipStream_t)stream;
  // (the library's fill kernel, not a runtime memset: see tfimm_hip_memset_async)
  if (int rc = tfimm_hip_memset_async(stats_ws, 0, (size_t)B * groups * 2 * sizeof(tfimm_sq_t), stream)) return rc;
  // Rows per workgroup: a function of the image size ONLY.  A thread sums its rows of a run in fp32 before it converts to fixed
  // point, so a run length chosen from the batch size (as it was: rows * B / 2048) made the statistics -- and the logits, by
  // up to 8e-3 at batch 32 against batch 2 -- depend on the batch.  8..64 rows: >= 2048 workgroups from batch 42 on at 56 x 56.
  int per = (int)cdiv64(rows, 16);
  if (per < 8) per = 8;
  if (per > 64) per = 64;
  if (per > rows) per = rows;
  const int chunks = (rows + per - 1) / per;
  const bool vec = (C & 7) == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 15) == 0;
  const dim3 grid((unsigned)chunks, (unsigned)B);
  if (vec) {
    TFIMM_LAUNCH(gn_stats_kernel<8>, grid, dim3(256), (size_t)C * 16, st, (const bf16_t*)x, stats_ws, rows, C, groups, per);
    TFIMM_LAUNCH(gn_apply_kernel<8>, grid, dim3(256), (size_t)groups * 8, st, (const bf16_t*)x, stats_ws, gamma, beta,
                 (const bf16_t*)residual, (bf16_t*)y, rows, C, groups, eps, act, act_after_res, per);
  } else {
    TFIMM_LAUNCH(gn_stats_kernel<1>, grid, dim3(256), (size_t)C * 16, st, (const bf16_t*)x, stats_ws, rows, C, groups, per);
    TFIMM_LAUNCH(gn_apply_kernel<1>, grid, dim3(256), (size_t)groups * 8, st, (const bf16_t*)x, stats_ws, gamma, beta,
                 (const bf16_t*)residual, (bf16_t*)y, rows, C, groups, eps, act, act_after_res, per);
  }
  return 0;
}

// =====================================================================================================================
// blur pooling / average pooling
// =====================================================================================================================
__device__ __forceinline__ int reflect_idx(int i, int n) {      // tf.pad(mode="REFLECT"): -1 -> 1, n -> n - 2
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

template <int V>
__global__ __launch_bounds__(256) void blur_pool_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int H, int W,
                                                        int C, int stride, int pad, int OH, int OW, int64_t total) {
  const int nvec = C / V;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
    const int v = (int)(it % nvec);
    int64_t p = it / nvec;
    const int ox = (int)(p % OW);
    p /= OW;
    const int oy = (int)(p % OH);
    const int64_t b = p / OH;
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = reflect_idx(oy * stride + dy - pad, H);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = reflect_idx(ox * stride + dx - pad, W);
        const float w = (float)((dy == 1 ? 2 : 1) * (dx == 1 ? 2 : 1)) * 0.0625f;      // [1 2 1] x [1 2 1] / 16
        const int64_t off = ((b * H + iy) * W + ix) * (int64_t)C + v * V;
        float f[V];
        if constexpr (V == 8) {
          unpack8(*(const uint4*)(x + off), f);
        } else {
          f[0] = bf2f(x[off]);
        }
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = fmaf(w, f[e], acc[e]);
      }
    }
    const int64_t off = ((b * OH + oy) * OW + ox) * (int64_t)C + v * V;
    if constexpr (V == 8) {
      *(uint4*)(y + off) = pack8(acc);
    } else {
      y[off] = (bf16_t)f2bf(acc[0]);
    }
  }
}

extern "C" int tfimm_hip_blur_pool(const void* x, void* y, int B, int H, int W, int C, int stride, void* stream) {
  if (!x || !y) TFIMM_FAIL(TFIMM_EINVAL, "blur_pool: null pointer");
  if (B <= 0 || H < 2 || W < 2 || C <= 0 || stride < 1) TFIMM_FAIL(TFIMM_EINVAL, "blur_pool: bad shape");
  const int pad = (3 + stride) / 2 - 1;                            // layers/blurpool.py:21
  const int OH = (H + 2 * pad - 3) / stride + 1, OW = (W + 2 * pad - 3) / stride + 1;
  const bool vec = (C & 7) == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
  const int64_t total = (int64_t)B * OH * OW * (vec ? C / 8 : C);
  int64_t blocks = cdiv64(total, 256);
  if (blocks > 16384) blocks = 16384;
  if (vec) {
    TFIMM_LAUNCH(blur_pool_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                 (bf16_t*)y, H, W, C, stride, pad, OH, OW, total);
  } else {
    TFIMM_LAUNCH(blur_pool_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                 (bf16_t*)y, H, W, C, stride, pad, OH, OW, total);
  }
  return 0;
}

template <int V>
__global__ __launch_bounds__(256) void avg_pool_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int H, int W,
                                                       int C, int k, int stride, int pad_t, int pad_l, int OH, int OW,
                                                       int64_t total) {
  const int nvec = C / V;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
    const int v = (int)(it % nvec);
    int64_t p = it / nvec;
    const int ox = (int)(p % OW);
    p /= OW;
    const int oy = (int)(p % OH);
    const int64_t b = p / OH;
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    int cnt = 0;
    for (int dy = 0; dy < k; ++dy) {
      const int iy = oy * stride + dy - pad_t;
      if (iy < 0 || iy >= H) continue;
      for (int dx = 0; dx < k; ++dx) {
        const int ix = ox * stride + dx - pad_l;
        if (ix < 0 || ix >= W) continue;
        const int64_t off = ((b * H + iy) * W + ix) * (int64_t)C + v * V;
        float f[V];
        if constexpr (V == 8) {
          unpack8(*(const uint4*)(x + off), f);
        } else {
          f[0] = bf2f(x[off]);
        }
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += f[e];
        ++cnt;
      }
    }
    const float inv = 1.f / (float)cnt;       // the divisor counts VALID elements only (Keras "same" average pooling)
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] *= inv;
    const int64_t off = ((b * OH + oy) * OW + ox) * (int64_t)C + v * V;
    if constexpr (V == 8) {
      *(uint4*)(y + off) = pack8(acc);
    } else {
      y[off] = (bf16_t)f2bf(acc[0]);
    }
  }
}

extern "C" int tfimm_hip_avg_pool(const void* x, void* y, int B, int H, int W, int C, int k, int stride, void* stream) {
  if (!x || !y) TFIMM_FAIL(TFIMM_EINVAL, "avg_pool: null pointer");
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || k < 1 || stride < 1) TFIMM_FAIL(TFIMM_EINVAL, "avg_pool: bad shape");
  const int OH = (H + stride - 1) / stride, OW = (W + stride - 1) / stride;             // "same": ceil(in / stride)
  const int tot_h = (OH - 1) * stride + k - H > 0 ? (OH - 1) * stride + k - H : 0;
  const int tot_w = (OW - 1) * stride + k - W > 0 ? (OW - 1) * stride + k - W : 0;
  const bool vec = (C & 7) == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
  const int64_t total = (int64_t)B * OH * OW * (vec ? C / 8 : C);
  int64_t blocks = cdiv64(total, 256);
  if (blocks > 16384) blocks = 16384;
  if (vec) {
    TFIMM_LAUNCH(avg_pool_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                 (bf16_t*)y, H, W, C, k, stride, tot_h / 2, tot_w / 2, OH, OW, total);
  } else {
    TFIMM_LAUNCH(avg_pool_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                 (bf16_t*)y, H, W, C, k, stride, tot_h / 2, tot_w / 2, OH, OW, total);
  }
  return 0;
}

// =====================================================================================================================
// ECA gate
// =====================================================================================================================
// gate[b][c] = act(sum_t w[t] * mean[b][c + t - pad]), zero padding over the channel axis (ZeroPadding1D + Conv1D VALID,
// layers/attention.py:110-126); mean = sums * inv_count.  One workgroup per image, the means staged in LDS.
__global__ __launch_bounds__(256) void eca_gate_kernel(const float* __restrict__ sums, float inv_count,
                                                       const float* __restrict__ w, float* __restrict__ gate, int C, int k,
                                                       int act) {
  extern __shared__ __attribute__((aligned(16))) float eca_m[];
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) eca_m[c] = sums[(int64_t)b * C + c] * inv_count;
  __syncthreads();
  const int pad = (k - 1) / 2;
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = 0.f;
    for (int t = 0; t < k; ++t) {
      const int i = c + t - pad;
      if (i >= 0 && i < C) acc = fmaf(w[t], eca_m[i], acc);
    }
    gate[(int64_t)b * C + c] = apply_act(acc, act);
  }
}

extern "C" int tfimm_hip_eca_gate(const float* sums, float inv_count, const float* w, float* gate, int B, int C, int k,
                                  int gate_act, void* stream) {
  if (!sums || !w || !gate) TFIMM_FAIL(TFIMM_EINVAL, "eca_gate: null pointer");
  if (B <= 0 || C <= 0 || k <= 0 || !(k & 1)) TFIMM_FAIL(TFIMM_EINVAL, "eca_gate: bad shape");
  if ((size_t)C * 4 > 64 * 1024) TFIMM_FAIL(TFIMM_EUNSUP, "eca_gate: %d channels", C);
  TFIMM_LAUNCH(eca_gate_kernel, dim3((unsigned)B), dim3(256), (size_t)C * 4, (hipStream_t)stream, sums, inv_count, w, gate,
               C, k, gate_act);
  return 0;
}

// =====================================================================================================================
// grouped 3x3 convolution (ResNeXt, resnet.py:229-236: Conv2D(groups = cardinality) between ZeroPadding2D(1) and bn2)
// =====================================================================================================================
// Groups of w = C / groups <= 32 channels (input and output width of a group are equal).  A dense kernel over the
// block-diagonal expansion multiplies C x C per tap -- `groups` times the useful work; here a wave owns ONE super-group
// of 32 output channels, whose inputs are exactly the same 32 channels, and multiplies 32 x 32 per tap (32 / w times the
// useful work, 1x at w = 32):
//   * the super-group's 9 x [32 x 32] weights (block-diagonal inside the 32 x 32) stay in REGISTERS for the whole kernel
//     as 18 MFMA A fragments (host-packed per lane: pack.pack_grouped3x3);
//   * a wave walks 32-pixel tiles; the B operand of v_mfma_f32_32x32x16_bf16 (one pixel per lane, 8 consecutive input
//     channels) is a 16-byte global load per lane and k-step straight from the NHWC tensor -- taps outside the image are
//     zeroed in the register -- no LDS at all;
//   * the 4 waves of a workgroup take 4 neighbouring super-groups of the SAME pixels, so together they read whole
//     256-byte runs of each pixel;
//   * D[n][m] = W . X^T leaves a lane with 4 consecutive channels of its pixel per accumulator quad: bias (folded BN),
//     activation, 8-byte stores.
__global__ __launch_bounds__(256, 2) void grouped_conv3x3_kernel(const bf16_t* __restrict__ x, const uint4* __restrict__ wfrag,
                                                                 const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                                 int B, int H, int W, int C, int stride, int OH, int OW,
                                                                 int act, int n_tiles, int tiles_per_block) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int frow = lane & 31, fhi = lane >> 5;
  const int sg = blockIdx.y * 4 + wave;                 // super-group: channels [32 sg, 32 sg + 32)
  if (sg * 32 >= C) return;
  bf16x8 wf[18];
#pragma unroll
  for (int ks = 0; ks < 18; ++ks) wf[ks] = __builtin_bit_cast(bf16x8, wfrag[((size_t)sg * 18 + ks) * 64 + lane]);
  f32x4 bq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 b4 = *reinterpret_cast<const float4*>(bias + sg * 32 + q * 8 + fhi * 4);
    bq[q] = f32x4{b4.x, b4.y, b4.z, b4.w};
  }
  const ActParams actp = make_act(act);
  const int64_t M = (int64_t)B * OH * OW;
  const int t0 = blockIdx.x * tiles_per_block;
  const int t1 = min(n_tiles, t0 + tiles_per_block);
  for (int t = t0; t < t1; ++t) {
    const int64_t m = (int64_t)t * 32 + frow;
    const bool mok = m < M;
    const int64_t mm = mok ? m : 0;
    const int b = (int)(mm / ((int64_t)OH * OW));
    const int rem = (int)(mm - (int64_t)b * OH * OW);
    const int oy = rem / OW, ox = rem - oy * OW;
    const int iy0 = oy * stride - 1, ix0 = ox * stride - 1;
    const bf16_t* xb = x + (size_t)b * H * W * C + sg * 32 + fhi * 8;
    uint4 xf[18];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int iy = iy0 + tap / 3, ix = ix0 + tap % 3;
      const bool ok = mok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const bf16_t* px = xb + ((size_t)(ok ? iy : 0) * W + (ok ? ix : 0)) * C;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const uint4 v = *reinterpret_cast<const uint4*>(px + half * 16);
        xf[tap * 2 + half] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
      }
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 18; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], __builtin_bit_cast(bf16x8, xf[ks]), acc, 0, 0, 0);
    if (mok) {
      bf16_t* py = y + (size_t)m * C + sg * 32 + fhi * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v0 = act1(acc[q * 4 + 0] + bq[q][0], actp), v1 = act1(acc[q * 4 + 1] + bq[q][1], actp);
        const float v2 = act1(acc[q * 4 + 2] + bq[q][2], actp), v3 = act1(acc[q * 4 + 3] + bq[q][3], actp);
        *reinterpret_cast<uint2*>(py + q * 8) = make_uint2(pack_bf2(v0, v1), pack_bf2(v2, v3));
      }
    }
  }
}

extern "C" int tfimm_hip_grouped_conv3x3(const void* x, const void* wfrag, const float* bias, void* y, int B, int H, int W,
                                         int C, int stride, int act, void* stream) {
  if (!x || !wfrag || !bias || !y) TFIMM_FAIL(TFIMM_EINVAL, "grouped_conv3x3: null pointer");
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 31) || (stride != 1 && stride != 2))
    TFIMM_FAIL(TFIMM_EINVAL, "grouped_conv3x3: bad shape (C must be a multiple of 32, stride 1 or 2)");
  if (((uintptr_t)x | (uintptr_t)wfrag | (uintptr_t)bias | (uintptr_t)y) & 15)
    TFIMM_FAIL(TFIMM_EINVAL, "grouped_conv3x3: pointers must be 16-byte aligned");
  const int OH = (H + 2 - 3) / stride + 1, OW = (W + 2 - 3) / stride + 1;
  const int64_t M = (int64_t)B * OH * OW;
  const int64_t n_tiles = cdiv64(M, 32);
  if (n_tiles > 0x7fffffffLL) TFIMM_FAIL(TFIMM_EINVAL, "grouped_conv3x3: too many pixels");
  const int gy = (C / 32 + 3) / 4;
  // enough workgroups to fill the chip a few times over, each keeping its weights for many tiles
  int64_t gx = cdiv64(2048, gy);
  if (gx > n_tiles) gx = n_tiles;
  const int per = (int)cdiv64(n_tiles, gx);
  gx = cdiv64(n_tiles, per);
  TFIMM_LAUNCH(grouped_conv3x3_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
               (const uint4*)wfrag, bias, (bf16_t*)y, B, H, W, C, stride, OH, OW, act, (int)n_tiles, per);
  return 0;
}
