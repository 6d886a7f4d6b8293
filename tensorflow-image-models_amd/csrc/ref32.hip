// tfimm_hip_ref_*: the FLOAT32 VERIFICATION PATH of the engine (TFIMM_PRECISION=fp32, tfimm/engine/precision.py).
//
// The product path stores activations and GEMM weights in bf16 and is held to bf16-sized parity bars.  The reference is
// float32 end to end and its own value-pinning test uses 1e-3 relative to the maximum (tests/test_timm.py:71).  To show
// that what the engine computes is the reference's arithmetic -- same folded BatchNorms, same epsilons, exact-erf GELU,
// same padding / pooling / window index maps -- and not merely something inside a bf16 error band, the SAME layer
// program (tfimm/engine/graph.py: same lowering code, same host-side weight transformations, same op list minus the
// cross-layer fusions) can be bound to the kernels below: float32 activations, float32 weights, float32 accumulation.
// They are deliberately plain (one thread or one wave per output, LDS-tiled GEMM at best): this path exists to be
// obviously right and is 20-50x slower than the bf16 kernels.  Signatures mirror the bf16 entry points (include/tfimm_hip.h)
// with every `bf16` tensor replaced by `float`, so a plan swaps the function and keeps the arguments.
#include "common.h"

#include <cstring>

namespace {

__device__ __forceinline__ float ref_act(float v, int act) {
  switch (act) {
    case TFIMM_ACT_RELU: return fmaxf(v, 0.f);
    case TFIMM_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));   // keras gelu(approximate=False)
    case TFIMM_ACT_SWISH: return v / (1.f + expf(-v));
    case TFIMM_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case TFIMM_ACT_RELU6: return fminf(fmaxf(v, 0.f), 6.f);
    case TFIMM_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

inline unsigned grid_for(int64_t total, int block = 256) {
  int64_t g = (total + block - 1) / block;
  if (g > 65535LL * 16) g = 65535LL * 16;
  return (unsigned)(g < 1 ? 1 : g);
}

// ------------------------------------------------------------------------------------------------- GEMM / conv
struct RefGemm {
  const float* a;
  const float* wt;
  const float* bias;
  const float* residual;
  float* out;
  const float* a_scale;
  int M, N, K, lda, ldw, ldr, ldc, act, act_after_res, res_mod, remap_in, remap_out, remap_off, mode;
  int B, H, W, Cin, KH, KW, stride, stride_w, pad_t, pad_l, OH, OW, rows_per_image, cpitch;
};

// 16 x 16 outputs per workgroup, K walked in 16-wide slices through LDS; sequential fp32 FMA chain per output
__global__ void __launch_bounds__(256) ref_gemm_kernel(const RefGemm p) {
  __shared__ float As[16][17], Bs[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int64_t m0 = (int64_t)blockIdx.y * 16;
  const int n0 = blockIdx.x * 16;
  // row of A this thread stages (ty) -- conv: decode the output pixel once
  const int64_t ma = m0 + ty;
  int ab = 0, aoy = 0, aox = 0;
  if (p.mode != TFIMM_A_DENSE && ma < p.M) {
    const int ohw = p.OH * p.OW;
    ab = (int)(ma / ohw);
    const int rem = (int)(ma - (int64_t)ab * ohw);
    aoy = rem / p.OW;
    aox = rem - aoy * p.OW;
  }
  float acc = 0.f;
  for (int k0 = 0; k0 < p.K; k0 += 16) {
    const int k = k0 + tx;
    float av = 0.f;
    if (ma < p.M && k < p.K) {
      if (p.mode == TFIMM_A_DENSE) {
        av = p.a[ma * p.lda + k];
        if (p.a_scale) av *= p.a_scale[(ma / p.rows_per_image) * p.K + k];
      } else {
        const int tap = k / p.Cin, ci = k - tap * p.Cin;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        const int iy = aoy * p.stride - p.pad_t + ky, ix = aox * p.stride_w - p.pad_l + kx;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) av = p.a[(((int64_t)ab * p.H + iy) * p.W + ix) * p.cpitch + ci];
      }
    }
    As[ty][tx] = av;
    const int nb = n0 + ty;
    Bs[ty][tx] = (nb < p.N && k < p.K) ? p.wt[(int64_t)nb * p.ldw + k] : 0.f;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) acc = fmaf(As[ty][kk], Bs[tx][kk], acc);
    __syncthreads();
  }
  const int64_t m = m0 + ty;
  const int n = n0 + tx;
  if (m >= p.M || n >= p.N) return;
  float v = acc + (p.bias ? p.bias[n] : 0.f);
  float r = 0.f;
  if (p.residual) {
    const int64_t rm = p.res_mod > 0 ? m % p.res_mod : m;
    r = p.residual[rm * p.ldr + n];
  }
  if (p.act_after_res) v += r;
  v = ref_act(v, p.act);
  if (!p.act_after_res) v += r;
  const int64_t om = p.remap_in > 0 ? (m / p.remap_in) * p.remap_out + m % p.remap_in + p.remap_off : m;
  p.out[om * p.ldc + n] = v;
}

// ------------------------------------------------------------------------------------------------- input
struct RefNorm { float mean[8], std[8]; };
__global__ void ref_cast_input_kernel(const void* in, int in_dtype, float* out, int64_t n_pixels, int c_in, int c_out, RefNorm nm) {
  const int64_t total = n_pixels * c_out;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(id % c_out);
    const int64_t px = id / c_out;
    float v = 0.f;
    if (c < c_in) {
      const int64_t s = px * c_in + c;
      if (in_dtype == 0) v = reinterpret_cast<const float*>(in)[s];
      else if (in_dtype == 1) v = bf2f(reinterpret_cast<const bf16_t*>(in)[s]);
      else v = ((float)reinterpret_cast<const uint8_t*>(in)[s] / 255.f - nm.mean[c]) / nm.std[c];   // factory.py:165-167
    }
    out[id] = v;
  }
}

// ------------------------------------------------------------------------------------------------- row ops
// one wave per row: two-pass population moments, y = x * inv + (beta - mean * inv), inv = rsqrt(var + eps) * gamma
__global__ void __launch_bounds__(256) ref_layernorm_kernel(const float* x, float* y, const float* gamma, const float* beta,
                                                            int64_t rows, int d, int64_t xs, int64_t ys, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + r * xs;
  float s = 0.f;
  for (int j = lane; j < d; j += 64) s += xr[j];
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
  for (int j = lane; j < d; j += 64) { const float t = xr[j] - mean; q += t * t; }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)d + eps);
  float* yr = y + r * ys;
  for (int j = lane; j < d; j += 64) {
    const float inv = rstd * gamma[j];
    yr[j] = xr[j] * inv + (beta[j] - mean * inv);
  }
}

__global__ void __launch_bounds__(256) ref_patch_merge_ln_kernel(const float* x, float* y, const float* gamma, const float* beta,
                                                                 int B, int H, int W, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int H2 = H / 2, W2 = W / 2, D = 4 * C;
  const int64_t rows = (int64_t)B * H2 * W2;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int x2 = (int)(r % W2), y2 = (int)((r / W2) % H2);
  const int64_t b = r / ((int64_t)W2 * H2);
  auto src = [&](int j) -> float {       // concat order (dy, dx) = (0,0), (1,0), (0,1), (1,1)   (swin.py:353-357)
    const int part = j / C, c = j - part * C;
    const int dy = part & 1, dx = part >> 1;
    return x[((b * H + (2 * y2 + dy)) * W + (2 * x2 + dx)) * C + c];
  };
  float s = 0.f;
  for (int j = lane; j < D; j += 64) s += src(j);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int j = lane; j < D; j += 64) { const float t = src(j) - mean; q += t * t; }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + eps);
  for (int j = lane; j < D; j += 64) {
    const float inv = rstd * gamma[j];
    y[r * D + j] = src(j) * inv + (beta[j] - mean * inv);
  }
}

__global__ void ref_copy_rows_kernel(const float* src, float* dst, int64_t total, int src_rows, int dst_rows, int dst_row0, int d) {
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(id % d);
    const int64_t row = id / d;
    const int64_t b = row / src_rows, r = row - b * src_rows;
    dst[(b * dst_rows + dst_row0 + r) * d + c] = src[id];
  }
}

__global__ void ref_bcast_rows_kernel(const float* src, float* dst, int B, int n_rows, int d, int dst_rpi) {
  const int64_t total = (int64_t)B * n_rows * d;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(id % d);
    const int64_t t = (id / d) % n_rows, b = id / ((int64_t)d * n_rows);
    dst[(b * dst_rpi + t) * d + c] = src[t * d + c];
  }
}

__global__ void ref_mean_rows_kernel(const float* x, float* y, int B, int R, int C) {
  const int64_t total = (int64_t)B * C;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(id % C);
    const int64_t b = id / C;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += x[(b * R + r) * C + c];
    y[id] = s / (float)R;
  }
}

__global__ void ref_scale_channels_kernel(const float* x, const float* gate, const float* residual, float* y, int B, int R, int C,
                                          int act_after) {
  const int64_t total = (int64_t)B * R * C;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(id % C);
    const int64_t b = id / ((int64_t)R * C);
    float v = x[id] * gate[b * C + c];
    if (residual) v += residual[id];
    if (act_after) v = fmaxf(v, 0.f);
    y[id] = v;
  }
}

// ------------------------------------------------------------------------------------------------- pooling / depthwise
__global__ void ref_maxpool_kernel(const float* x, float* y, int B, int H, int W, int C, int k, int stride, int pad, int OH, int OW) {
  const int64_t total = (int64_t)B * OH * OW * C;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(id % C);
    const int ox = (int)((id / C) % OW), oy = (int)((id / ((int64_t)C * OW)) % OH);
    const int64_t b = id / ((int64_t)C * OW * OH);
    float m = -3.4e38f;
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx) {
        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        // the border is ZeroPadding2D (resnet.py:538-540): it contributes zeros, not -inf
        const float v = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[((b * H + iy) * W + ix) * C + c] : 0.f;
        m = fmaxf(m, v);
      }
    y[id] = m;
  }
}

__global__ void ref_avg_pool_kernel(const float* x, float* y, int B, int H, int W, int C, int k, int stride, int OH, int OW,
                                    int pt, int pl) {
  const int64_t total = (int64_t)B * OH * OW * C;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(id % C);
    const int ox = (int)((id / C) % OW), oy = (int)((id / ((int64_t)C * OW)) % OH);
    const int64_t b = id / ((int64_t)C * OW * OH);
    float s = 0.f;
    int n = 0;
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx) {
        const int iy = oy * stride - pt + ky, ix = ox * stride - pl + kx;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) { s += x[((b * H + iy) * W + ix) * C + c]; ++n; }
      }
    y[id] = s / (float)n;       // windows clipped at the border average their valid elements (AveragePooling2D "same")
  }
}

__global__ void ref_blur_pool_kernel(const float* x, float* y, int B, int H, int W, int C, int stride, int OH, int OW, int p) {
  const int64_t total = (int64_t)B * OH * OW * C;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(id % C);
    const int ox = (int)((id / C) % OW), oy = (int)((id / ((int64_t)C * OW)) % OH);
    const int64_t b = id / ((int64_t)C * OW * OH);
    float s = 0.f;
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) {
        int iy = oy * stride - p + ky, ix = ox * stride - p + kx;
        iy = iy < 0 ? -iy : (iy >= H ? 2 * (H - 1) - iy : iy);      // tf.pad(REFLECT)
        ix = ix < 0 ? -ix : (ix >= W ? 2 * (W - 1) - ix : ix);
        const float wgt = (float)((ky == 1 ? 2 : 1) * (kx == 1 ? 2 : 1)) / 16.f;
        s += wgt * x[((b * H + iy) * W + ix) * C + c];
      }
    y[id] = s;
  }
}

__global__ void ref_dwconv_kernel(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C, int k,
                                  int stride, int pad_t, int pad_l, int OH, int OW, int act) {
  const int64_t total = (int64_t)B * OH * OW * C;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(id % C);
    const int ox = (int)((id / C) % OW), oy = (int)((id / ((int64_t)C * OW)) % OH);
    const int64_t b = id / ((int64_t)C * OW * OH);
    float s = 0.f;
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx) {
        const int iy = oy * stride - pad_t + ky, ix = ox * stride - pad_l + kx;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) s = fmaf(x[((b * H + iy) * W + ix) * C + c], w[(ky * k + kx) * C + c], s);
      }
    y[id] = ref_act(s + (bias ? bias[c] : 0.f), act);
  }
}

// GroupNormalization: one workgroup per (image, group); two-pass population moments over (rows, C / groups)
__global__ void __launch_bounds__(256) ref_group_norm_kernel(const float* x, const float* gamma, const float* beta,
                                                             const float* residual, float* y, int rows, int C, int groups, float eps,
                                                             int act, int act_after) {
  __shared__ float red[4];
  const int g = blockIdx.x % groups;
  const int64_t b = blockIdx.x / groups;
  const int cg = C / groups;
  const int64_t n = (int64_t)rows * cg;
  const float* xb = x + b * rows * C + g * cg;
  auto block_sum = [&](float v) -> float {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += xb[(i / cg) * C + (i % cg)];
  const float mean = block_sum(s) / (float)n;
  float q = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) { const float t = xb[(i / cg) * C + (i % cg)] - mean; q += t * t; }
  const float rstd = 1.f / sqrtf(block_sum(q) / (float)n + eps);
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const int c = g * cg + (int)(i % cg);
    const int64_t off = b * rows * C + (i / cg) * C + c;
    const float inv = rstd * gamma[c];
    float v = ref_act(x[off] * inv + (beta[c] - mean * inv), act);
    if (residual) v = ref_act(v + residual[off], act_after);
    y[off] = v;
  }
}

// ------------------------------------------------------------------------------------------------- attention
struct RefAttn {
  const float* qkv;
  float* out;
  const float* rel_bias;
  int batch, n_tokens, heads, hd;
  float scale;
  int window, shift, res_h, res_w, n, nwx, nw, ld, dmodel;
};

__device__ __forceinline__ int64_t ref_token_row(const RefAttn& p, int seq, int t, int* region) {
  if (p.window == 0) {
    *region = 0;
    return (int64_t)seq * p.n_tokens + t;
  }
  const int b = seq / p.nw, w = seq - b * p.nw;
  const int wy = w / p.nwx, wx = w - wy * p.nwx;
  const int ty = t / p.window, tx = t - ty * p.window;
  const int ys = wy * p.window + ty, xs = wx * p.window + tx;      // coordinates in the rolled (tf.roll(-shift)) frame
  int y = ys + p.shift, x = xs + p.shift;
  if (y >= p.res_h) y -= p.res_h;
  if (x >= p.res_w) x -= p.res_w;
  // region ids of swin.py:249-263 (slices (0, -ws), (-ws, -shift), (-shift, None))
  const int rh = ys < p.res_h - p.window ? 0 : (ys < p.res_h - p.shift ? 1 : 2);
  const int rw = xs < p.res_w - p.window ? 0 : (xs < p.res_w - p.shift ? 1 : 2);
  *region = rh * 3 + rw;
  return (int64_t)b * p.n_tokens + (int64_t)y * p.res_w + x;
}

// one wave per (sequence, head, query): scores in LDS, softmax by wave reductions, lane = channel for P . V
__global__ void __launch_bounds__(64) ref_attention_kernel(const RefAttn p) {
  extern __shared__ float sc[];      // [n] scores / probabilities
  const int lane = threadIdx.x;
  int bid = blockIdx.x;
  const int q = bid % p.n; bid /= p.n;
  const int h = bid % p.heads;
  const int seq = bid / p.heads;
  int qreg = 0;
  const int64_t qrow = ref_token_row(p, seq, q, &qreg);
  const float* qp = p.qkv + qrow * p.ld + h * p.hd;
  float mx = -3.4e38f;
  for (int j = lane; j < p.n; j += 64) {
    int kreg = 0;
    const int64_t krow = ref_token_row(p, seq, j, &kreg);
    const float* kp = p.qkv + krow * p.ld + p.dmodel + h * p.hd;
    float s = 0.f;
    for (int d = 0; d < p.hd; ++d) s = fmaf(qp[d], kp[d], s);
    s *= p.scale;
    if (p.rel_bias) s += p.rel_bias[((int64_t)h * p.n + q) * p.n + j];
    if (p.window && p.shift > 0 && kreg != qreg) s += -100.0f;      // swin.py:249-273
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < p.n; j += 64) {
    const float e = expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __syncthreads();
  for (int d = lane; d < p.hd; d += 64) {
    float acc = 0.f;
    for (int j = 0; j < p.n; ++j) {
      int rg;
      const int64_t vrow = ref_token_row(p, seq, j, &rg);
      acc = fmaf(sc[j] / sum, p.qkv[vrow * p.ld + 2 * p.dmodel + h * p.hd + d], acc);
    }
    p.out[qrow * p.dmodel + h * p.hd + d] = acc;
  }
}

__global__ void __launch_bounds__(64) ref_attention_probs_kernel(const float* qkv, float* probs, int n, int heads, int hd, float scale) {
  extern __shared__ float sc[];
  const int lane = threadIdx.x;
  int bid = blockIdx.x;
  const int q = bid % n; bid /= n;
  const int h = bid % heads;
  const int64_t b = bid / heads;
  const int ld = 3 * heads * hd, dm = heads * hd;
  const float* qp = qkv + (b * n + q) * ld + h * hd;
  float mx = -3.4e38f;
  for (int j = lane; j < n; j += 64) {
    const float* kp = qkv + (b * n + j) * ld + dm + h * hd;
    float s = 0.f;
    for (int d = 0; d < hd; ++d) s = fmaf(qp[d], kp[d], s);
    s *= scale;
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) { const float e = expf(sc[j] - mx); sc[j] = e; sum += e; }
  sum = wave_sum(sum);
  float* pr = probs + ((b * heads + h) * n + q) * n;
  for (int j = lane; j < n; j += 64) pr[j] = sc[j] / sum;
}

struct RefThaW { float wl[256], bl[16], ww[256], bw[16]; };
// CaiT talking-heads attention (cait.py:233-262): one workgroup per query row, all heads
__global__ void __launch_bounds__(256) ref_tha_kernel(const float* qkv, float* out, int n, int H, int hd, float scale, const RefThaW w) {
  extern __shared__ float tg[];
  float* s0 = tg;
  float* s1 = tg + (size_t)H * n;
  const int tid = threadIdx.x;
  const int64_t img = blockIdx.x / n;
  const int i = blockIdx.x - (int)img * n;
  const int ld = 3 * H * hd, dm = H * hd;
  const int64_t row0 = img * n;
  const float* qrow = qkv + (row0 + i) * ld;
  for (int id = tid; id < H * n; id += 256) {
    const int h = id / n, j = id - h * n;
    const float* krow = qkv + (row0 + j) * ld + dm + h * hd;
    float acc = 0.f;
    for (int d = 0; d < hd; ++d) acc = fmaf(scale * qrow[h * hd + d], krow[d], acc);
    s0[id] = acc;
  }
  __syncthreads();
  for (int id = tid; id < H * n; id += 256) {
    const int hp = id / n, j = id - hp * n;
    float acc = w.bl[hp];
    for (int h = 0; h < H; ++h) acc = fmaf(s0[h * n + j], w.wl[h * H + hp], acc);
    s1[id] = acc;
  }
  __syncthreads();
  for (int hp = tid >> 6; hp < H; hp += 4) {
    const int lane = tid & 63;
    float m = -3.4e38f;
    for (int j = lane; j < n; j += 64) m = fmaxf(m, s1[hp * n + j]);
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) { const float e = expf(s1[hp * n + j] - m); s1[hp * n + j] = e; sum += e; }
    sum = wave_sum(sum);
    for (int j = lane; j < n; j += 64) s1[hp * n + j] /= sum;
  }
  __syncthreads();
  for (int id = tid; id < H * n; id += 256) {
    const int hq = id / n, j = id - hq * n;
    float acc = w.bw[hq];
    for (int hp = 0; hp < H; ++hp) acc = fmaf(s1[hp * n + j], w.ww[hp * H + hq], acc);
    s0[id] = acc;
  }
  __syncthreads();
  for (int id = tid; id < dm; id += 256) {
    const int hq = id / hd;
    float acc = 0.f;
    for (int j = 0; j < n; ++j) acc = fmaf(s0[hq * n + j], qkv[(row0 + j) * ld + 2 * dm + id], acc);
    out[(row0 + i) * dm + id] = acc;
  }
}

__global__ void __launch_bounds__(64) ref_class_attn_kernel(const float* q, const float* kv, float* out, int n, int heads, int hd,
                                                            int ldq, int ldkv, int ldo) {
  extern __shared__ float sc[];
  const int lane = threadIdx.x;
  const int h = blockIdx.x % heads;
  const int64_t img = blockIdx.x / heads;
  const int dm = heads * hd;
  const float* qp = q + img * ldq + h * hd;
  const float* kbase = kv + img * n * ldkv + h * hd;
  float mx = -3.4e38f;
  for (int j = lane; j < n; j += 64) {
    float s = 0.f;
    for (int d = 0; d < hd; ++d) s = fmaf(qp[d], kbase[(int64_t)j * ldkv + d], s);
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) { const float e = expf(sc[j] - mx); sc[j] = e; sum += e; }
  sum = wave_sum(sum);
  __syncthreads();
  for (int d = lane; d < hd; d += 64) {
    float acc = 0.f;
    for (int j = 0; j < n; ++j) acc = fmaf(sc[j] / sum, kbase[(int64_t)j * ldkv + dm + d], acc);
    out[img * ldo + h * hd + d] = acc;
  }
}

}  // namespace

#define REF_LAUNCH(kernel, grid, block, lds, stream, ...) TFIMM_LAUNCH(kernel, dim3(grid), dim3(block), lds, (hipStream_t)stream, __VA_ARGS__)

extern "C" {

int tfimm_hip_ref_gemm(const tfimm_gemm_desc* dp, void* stream) {
  if (!dp) TFIMM_FAIL(TFIMM_EINVAL, "ref_gemm: null descriptor");
  const tfimm_gemm_desc& d = *dp;
  if (!d.a || !d.wt || !d.out || d.M <= 0 || d.N <= 0 || d.K <= 0) TFIMM_FAIL(TFIMM_EINVAL, "ref_gemm: bad descriptor");
  if (d.ln_stats || d.ln_c1) TFIMM_FAIL(TFIMM_EUNSUP, "ref_gemm: LayerNorm folding does not exist on the fp32 path");
  if (d.mode != TFIMM_A_DENSE && d.mode != TFIMM_A_CONV) TFIMM_FAIL(TFIMM_EUNSUP, "ref_gemm: mode %d", d.mode);
  RefGemm p;
  p.a = (const float*)d.a; p.wt = (const float*)d.wt; p.bias = d.bias; p.residual = (const float*)d.residual;
  p.out = (float*)d.out; p.a_scale = d.a_scale;
  p.M = d.M; p.N = d.N; p.K = d.K; p.lda = d.lda; p.ldw = d.ldw; p.ldr = d.ldr; p.ldc = d.ldc;
  p.act = d.act; p.act_after_res = d.act_after_res; p.res_mod = d.res_mod;
  p.remap_in = d.remap_in; p.remap_out = d.remap_out; p.remap_off = d.remap_off; p.mode = d.mode;
  p.B = d.B; p.H = d.H; p.W = d.W; p.Cin = d.Cin; p.KH = d.KH; p.KW = d.KW; p.stride = d.stride;
  p.stride_w = d.stride_w > 0 ? d.stride_w : d.stride;
  p.pad_t = d.pad_t; p.pad_l = d.pad_l; p.OH = d.OH; p.OW = d.OW; p.rows_per_image = d.rows_per_image > 0 ? d.rows_per_image : 1;
  p.cpitch = d.pix_pitch > 0 ? d.pix_pitch : d.Cin;
  if (d.mode == TFIMM_A_CONV && d.K != d.KH * d.KW * d.Cin) TFIMM_FAIL(TFIMM_EINVAL, "ref_gemm: K != KH*KW*Cin");
  const int64_t gy = ((int64_t)d.M + 15) / 16;
  if (gy > 0x7fffffffLL) TFIMM_FAIL(TFIMM_EINVAL, "ref_gemm: grid too large");
  (void)hipGetLastError();
  hipLaunchKernelGGL(ref_gemm_kernel, dim3((unsigned)((d.N + 15) / 16), (unsigned)gy), dim3(256), 0, (hipStream_t)stream, p);
  TFIMM_LAUNCH_CHECK();
  return 0;
}

/* in_dtype: 0 float32, 1 bf16, 2 uint8 with (mean, std) applied as create_preprocessing does (models/factory.py:165-167);
   mean / std: HOST arrays of c_in floats (uint8 only, else NULL) */
int tfimm_hip_ref_cast_input(const void* in, int in_dtype, void* out, int64_t n_pixels, int c_in, int c_out, const float* mean,
                             const float* std, void* stream) {
  if (!in || !out || n_pixels <= 0 || c_in <= 0 || c_out < c_in) TFIMM_FAIL(TFIMM_EINVAL, "ref_cast_input: bad arguments");
  RefNorm nm;
  memset(&nm, 0, sizeof(nm));
  if (in_dtype == 2) {
    if (!mean || !std || c_in > 8) TFIMM_FAIL(TFIMM_EINVAL, "ref_cast_input: uint8 input needs mean / std of <= 8 channels");
    for (int c = 0; c < c_in; ++c) { nm.mean[c] = mean[c]; nm.std[c] = std[c]; }
  }
  REF_LAUNCH(ref_cast_input_kernel, grid_for(n_pixels * c_out), 256, 0, stream, in, in_dtype, (float*)out, n_pixels, c_in, c_out, nm);
  return 0;
}

int tfimm_hip_ref_layernorm(const void* x, void* y, const float* gamma, const float* beta, int64_t rows, int d, int64_t x_stride,
                            int64_t y_stride, float eps, void* stream) {
  if (!x || !y || !gamma || !beta || rows <= 0 || d <= 0) TFIMM_FAIL(TFIMM_EINVAL, "ref_layernorm: bad arguments");
  REF_LAUNCH(ref_layernorm_kernel, (unsigned)((rows + 3) / 4), 256, 0, stream, (const float*)x, (float*)y, gamma, beta, rows, d,
             x_stride, y_stride, eps);
  return 0;
}

int tfimm_hip_ref_patch_merge_ln(const void* x, void* y, const float* gamma, const float* beta, int B, int H, int W, int C, float eps,
                                 void* stream) {
  if (!x || !y || B <= 0 || (H & 1) || (W & 1)) TFIMM_FAIL(TFIMM_EINVAL, "ref_patch_merge_ln: bad arguments");
  const int64_t rows = (int64_t)B * (H / 2) * (W / 2);
  REF_LAUNCH(ref_patch_merge_ln_kernel, (unsigned)((rows + 3) / 4), 256, 0, stream, (const float*)x, (float*)y, gamma, beta, B, H, W, C,
             eps);
  return 0;
}

int tfimm_hip_ref_copy_rows(const void* src, void* dst, int B, int src_rows, int dst_rows, int dst_row0, int d, void* stream) {
  const int64_t total = (int64_t)B * src_rows * d;
  if (!src || !dst || total <= 0) TFIMM_FAIL(TFIMM_EINVAL, "ref_copy_rows: bad arguments");
  REF_LAUNCH(ref_copy_rows_kernel, grid_for(total), 256, 0, stream, (const float*)src, (float*)dst, total, src_rows, dst_rows, dst_row0, d);
  return 0;
}

int tfimm_hip_ref_bcast_rows(const void* src, void* dst, int B, int n_rows, int d, int dst_rows_per_image, void* stream) {
  if (!src || !dst || B <= 0) TFIMM_FAIL(TFIMM_EINVAL, "ref_bcast_rows: bad arguments");
  REF_LAUNCH(ref_bcast_rows_kernel, grid_for((int64_t)B * n_rows * d), 256, 0, stream, (const float*)src, (float*)dst, B, n_rows, d,
             dst_rows_per_image);
  return 0;
}

int tfimm_hip_ref_mean_rows(const void* x, void* y, int B, int R, int C, int out_f32, void* stream) {
  (void)out_f32;
  if (!x || !y || B <= 0 || R <= 0 || C <= 0) TFIMM_FAIL(TFIMM_EINVAL, "ref_mean_rows: bad arguments");
  REF_LAUNCH(ref_mean_rows_kernel, grid_for((int64_t)B * C, 64), 64, 0, stream, (const float*)x, (float*)y, B, R, C);
  return 0;
}

int tfimm_hip_ref_scale_channels(const void* x, const float* gate, const void* residual, void* y, int B, int R, int C, int act_after,
                                 void* stream) {
  if (!x || !gate || !y) TFIMM_FAIL(TFIMM_EINVAL, "ref_scale_channels: null pointer");
  REF_LAUNCH(ref_scale_channels_kernel, grid_for((int64_t)B * R * C), 256, 0, stream, (const float*)x, gate, (const float*)residual,
             (float*)y, B, R, C, act_after);
  return 0;
}

int tfimm_hip_ref_maxpool(const void* x, void* y, int B, int H, int W, int C, int k, int stride, int pad, int OH, int OW, void* stream) {
  if (!x || !y) TFIMM_FAIL(TFIMM_EINVAL, "ref_maxpool: null pointer");
  REF_LAUNCH(ref_maxpool_kernel, grid_for((int64_t)B * OH * OW * C), 256, 0, stream, (const float*)x, (float*)y, B, H, W, C, k, stride,
             pad, OH, OW);
  return 0;
}

int tfimm_hip_ref_avg_pool(const void* x, void* y, int B, int H, int W, int C, int k, int stride, void* stream) {
  if (!x || !y) TFIMM_FAIL(TFIMM_EINVAL, "ref_avg_pool: null pointer");
  const int OH = (H + stride - 1) / stride, OW = (W + stride - 1) / stride;
  const int th = (OH - 1) * stride + k - H, tw = (OW - 1) * stride + k - W;      // TF "same": before = total / 2
  REF_LAUNCH(ref_avg_pool_kernel, grid_for((int64_t)B * OH * OW * C), 256, 0, stream, (const float*)x, (float*)y, B, H, W, C, k, stride,
             OH, OW, (th > 0 ? th : 0) / 2, (tw > 0 ? tw : 0) / 2);
  return 0;
}

int tfimm_hip_ref_blur_pool(const void* x, void* y, int B, int H, int W, int C, int stride, void* stream) {
  if (!x || !y) TFIMM_FAIL(TFIMM_EINVAL, "ref_blur_pool: null pointer");
  const int p = (3 + stride) / 2 - 1;
  const int OH = (H + 2 * p - 3) / stride + 1, OW = (W + 2 * p - 3) / stride + 1;
  REF_LAUNCH(ref_blur_pool_kernel, grid_for((int64_t)B * OH * OW * C), 256, 0, stream, (const float*)x, (float*)y, B, H, W, C, stride, OH,
             OW, p);
  return 0;
}

int tfimm_hip_ref_dwconv(const void* x, const float* w, const float* bias, void* y, void* sum_out, int B, int H, int W, int C, int k,
                         int stride, int pad_t, int pad_l, int OH, int OW, int act, void* stream) {
  if (!x || !w || !y) TFIMM_FAIL(TFIMM_EINVAL, "ref_dwconv: null pointer");
  if (sum_out) TFIMM_FAIL(TFIMM_EUNSUP, "ref_dwconv: the squeeze is a separate mean_rows launch on the fp32 path");
  REF_LAUNCH(ref_dwconv_kernel, grid_for((int64_t)B * OH * OW * C), 256, 0, stream, (const float*)x, w, bias, (float*)y, B, H, W, C, k,
             stride, pad_t, pad_l, OH, OW, act);
  return 0;
}

int tfimm_hip_ref_group_norm(const void* x, const float* gamma, const float* beta, const void* residual, void* y, void* stats_ws, int B,
                             int rows, int C, int groups, float eps, int act, int act_after_res, void* stream) {
  (void)stats_ws;
  if (!x || !gamma || !beta || !y || groups <= 0 || C % groups) TFIMM_FAIL(TFIMM_EINVAL, "ref_group_norm: bad arguments");
  REF_LAUNCH(ref_group_norm_kernel, (unsigned)(B * groups), 256, 0, stream, (const float*)x, gamma, beta, (const float*)residual,
             (float*)y, rows, C, groups, eps, act, act_after_res);
  return 0;
}

int tfimm_hip_ref_attention(const tfimm_attn_desc* dp, void* stream) {
  if (!dp || !dp->qkv || !dp->out) TFIMM_FAIL(TFIMM_EINVAL, "ref_attention: null pointer");
  const tfimm_attn_desc& d = *dp;
  RefAttn p;
  p.qkv = (const float*)d.qkv; p.out = (float*)d.out; p.rel_bias = d.rel_bias;
  p.batch = d.batch; p.n_tokens = d.n_tokens; p.heads = d.heads; p.hd = d.hd; p.scale = d.scale;
  p.window = d.window; p.shift = d.shift; p.res_h = d.res_h; p.res_w = d.res_w;
  p.ld = 3 * d.heads * d.hd; p.dmodel = d.heads * d.hd;
  int nseq;
  if (d.window > 0) {
    if (d.res_h % d.window || d.res_w % d.window || d.res_h * d.res_w != d.n_tokens) TFIMM_FAIL(TFIMM_EINVAL, "ref_attention: window grid");
    p.n = d.window * d.window; p.nwx = d.res_w / d.window; p.nw = (d.res_h / d.window) * p.nwx;
    nseq = d.batch * p.nw;
  } else {
    p.n = d.n_tokens; p.nwx = p.nw = 1;
    nseq = d.batch;
  }
  const int64_t blocks = (int64_t)nseq * d.heads * p.n;
  if (blocks > 0x7fffffffLL || (size_t)p.n * 4 > 64 * 1024) TFIMM_FAIL(TFIMM_EUNSUP, "ref_attention: too large");
  REF_LAUNCH(ref_attention_kernel, (unsigned)blocks, 64, (size_t)p.n * 4, stream, p);
  return 0;
}

int tfimm_hip_ref_attention_probs(const void* qkv, void* probs, int B, int n_tokens, int heads, int hd, float scale, void* stream) {
  if (!qkv || !probs) TFIMM_FAIL(TFIMM_EINVAL, "ref_attention_probs: null pointer");
  const int64_t blocks = (int64_t)B * heads * n_tokens;
  if (blocks > 0x7fffffffLL || (size_t)n_tokens * 4 > 64 * 1024) TFIMM_FAIL(TFIMM_EUNSUP, "ref_attention_probs: too large");
  REF_LAUNCH(ref_attention_probs_kernel, (unsigned)blocks, 64, (size_t)n_tokens * 4, stream, (const float*)qkv, (float*)probs, n_tokens,
             heads, hd, scale);
  return 0;
}

int tfimm_hip_ref_talking_heads_attention(const tfimm_tha_desc* dp, void* stream) {
  if (!dp || !dp->qkv || !dp->out || !dp->proj_l_w || !dp->proj_l_b || !dp->proj_w_w || !dp->proj_w_b)
    TFIMM_FAIL(TFIMM_EINVAL, "ref_talking_heads_attention: null pointer");
  const tfimm_tha_desc& d = *dp;
  if (d.heads > 16) TFIMM_FAIL(TFIMM_EUNSUP, "ref_talking_heads_attention: heads = %d > 16", d.heads);
  RefThaW w;
  memset(&w, 0, sizeof(w));
  memcpy(w.wl, d.proj_l_w, sizeof(float) * d.heads * d.heads);
  memcpy(w.bl, d.proj_l_b, sizeof(float) * d.heads);
  memcpy(w.ww, d.proj_w_w, sizeof(float) * d.heads * d.heads);
  memcpy(w.bw, d.proj_w_b, sizeof(float) * d.heads);
  const size_t lds = (size_t)2 * d.heads * d.n_tokens * 4;
  if (lds > 160 * 1024) TFIMM_FAIL(TFIMM_EUNSUP, "ref_talking_heads_attention: %d tokens x %d heads do not fit in LDS", d.n_tokens, d.heads);
  static tfimm_once_t big_lds;
  if (big_lds.need()) {      // cait_m36_384 / cait_m48_448: 576 / 784 tokens x 16 heads need 72 / 98 KiB
    TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)ref_tha_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    big_lds.mark();
  }
  REF_LAUNCH(ref_tha_kernel, (unsigned)(d.batch * d.n_tokens), 256, lds, stream, (const float*)d.qkv, (float*)d.out, d.n_tokens, d.heads,
             d.hd, d.scale, w);
  return 0;
}

int tfimm_hip_ref_class_attention(const void* q, const void* kv, void* out, int B, int n_tokens, int heads, int hd, int ldq, int ldkv,
                                  int ldo, void* stream) {
  if (!q || !kv || !out) TFIMM_FAIL(TFIMM_EINVAL, "ref_class_attention: null pointer");
  if ((size_t)n_tokens * 4 > 64 * 1024) TFIMM_FAIL(TFIMM_EUNSUP, "ref_class_attention: too many tokens");
  REF_LAUNCH(ref_class_attn_kernel, (unsigned)(B * heads), 64, (size_t)n_tokens * 4, stream, (const float*)q, (const float*)kv, (float*)out,
             n_tokens, heads, hd, ldq, ldkv, ldo);
  return 0;
}

}  // extern "C"
