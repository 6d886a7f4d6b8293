// tfimm_hip_attention: fused multi-head attention for ViT/DeiT/CaiT-style global attention
// and Swin (shifted-)window attention.  See include/tfimm_hip.h for the reference call sites.
//
// Structure (gfx950): one workgroup = 4 waves = 64 query rows of one (sequence, head); each
// wave owns 16 query rows and walks the keys in blocks of 64 with an online softmax.
//   * S^T = K . Q^T via v_mfma_f32_16x16x32_bf16 with K as the "a" operand, so a lane holds
//     scores of ONE query (col = lane&15) for keys {16t + 4g + r}: the softmax row reduction
//     is 16 in-register values + two cross-lane steps (g = lane>>4).
//   * the bf16 P values are reused in place as the "b" operand of O^T = V^T . P^T; the MFMA
//     k index is the permuted key order {16*(2s) + 4g + r, 16*(2s+1) + 4g + r}, and V^T is
//     read from LDS in exactly that order (8-byte ds_read_b64 pairs), so no cross-lane
//     shuffle of P is ever needed.
//   * K block [64][HD+8] and V^T block [HD][72] live in LDS; both row strides are 36 dwords,
//     which makes the 16-row ds_read_b128 / ds_read_b64 fragment reads conflict free.
//   * Swin: tf.roll / window_partition / window_reverse are pure index maps applied when
//     rows are loaded and stored; the -100 shift mask is recomputed from region ids.
#include "common.h"

namespace {

struct AttnArgs {
  const bf16_t* qkv;
  bf16_t* out;
  const float* rel_bias;
  int batch, n_tokens, heads, hd;
  float scale;
  int window, shift, res_h, res_w;
  int n;        // tokens per sequence
  int nwx, nw;  // windows per grid row / per image
  int qchunks;
  int vec;      // hd % 8 == 0 and all rows 16-byte aligned
  int ld;       // 3 * heads * hd
  int dmodel;   // heads * hd
};

template <bool SWIN>
__device__ __forceinline__ int64_t token_row(const AttnArgs& p, int seq, int t, int* region) {
  if (!SWIN) {
    *region = 0;
    return (int64_t)seq * p.n_tokens + t;
  }
  const int b = seq / p.nw, w = seq - b * p.nw;
  const int wy = w / p.nwx, wx = w - wy * p.nwx;
  const int ty = t / p.window, tx = t - ty * p.window;
  const int ys = wy * p.window + ty, xs = wx * p.window + tx;  // shifted-frame coords
  int y = ys + p.shift, x = xs + p.shift;
  if (y >= p.res_h) y -= p.res_h;
  if (x >= p.res_w) x -= p.res_w;
  // region ids of swin.py:249-263 (slices (0,-ws), (-ws,-shift), (-shift,None))
  const int rh = ys < p.res_h - p.window ? 0 : (ys < p.res_h - p.shift ? 1 : 2);
  const int rw = xs < p.res_w - p.window ? 0 : (xs < p.res_w - p.shift ? 1 : 2);
  *region = rh * 3 + rw;
  return (int64_t)b * p.n_tokens + (int64_t)y * p.res_w + x;
}

template <int HD, bool SWIN>
__global__ void __launch_bounds__(256) attn_kernel(const AttnArgs p) {
  constexpr int KSTR = HD + 8;     // K row stride (elements)
  constexpr int VSTR = 72;         // V^T row stride (elements): 64 keys + 8 pad
  constexpr int CH = HD / 8;       // 16-byte chunks per head row
  constexpr int KS = HD / 32;      // MFMA k-steps over head dim
  constexpr int DT = HD / 16;      // output d tiles
  __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * KSTR];
  __shared__ __attribute__((aligned(16))) bf16_t Vt[HD * VSTR];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;

  int bid = blockIdx.x;
  const int qc = bid % p.qchunks; bid /= p.qchunks;
  const int h = bid % p.heads;
  const int seq = bid / p.heads;

  const int q = qc * 64 + wave * 16 + l15;  // this lane's query token
  const bool q_ok = q < p.n;
  int q_region = 0;
  const int64_t q_row = token_row<SWIN>(p, seq, q_ok ? q : 0, &q_region);

  // ---- Q fragments straight from global: lane (q, g) holds d = ks*32 + g*8 .. +8 ----
  bf16x8 qf[KS];
  {
    const bf16_t* qp = p.qkv + q_row * p.ld + h * p.hd;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = ks * 32 + g * 8;
      uint4 u = make_uint4(0u, 0u, 0u, 0u);
      if (q_ok) {
        if (p.vec) {
          if (d0 < p.hd) u = *reinterpret_cast<const uint4*>(qp + d0);
        } else {
          uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const uint32_t v = (d0 + e < p.hd) ? (uint32_t)qp[d0 + e] : 0u;
            w[e >> 1] |= v << ((e & 1) * 16);
          }
          u = make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
      qf[ks] = __builtin_bit_cast(bf16x8, u);
    }
  }

  f32x4 o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;

  for (int kb = 0; kb < p.n; kb += 64) {
    __syncthreads();  // previous block fully consumed
    // ---- stage K block (row major) and V block (transposed) ----
    for (int id = tid; id < 64 * CH; id += 256) {
      const int key = id / CH, c = id - key * CH;
      const int t = kb + key;
      uint4 ku = make_uint4(0u, 0u, 0u, 0u), vu = make_uint4(0u, 0u, 0u, 0u);
      if (t < p.n && c * 8 < p.hd) {
        int rg;
        const int64_t row = token_row<SWIN>(p, seq, t, &rg);
        const bf16_t* kp = p.qkv + row * p.ld + p.dmodel + h * p.hd + c * 8;
        const bf16_t* vp = kp + p.dmodel;
        if (p.vec) {
          ku = *reinterpret_cast<const uint4*>(kp);
          vu = *reinterpret_cast<const uint4*>(vp);
        } else {
          uint32_t kw[4] = {0u, 0u, 0u, 0u}, vw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const bool ok = c * 8 + e < p.hd;
            const uint32_t kv = ok ? (uint32_t)kp[e] : 0u;
            const uint32_t vv = ok ? (uint32_t)vp[e] : 0u;
            kw[e >> 1] |= kv << ((e & 1) * 16);
            vw[e >> 1] |= vv << ((e & 1) * 16);
          }
          ku = make_uint4(kw[0], kw[1], kw[2], kw[3]);
          vu = make_uint4(vw[0], vw[1], vw[2], vw[3]);
        }
      }
      *reinterpret_cast<uint4*>(&Ks[key * KSTR + c * 8]) = ku;
      const uint32_t vw[4] = {vu.x, vu.y, vu.z, vu.w};
#pragma unroll
      for (int e = 0; e < 8; ++e)
        Vt[(c * 8 + e) * VSTR + key] = (bf16_t)((vw[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
    }
    __syncthreads();

    // ---- S^T[key][q] for the 4 key tiles of this block ----
    float s[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint4 ku = *reinterpret_cast<const uint4*>(&Ks[(t * 16 + l15) * KSTR + ks * 32 + g * 8]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ku), qf[ks], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kb + t * 16 + g * 4 + r;
        float v = acc[r] * p.scale;
        if (key < p.n) {
          if (SWIN && q_ok) {
            if (p.rel_bias) v += p.rel_bias[((size_t)h * p.n + q) * p.n + key];
            if (p.shift > 0) {
              int kr;
              (void)token_row<SWIN>(p, seq, key, &kr);
              if (kr != q_region) v += -100.0f;
            }
          }
        } else {
          v = -1e30f;
        }
        s[t][r] = v;
      }
    }

    // ---- online softmax (row = this lane's query; keys spread over regs and g) ----
    float mloc = s[0][0];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) mloc = fmaxf(mloc, s[t][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(s[t][r] - m_new);
        s[t][r] = e;
        psum += e;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i] *= alpha;

    // ---- O^T[d][q] += V^T[d][key] . P^T[key][q] ----
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      uint4 pu;
      pu.x = pack_bf2(s[2 * s2][0], s[2 * s2][1]);
      pu.y = pack_bf2(s[2 * s2][2], s[2 * s2][3]);
      pu.z = pack_bf2(s[2 * s2 + 1][0], s[2 * s2 + 1][1]);
      pu.w = pack_bf2(s[2 * s2 + 1][2], s[2 * s2 + 1][3]);
      const bf16x8 pf = __builtin_bit_cast(bf16x8, pu);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const bf16_t* vrow = &Vt[(dt * 16 + l15) * VSTR];
        const uint2 v0 = *reinterpret_cast<const uint2*>(vrow + (2 * s2) * 16 + g * 4);
        const uint2 v1 = *reinterpret_cast<const uint2*>(vrow + (2 * s2 + 1) * 16 + g * 4);
        const uint4 vu = make_uint4(v0.x, v0.y, v1.x, v1.y);
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vu), pf, o[dt], 0, 0, 0);
      }
    }
  }

  // ---- normalise and store: lane (q, g) holds d = dt*16 + g*4 + r ----
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  if (!q_ok) return;
  const float inv = 1.f / l_tot;
  bf16_t* op = p.out + q_row * p.dmodel + h * p.hd;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    const int d0 = dt * 16 + g * 4;
    if (d0 >= p.hd) continue;
    const float v0 = o[dt][0] * inv, v1 = o[dt][1] * inv, v2 = o[dt][2] * inv, v3 = o[dt][3] * inv;
    if (p.vec) {
      *reinterpret_cast<uint2*>(op + d0) = make_uint2(pack_bf2(v0, v1), pack_bf2(v2, v3));
    } else {
      const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (d0 + e < p.hd) op[d0 + e] = (bf16_t)f2bf(vv[e]);
    }
  }
}

}  // namespace

extern "C" int tfimm_hip_attention(const tfimm_attn_desc* dp, void* stream) {
  if (!dp) TFIMM_FAIL(TFIMM_EINVAL, "attention: null descriptor");
  const tfimm_attn_desc& d = *dp;
  if (!d.qkv || !d.out) TFIMM_FAIL(TFIMM_EINVAL, "attention: null pointer");
  if (d.batch <= 0 || d.n_tokens <= 0 || d.heads <= 0 || d.hd <= 0)
    TFIMM_FAIL(TFIMM_EINVAL, "attention: bad shape");
  if (d.hd > 64) TFIMM_FAIL(TFIMM_EUNSUP, "attention: head dim %d > 64 not built", d.hd);
  AttnArgs a;
  a.qkv = (const bf16_t*)d.qkv; a.out = (bf16_t*)d.out; a.rel_bias = d.rel_bias;
  a.batch = d.batch; a.n_tokens = d.n_tokens; a.heads = d.heads; a.hd = d.hd; a.scale = d.scale;
  a.window = d.window; a.shift = d.shift; a.res_h = d.res_h; a.res_w = d.res_w;
  a.dmodel = d.heads * d.hd; a.ld = 3 * a.dmodel;
  int64_t nseq;
  if (d.window > 0) {
    if (d.res_h <= 0 || d.res_w <= 0 || d.res_h % d.window || d.res_w % d.window ||
        d.res_h * d.res_w != d.n_tokens || d.shift < 0 || d.shift >= d.window)
      TFIMM_FAIL(TFIMM_EINVAL, "attention: bad window geometry");
    a.n = d.window * d.window;
    a.nwx = d.res_w / d.window;
    a.nw = a.nwx * (d.res_h / d.window);
    nseq = (int64_t)d.batch * a.nw;
  } else {
    a.n = d.n_tokens; a.nwx = 1; a.nw = 1;
    nseq = d.batch;
  }
  a.qchunks = (a.n + 63) / 64;
  a.vec = ((d.hd & 7) == 0) && (((uintptr_t)d.qkv & 15) == 0) && (((uintptr_t)d.out & 7) == 0);
  const int64_t nblocks = nseq * d.heads * a.qchunks;
  if (nblocks > 0x7fffffffLL) TFIMM_FAIL(TFIMM_EINVAL, "attention: grid too large");
  const dim3 grid((unsigned)nblocks), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (d.window > 0) {
    if (d.hd <= 32) TFIMM_LAUNCH((attn_kernel<32, true>), grid, block, 0, st, a);
    else TFIMM_LAUNCH((attn_kernel<64, true>), grid, block, 0, st, a);
  } else {
    if (d.hd <= 32) TFIMM_LAUNCH((attn_kernel<32, false>), grid, block, 0, st, a);
    else TFIMM_LAUNCH((attn_kernel<64, false>), grid, block, 0, st, a);
  }
  return 0;
}
