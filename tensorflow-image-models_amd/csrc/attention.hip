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

#include <cstdlib>

namespace {

struct AttnArgs {
  const bf16_t* qkv;
  bf16_t* out;
  const float* rel_bias;
  const float* bias_log2;   // optional pre-combined (bias + mask) * log2e tiles, see tfimm_hip.h
  int batch, n_tokens, heads, hd;
  float scale;
  int window, shift, res_h, res_w;
  int n;        // tokens per sequence
  int nwx, nw;  // windows per grid row / per image
  int qchunks;
  int vec;      // hd % 8 == 0 and all rows 16-byte aligned
  int ld;       // 3 * heads * hd
  int dmodel;   // heads * hd
  int dbg;      // TFIMM_ATTN_DBG (profiling only): 1 skip the key loop, 2 skip K/V staging
  int xcd_map;  // resident kernel: XCD-contiguous (sequence, head) ranges (TFIMM_ATTN_XCD=0 switches it off)
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


// ---------------------------------------------------------------------------------------------
// Resident variant (sequences whose K and V^T fit in LDS): ONE workgroup per (sequence, head).
//   * K [n][HD] (16-byte XOR-swizzled rows for HD = 64) and V^T [HD][n] are staged once -- V^T by
//     packed 32-bit stores of key PAIRS (lane = pair: consecutive lanes, consecutive banks) -- then
//     every wave walks its own 16-query tiles with no further workgroup synchronisation.
//   * the softmax runs in the exp2 domain with the scale folded into one FMA per score
//     (p = exp2(s * scale*log2e - m)), the running max is taken on raw scores, keys beyond n are
//     masked only in the last key block, and all LDS addresses are base + immediate.  The first
//     version of this loop issued ~320 VALU instructions per 16x64 score block and was VALU-bound
//     (SQ_ACTIVE_INST_VALU 61 % of the kernel, MFMA 12 %).
//   * Swin: relative-position bias AND the -100 shift mask (swin.py:175-190) are combined per
//     workgroup into one fp32 tile [n][nkp] in LDS (pre-multiplied by log2e), from per-key region ids
//     computed once -- the inner loop adds it with the same FMA, no divisions, no global loads.
// ---------------------------------------------------------------------------------------------
template <int HD>
__device__ __forceinline__ int k_slot(int row, int chunk) {   // 16-byte slot of (key row, d chunk) in Ks
  if (HD == 64) return row * 8 + (chunk ^ ((row >> 1) & 7));  // 128-byte rows: the GEMM swizzle
  return row * (HD / 8 + 1) + chunk;                          // HD = 32: 80-byte padded rows
}

template <int HD>
__device__ __forceinline__ int v_slot(int row, int chunk) {   // 16-byte slot of (key row, d chunk) in Vs
  // row-major V, read back TRANSPOSED by ds_read_b64_tr_b16 (a 16-lane group fetches a 4-key x 16-d block
  // as 8-byte pieces and every lane receives one d column of 4 keys); the XOR keeps the 8 rows that the
  // two groups of a 32-lane half touch on distinct 16-byte slots of the 256-byte bank row
  if (HD == 64) return row * 8 + (chunk ^ (((row >> 1) & 3) << 1));
  return row * (HD / 8) + (chunk ^ (((row >> 2) & 1) << 1));
}

template <int HD, bool SWIN, int NW, int TQ>
__global__ void __launch_bounds__(NW * 64) attn_resident_kernel(const AttnArgs p, const int nkp) {
  constexpr int NT = NW * 64;
  constexpr int CH = HD / 8;       // 16-byte chunks per head row
  constexpr int KROW = HD == 64 ? 8 : CH + 1;   // slots per K row
  constexpr int KS = HD / 32;      // MFMA k-steps over head dim
  constexpr int DT = HD / 16;      // output d tiles
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) char smem_attn[];
  uint4* Ks = reinterpret_cast<uint4*>(smem_attn);               // [nkp][KROW] 16-byte slots
  uint4* Vs = Ks + (size_t)nkp * KROW;                            // [nkp][CH] 16-byte slots, row-major V
  float* Bs = reinterpret_cast<float*>(Vs + (size_t)nkp * CH);    // SWIN: [n][nkp] bias+mask (log2 units)
  const bool tiles = SWIN && p.bias_log2 != nullptr;            // pre-combined bias tiles: no Bs staging
  int* Rg = reinterpret_cast<int*>(Bs + ((SWIN && !tiles) ? (size_t)p.n * nkp : 0));   // SWIN: [n] region ids
  int* Rw = Rg + (SWIN ? p.n : 0);                                // SWIN: [n] global row of window token t

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  // Workgroups are dealt to the 8 XCDs round-robin; the heads of one sequence read 64 / 128-byte slices of the SAME token
  // rows (Swin: two or four heads per 128-byte line), so consecutive (sequence, head) items go to ONE XCD -- each XCD owns
  // a contiguous range of items and a row is pulled through one L2 instead of up to eight.
  int item = blockIdx.x;
  if (p.xcd_map) {
    const int G = (int)gridDim.x, xcd = item & 7, idx = item >> 3;
    item = xcd * (G >> 3) + min(xcd, G & 7) + idx;
  }
  const int h = item % p.heads;
  const int seq = item / p.heads;

  // ---- Swin: global row (and region id) of every token of this window, once per workgroup -- the
  //      roll / window_partition index map costs several integer divisions per token
  if (SWIN) {
    for (int t = tid; t < p.n; t += NT) {
      int rg;
      Rw[t] = (int)token_row<SWIN>(p, seq, t, &rg);
      Rg[t] = rg;
    }
    __syncthreads();
  }
  auto row_of = [&](int t) -> int64_t {
    if (SWIN) return (int64_t)Rw[t];
    return (int64_t)seq * p.n_tokens + t;
  };
  // ---- Q fragments first: their global loads overlap the K / V staging below
  // Every wave owns TQ (1 or 2) 16-query tiles, qt = wave (and wave + NW); the launcher picks NW, TQ so
  // that NW * TQ tiles cover the sequence.  With two tiles the keys are walked ONCE for both: two
  // independent softmax / MFMA chains per wave, K and V fragments read once.
  int qi[TQ];
  bool q_ok[TQ];
  int64_t q_row[TQ];
#pragma unroll
  for (int u = 0; u < TQ; ++u) {
    qi[u] = (wave + u * NW) * 16 + l15;
    q_ok[u] = qi[u] < p.n;
    q_row[u] = row_of(q_ok[u] ? qi[u] : 0);
  }
  bf16x8 qf[TQ][KS];
#pragma unroll
  for (int u = 0; u < TQ; ++u) {
    const bf16_t* qp = p.qkv + q_row[u] * p.ld + h * p.hd;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = ks * 32 + g * 8;
      uint4 uu = make_uint4(0u, 0u, 0u, 0u);
      if (q_ok[u]) {
        if (p.vec) {
          if (d0 < p.hd) uu = *reinterpret_cast<const uint4*>(qp + d0);
        } else {
          uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const uint32_t v = (d0 + e < p.hd) ? (uint32_t)qp[d0 + e] : 0u;
            w[e >> 1] |= v << ((e & 1) * 16);
          }
          uu = make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
      qf[u][ks] = __builtin_bit_cast(bf16x8, uu);
    }
  }

  // ---- stage K (16-byte stores)
  for (int id = (TFIMM_PROBE(p.dbg) & 2) ? nkp * CH : tid; id < nkp * CH; id += NT) {
    const int key = id / CH, c = id - key * CH;
    uint4 ku = make_uint4(0u, 0u, 0u, 0u);
    if (key < p.n && c * 8 < p.hd) {
      const int64_t row = row_of(key);
      const bf16_t* kp = p.qkv + row * p.ld + p.dmodel + h * p.hd + c * 8;
      if (p.vec) {
        ku = *reinterpret_cast<const uint4*>(kp);
      } else {
        uint32_t kw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t kv = (c * 8 + e < p.hd) ? (uint32_t)kp[e] : 0u;
          kw[e >> 1] |= kv << ((e & 1) * 16);
        }
        ku = make_uint4(kw[0], kw[1], kw[2], kw[3]);
      }
    }
    Ks[k_slot<HD>(key, c)] = ku;
  }
  // ---- stage V row-major (16-byte stores, same coalescing as K)
  for (int id = (TFIMM_PROBE(p.dbg) & 2) ? nkp * CH : tid; id < nkp * CH; id += NT) {
    const int key = id / CH, c = id - key * CH;
    uint4 vu = make_uint4(0u, 0u, 0u, 0u);
    if (key < p.n && c * 8 < p.hd) {
      const int64_t row = row_of(key);
      const bf16_t* vp = p.qkv + row * p.ld + 2 * p.dmodel + h * p.hd + c * 8;
      if (p.vec) {
        vu = *reinterpret_cast<const uint4*>(vp);
      } else {
        uint32_t vw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t vv = (c * 8 + e < p.hd) ? (uint32_t)vp[e] : 0u;
          vw[e >> 1] |= vv << ((e & 1) * 16);
        }
        vu = make_uint4(vw[0], vw[1], vw[2], vw[3]);
      }
    }
    Vs[v_slot<HD>(key, c)] = vu;
  }
  if (SWIN && !tiles) {
    const bool masked = p.shift > 0;
    for (int id = tid; id < p.n * nkp; id += NT) {
      const int q = id / nkp, key = id - q * nkp;
      float b = 0.f;
      if (key < p.n) {
        if (p.rel_bias) b = p.rel_bias[((size_t)h * p.n + q) * p.n + key] * LOG2E;
        if (masked && Rg[q] != Rg[key]) b += -100.0f * LOG2E;
      }
      Bs[id] = b;
    }
  }
  __syncthreads();

  const float cs = p.scale * LOG2E;
  const int nkb = (p.n + 63) >> 6;
  f32x4 o[TQ][DT];
  float m_run[TQ], l_run[TQ];               // running max / sum, log2 units
  const float* blane[TQ];
#pragma unroll
  for (int u = 0; u < TQ; ++u) {
#pragma unroll
    for (int i = 0; i < DT; ++i) o[u][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    m_run[u] = -1e30f;
    l_run[u] = 0.f;
    blane[u] = Bs + (size_t)(q_ok[u] ? qi[u] : 0) * nkp + g * 4;
    if (tiles) {   // window kind: 2 * (last window row) + (last window column) when shifted
      const int w = seq % p.nw;
      const int wy = w / p.nwx, wx = w - wy * p.nwx;
      const int kind = p.shift > 0 ? 2 * (wy == p.nw / p.nwx - 1 ? 1 : 0) + (wx == p.nwx - 1 ? 1 : 0) : 0;
      blane[u] = p.bias_log2 + (((size_t)kind * p.heads + h) * p.n + (q_ok[u] ? qi[u] : 0)) * nkp + g * 4;
    }
  }
  const bool two = TQ == 2 && (wave + NW) * 16 < p.n;   // second tile present (wave-uniform)

  if (wave * 16 < p.n) {
    for (int kb = (TFIMM_PROBE(p.dbg) & 1) ? nkb : 0; kb < nkb; ++kb) {
      // ---- raw scores S^T[key][q] of the 4 key tiles of this block, both query tiles
      f32x4 acc[TQ][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int u = 0; u < TQ; ++u) acc[u][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 kf = __builtin_bit_cast(bf16x8, Ks[k_slot<HD>(kb * 64 + t * 16 + l15, ks * 4 + g)]);
#pragma unroll
          for (int u = 0; u < TQ; ++u)
            if (u == 0 || two) acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[u][ks], acc[u][t], 0, 0, 0);
        }
      }
      bf16x8 pf[TQ][2];
#pragma unroll
      for (int u = 0; u < TQ; ++u) {
        if (u == 1 && !two) break;
        float sc[16];
        if (SWIN) {   // log2-domain logits: s * scale * log2e + (bias + mask) * log2e
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float4 b4 = *reinterpret_cast<const float4*>(blane[u] + kb * 64 + t * 16);
            sc[t * 4 + 0] = fmaf(acc[u][t][0], cs, b4.x); sc[t * 4 + 1] = fmaf(acc[u][t][1], cs, b4.y);
            sc[t * 4 + 2] = fmaf(acc[u][t][2], cs, b4.z); sc[t * 4 + 3] = fmaf(acc[u][t][3], cs, b4.w);
          }
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[t * 4 + r] = acc[u][t][r];
        }
        if (kb * 64 + 64 > p.n) {   // last block: keys beyond n never win the max and get p = 0
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (kb * 64 + t * 16 + g * 4 + r >= p.n) sc[t * 4 + r] = -__builtin_inff();
        }
        // ---- online softmax, exp2 domain.  Non-Swin: sc are RAW scores, the scale rides on the FMA.
        const float mul = SWIN ? 1.f : cs;
        float mloc = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) mloc = fmaxf(fmaxf(mloc, sc[i]), sc[i + 1]);
        mloc = fmaxf(mloc, sc[15]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run[u], mloc * mul);
        const float alpha = __builtin_amdgcn_exp2f(m_run[u] - m_new);
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          sc[i] = __builtin_amdgcn_exp2f(fmaf(sc[i], mul, -m_new));
          psum += sc[i];
        }
        l_run[u] = l_run[u] * alpha + psum;
        m_run[u] = m_new;
#pragma unroll
        for (int i = 0; i < DT; ++i) o[u][i] *= alpha;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          uint4 pu;
          pu.x = pack_bf2(sc[8 * s2 + 0], sc[8 * s2 + 1]);
          pu.y = pack_bf2(sc[8 * s2 + 2], sc[8 * s2 + 3]);
          pu.z = pack_bf2(sc[8 * s2 + 4], sc[8 * s2 + 5]);
          pu.w = pack_bf2(sc[8 * s2 + 6], sc[8 * s2 + 7]);
          pf[u][s2] = __builtin_bit_cast(bf16x8, pu);
        }
      }
      // ---- O^T[d][q] += V^T[d][key] . P^T[key][q]   (V^T fragments read once for both tiles)
      typedef __attribute__((ext_vector_type(4))) short s16x4;
      typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          // lane (l15, g) fetches the 8-byte piece (key 4g + l15/4, d 4*(l15%4)..+3) of the two 16-key
          // tiles and receives keys 4g..4g+3 at d = l15 of each: exactly the MFMA A fragment of V^T
          const int krow = kb * 64 + g * 4 + (l15 >> 2);
          const int chunk = dt * 2 + ((l15 & 3) >> 1);
          const char* base = reinterpret_cast<const char*>(Vs) + (l15 & 1) * 8;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (lds_s16x4_ptr)(base + (size_t)v_slot<HD>(krow + (2 * s2) * 16, chunk) * 16));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (lds_s16x4_ptr)(base + (size_t)v_slot<HD>(krow + (2 * s2 + 1) * 16, chunk) * 16));
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 cat = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          const bf16x8 vf = __builtin_bit_cast(bf16x8, cat);
#pragma unroll
          for (int u = 0; u < TQ; ++u)
            if (u == 0 || two) o[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[u][s2], o[u][dt], 0, 0, 0);
        }
      }
    }

#pragma unroll
    for (int u = 0; u < TQ; ++u) {
      float l_tot = l_run[u] + __shfl_xor(l_run[u], 16, 64);
      l_tot += __shfl_xor(l_tot, 32, 64);
      if (q_ok[u]) {
        const float inv = 1.f / l_tot;
        bf16_t* op = p.out + q_row[u] * p.dmodel + h * p.hd;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int d0 = dt * 16 + g * 4;
          if (d0 >= p.hd) continue;
          const float v0 = o[u][dt][0] * inv, v1 = o[u][dt][1] * inv, v2 = o[u][dt][2] * inv, v3 = o[u][dt][3] * inv;
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
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Global attention (ViT / DeiT / CaiT main blocks), persistent: one workgroup per CU walks a list of
// (image, head) items.  While item i is being computed from LDS, the K and V rows and the Q fragments of
// item i + 1 are already on their way from HBM into registers (8 + 8 x 16 bytes per thread at n <= 256);
// they are written to LDS when item i is done.  Measured on ViT-B/16 (batch 512): the one-item-per-workgroup
// kernel above spends 91 us in Q loads / O stores / launches, +59 us in K/V staging, +82 us in arithmetic,
// strictly one after the other (232 us; the HBM floor is ~120 us).  This kernel: 16 waves x one query tile (8 x 2
// leaves two waves per SIMD and is slower), 215 us -- the arithmetic (VALU-bound on the exponentials of the padded
// 256 x 256 score block) now hides the loads, and is what is left.
// Same arithmetic as attn_resident_kernel (vec layout only: hd == HD, 16-byte aligned rows).
// ---------------------------------------------------------------------------------------------
// K / V pieces and Q fragments of one (image, head) item, held in registers while the previous item is computed.
// Named members, not arrays: hipcc left the array version of this in scratch memory (store after every load = a
// wait for it), which serialised exactly what is meant to overlap.
typedef __attribute__((ext_vector_type(4))) unsigned int attn_u32x4;    // (a vector, not HIP's uint4 struct)
struct AttnPrefetch {
  attn_u32x4 k0, k1, k2, k3, v0, v1, v2, v3;
  attn_u32x4 q00, q01, q10, q11;     // [query tile][k-step]
};

// Unconditional loads from clamped rows: a select on a loaded value would make hipcc wait for it during the arithmetic.
template <int HD, int PF, int KS, int NT, int TQ>
__device__ __forceinline__ void attn_fetch_item(const AttnArgs& p, int item, int tid, int q0, int q1, int g, AttnPrefetch& r) {
  constexpr int CH = HD / 8;
  const int h = item % p.heads, seq = item / p.heads;
  const bf16_t* base = p.qkv + (int64_t)seq * p.n_tokens * p.ld + h * HD;
  auto piece = [&](int j, attn_u32x4& k, attn_u32x4& v) __attribute__((always_inline)) {
    const int id = tid + j * NT;
    const int key = id / CH, c = id - key * CH;
    const bf16_t* kp = base + (int64_t)min(key, p.n - 1) * p.ld + p.dmodel + c * 8;
    k = *reinterpret_cast<const attn_u32x4*>(kp);
    v = *reinterpret_cast<const attn_u32x4*>(kp + p.dmodel);
  };
  piece(0, r.k0, r.v0);
  if (PF > 1) piece(1, r.k1, r.v1);
  if (PF > 2) piece(2, r.k2, r.v2);
  if (PF > 3) piece(3, r.k3, r.v3);
  const bf16_t* qa = base + (int64_t)min(q0, p.n - 1) * p.ld + g * 8;
  r.q00 = *reinterpret_cast<const attn_u32x4*>(qa);
  if (KS > 1) r.q01 = *reinterpret_cast<const attn_u32x4*>(qa + 32);
  if (TQ > 1) {
    const bf16_t* qb = base + (int64_t)min(q1, p.n - 1) * p.ld + g * 8;
    r.q10 = *reinterpret_cast<const attn_u32x4*>(qb);
    if (KS > 1) r.q11 = *reinterpret_cast<const attn_u32x4*>(qb + 32);
  }
}

template <int HD, int NW, int TQ>
__global__ void __launch_bounds__(NW * 64) attn_stream_kernel(const AttnArgs p, const int nkp, const int items) {
  constexpr int NT = NW * 64;
  static_assert(NW * TQ == 16, "16 query tiles: n <= 256");
  constexpr int CH = HD / 8;
  constexpr int KROW = HD == 64 ? 8 : CH + 1;
  constexpr int KS = HD / 32;
  constexpr int DT = HD / 16;
  constexpr int PF = 256 * CH / NT;          // 16-byte K (and V) pieces per thread for nkp = 256
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) char smem_attn[];
  uint4* Ks = reinterpret_cast<uint4*>(smem_attn);
  uint4* Vs = Ks + (size_t)nkp * KROW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const float cs = p.scale * LOG2E;
  const int nkb = (p.n + 63) >> 6;
  const bool two = TQ == 2 && (wave + NW) * 16 < p.n;   // second query tile present (wave-uniform)
  const bool active = wave * 16 < p.n;

  int qi[TQ];
  bool q_ok[TQ];
#pragma unroll
  for (int u = 0; u < TQ; ++u) {
    qi[u] = (wave + u * NW) * 16 + l15;
    q_ok[u] = qi[u] < p.n;
  }
  AttnPrefetch pre;
  int item = blockIdx.x;
  if (item < items) attn_fetch_item<HD, PF, KS, NT, TQ>(p, item, tid, qi[0], qi[TQ - 1], g, pre);
  for (; item < items; item += gridDim.x) {
    __syncthreads();                          // every wave is done reading the previous item's K / V
    // rows n .. nkp-1 hold copies of row n-1: their scores are masked to -inf below and their P is 0
    auto put = [&](int j, const attn_u32x4 k, const attn_u32x4 v) __attribute__((always_inline)) {
      const int id = tid + j * NT;
      const int key = id / CH, c = id - key * CH;
      if (key < nkp) {
        reinterpret_cast<attn_u32x4*>(Ks)[k_slot<HD>(key, c)] = k;
        reinterpret_cast<attn_u32x4*>(Vs)[v_slot<HD>(key, c)] = v;
      }
    };
    put(0, pre.k0, pre.v0);
    if (PF > 1) put(1, pre.k1, pre.v1);
    if (PF > 2) put(2, pre.k2, pre.v2);
    if (PF > 3) put(3, pre.k3, pre.v3);
    bf16x8 qf[TQ][KS];             // (queries beyond n: a copy of query n-1, never stored)
    qf[0][0] = __builtin_bit_cast(bf16x8, pre.q00);
    if (KS > 1) qf[0][KS - 1] = __builtin_bit_cast(bf16x8, pre.q01);
    if (TQ > 1) {
      qf[TQ - 1][0] = __builtin_bit_cast(bf16x8, pre.q10);
      if (KS > 1) qf[TQ - 1][KS - 1] = __builtin_bit_cast(bf16x8, pre.q11);
    }
    __syncthreads();
    if (item + (int)gridDim.x < items && !(TFIMM_PROBE(p.dbg) & 2))          // in flight during the arithmetic below
      attn_fetch_item<HD, PF, KS, NT, TQ>(p, item + gridDim.x, tid, qi[0], qi[TQ - 1], g, pre);

    const int h = item % p.heads, seq = item / p.heads;
    f32x4 o[TQ][DT];
    float m_run[TQ], l_run[TQ];
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
#pragma unroll
      for (int i = 0; i < DT; ++i) o[u][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      m_run[u] = -1e30f;
      l_run[u] = 0.f;
    }
    if (active) {
      // One block of 64 keys per iteration.  In the last block, when n is not a multiple of 64, only the first `nt` 16-key
      // sub-tiles hold keys (n = 197: one of four): the score MFMAs, the exponentials and the P.V k-steps of the others are
      // skipped (wave-uniform branches); what they would have contributed are exact zeros (P = 0), so the result is
      // bit-identical to multiplying the padding.
      for (int kb = (TFIMM_PROBE(p.dbg) & 1) ? nkb : 0; kb < nkb; ++kb) {
        const int nt = min(4, (p.n - kb * 64 + 15) >> 4);      // wave-uniform
        const bool PART = nt < 4 || kb * 64 + 64 > p.n;
        f32x4 acc[TQ][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
          for (int u = 0; u < TQ; ++u) acc[u][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (PART && t >= nt) continue;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 kf = __builtin_bit_cast(bf16x8, Ks[k_slot<HD>(kb * 64 + t * 16 + l15, ks * 4 + g)]);
#pragma unroll
            for (int u = 0; u < TQ; ++u)
              if (u == 0 || two) acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[u][ks], acc[u][t], 0, 0, 0);
          }
        }
        bf16x8 pf[TQ][2];
#pragma unroll
        for (int u = 0; u < TQ; ++u) {
          if (u == 1 && !two) break;
          float sc[16];
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[t * 4 + r] = acc[u][t][r];
          if (PART) {   // keys beyond n never win the max and get p = 0
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (kb * 64 + t * 16 + g * 4 + r >= p.n) sc[t * 4 + r] = -__builtin_inff();
          }
          // online softmax, exp2 domain: sc are RAW scores, the scale rides on the FMA
          float mloc = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
#pragma unroll
          for (int i = 3; i < 15; i += 2) mloc = fmaxf(fmaxf(mloc, sc[i]), sc[i + 1]);
          mloc = fmaxf(mloc, sc[15]);
          mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
          mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
          const float m_new = fmaxf(m_run[u], mloc * cs);
          const float alpha = __builtin_amdgcn_exp2f(m_run[u] - m_new);
          float psum = 0.f;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (PART && t >= nt) {
#pragma unroll
              for (int r = 0; r < 4; ++r) sc[t * 4 + r] = 0.f;
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                sc[t * 4 + r] = __builtin_amdgcn_exp2f(fmaf(sc[t * 4 + r], cs, -m_new));
                psum += sc[t * 4 + r];
              }
            }
          }
          l_run[u] = l_run[u] * alpha + psum;
          m_run[u] = m_new;
#pragma unroll
          for (int i = 0; i < DT; ++i) o[u][i] *= alpha;
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            uint4 pu;
            pu.x = pack_bf2(sc[8 * s2 + 0], sc[8 * s2 + 1]);
            pu.y = pack_bf2(sc[8 * s2 + 2], sc[8 * s2 + 3]);
            pu.z = pack_bf2(sc[8 * s2 + 4], sc[8 * s2 + 5]);
            pu.w = pack_bf2(sc[8 * s2 + 6], sc[8 * s2 + 7]);
            pf[u][s2] = __builtin_bit_cast(bf16x8, pu);
          }
        }
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          if (PART && 2 * s2 >= nt) continue;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const int krow = kb * 64 + g * 4 + (l15 >> 2);
            const int chunk = dt * 2 + ((l15 & 3) >> 1);
            const char* base = reinterpret_cast<const char*>(Vs) + (l15 & 1) * 8;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (lds_s16x4_ptr)(base + (size_t)v_slot<HD>(krow + (2 * s2) * 16, chunk) * 16));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (lds_s16x4_ptr)(base + (size_t)v_slot<HD>(krow + (2 * s2 + 1) * 16, chunk) * 16));
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            const s16x8 cat = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            const bf16x8 vf = __builtin_bit_cast(bf16x8, cat);
#pragma unroll
            for (int u = 0; u < TQ; ++u)
              if (u == 0 || two) o[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[u][s2], o[u][dt], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < TQ; ++u) {
        float l_tot = l_run[u] + __shfl_xor(l_run[u], 16, 64);
        l_tot += __shfl_xor(l_tot, 32, 64);
        if (q_ok[u]) {
          const float inv = 1.f / l_tot;
          bf16_t* op = p.out + ((int64_t)seq * p.n_tokens + qi[u]) * p.dmodel + h * HD;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const int d0 = dt * 16 + g * 4;
            *reinterpret_cast<uint2*>(op + d0) = make_uint2(pack_bf2(o[u][dt][0] * inv, o[u][dt][1] * inv),
                                                            pack_bf2(o[u][dt][2] * inv, o[u][dt][3] * inv));
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Swin window attention, several heads per workgroup (hd = 32, n <= 64 tokens per window, pre-combined bias tiles).
// attn_resident_kernel runs one (window, head) per workgroup: 16384 workgroups of ~5 us for a Swin-B stage-3 layer, each
// a serial chain (index map -> Q / K / V loads -> LDS -> scores -> softmax -> P.V -> store) whose loads nothing overlaps, and
// the roll / window_partition index map is recomputed for each of a window's heads.  Here a workgroup owns a window and a
// RANGE of its heads: the index map is computed once, and while head h is being multiplied out of LDS the Q fragment and
// the one K and one V chunk each thread stages for head h + 1 are already on their way into registers (12 VGPRs), written
// to LDS behind the barrier that ends head h.  Same arithmetic, same LDS layouts and the same transposing V reads as the
// resident kernel (TQ = 1: a wave owns one 16-query tile).
template <int HD>
__global__ void __launch_bounds__(256) attn_window_kernel(const AttnArgs p, const int hsplit) {
  constexpr int NT = 256, NW = 4;
  constexpr int CH = HD / 8, KROW = CH + 1, DT = HD / 16;
  static_assert(HD == 32, "window kernel: head dim 32");
  constexpr int NKP = 64;
  constexpr float LOG2E = 1.4426950408889634f;
  __shared__ __attribute__((aligned(16))) uint4 Ks[NKP * KROW];
  __shared__ __attribute__((aligned(16))) uint4 Vs[NKP * CH];
  __shared__ int Rw[NKP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  int item = blockIdx.x;
  {   // XCD-contiguous ranges, as in the resident kernel
    const int G = (int)gridDim.x, xcd = item & 7, idx = item >> 3;
    item = xcd * (G >> 3) + min(xcd, G & 7) + idx;
  }
  const int seq = item / hsplit, part = item - seq * hsplit;
  const int hpw = p.heads / hsplit;
  const int h0 = part * hpw, h1 = h0 + hpw;

  if (tid < NKP) {
    int rg;
    Rw[tid] = tid < p.n ? (int)token_row<true>(p, seq, tid, &rg) : 0;
  }
  __syncthreads();
  const int qi = wave * 16 + l15;
  const bool q_ok = qi < p.n;
  const int64_t q_row = Rw[q_ok ? qi : 0];
  // staging role of this thread: key row tid / CH, chunk tid % CH of K and of V (NKP * CH == NT)
  const int skey = tid / CH, sc8 = tid - skey * CH;
  const bool s_ok = skey < p.n;
  const bf16_t* srow = p.qkv + (int64_t)Rw[s_ok ? skey : 0] * p.ld + sc8 * 8;
  const bf16_t* qrow = p.qkv + q_row * p.ld + g * 8;
  const int w = seq % p.nw;
  const int wy = w / p.nwx, wx = w - wy * p.nwx;
  const int kind = p.shift > 0 ? 2 * (wy == p.nw / p.nwx - 1 ? 1 : 0) + (wx == p.nwx - 1 ? 1 : 0) : 0;
  const float cs = p.scale * LOG2E;

  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  u32x4 kreg, vreg, qreg;
  float4 breg[4];                                  // this lane's (bias + mask) * log2e values of the next head's score block
  const float* bbase = p.bias_log2 + ((size_t)kind * p.heads * p.n + (q_ok ? qi : 0)) * NKP + g * 4;
  auto fetch = [&](int h) __attribute__((always_inline)) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    kreg = s_ok ? *reinterpret_cast<const u32x4*>(srow + p.dmodel + h * HD) : z;
    vreg = s_ok ? *reinterpret_cast<const u32x4*>(srow + 2 * p.dmodel + h * HD) : z;
    qreg = q_ok ? *reinterpret_cast<const u32x4*>(qrow + h * HD) : z;
    const float* bl = bbase + (size_t)h * p.n * NKP;
#pragma unroll
    for (int t = 0; t < 4; ++t) breg[t] = *reinterpret_cast<const float4*>(bl + t * 16);
  };
  fetch(h0);
  for (int h = h0; h < h1; ++h) {
    reinterpret_cast<u32x4*>(Ks)[k_slot<HD>(skey, sc8)] = kreg;
    reinterpret_cast<u32x4*>(Vs)[v_slot<HD>(skey, sc8)] = vreg;
    const bf16x8 qf = __builtin_bit_cast(bf16x8, qreg);
    const float4 bq[4] = {breg[0], breg[1], breg[2], breg[3]};
    __syncthreads();
    if (h + 1 < h1) fetch(h + 1);                 // in flight under this head's arithmetic
    if (wave * 16 < p.n) {
      f32x4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bf16x8 kf = __builtin_bit_cast(bf16x8, Ks[k_slot<HD>(t * 16 + l15, g)]);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, acc[t], 0, 0, 0);
      }
      float sc[16];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 b4 = bq[t];
        sc[t * 4 + 0] = fmaf(acc[t][0], cs, b4.x); sc[t * 4 + 1] = fmaf(acc[t][1], cs, b4.y);
        sc[t * 4 + 2] = fmaf(acc[t][2], cs, b4.z); sc[t * 4 + 3] = fmaf(acc[t][3], cs, b4.w);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (t * 16 + g * 4 + r >= p.n) sc[t * 4 + r] = -__builtin_inff();
      float mloc = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
#pragma unroll
      for (int i = 3; i < 15; i += 2) mloc = fmaxf(fmaxf(mloc, sc[i]), sc[i + 1]);
      mloc = fmaxf(mloc, sc[15]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      float psum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        sc[i] = __builtin_amdgcn_exp2f(sc[i] - mloc);
        psum += sc[i];
      }
      bf16x8 pf[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        uint4 pu;
        pu.x = pack_bf2(sc[8 * s2 + 0], sc[8 * s2 + 1]);
        pu.y = pack_bf2(sc[8 * s2 + 2], sc[8 * s2 + 3]);
        pu.z = pack_bf2(sc[8 * s2 + 4], sc[8 * s2 + 5]);
        pu.w = pack_bf2(sc[8 * s2 + 6], sc[8 * s2 + 7]);
        pf[s2] = __builtin_bit_cast(bf16x8, pu);
      }
      f32x4 o[DT];
#pragma unroll
      for (int i = 0; i < DT; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      typedef __attribute__((ext_vector_type(4))) short s16x4;
      typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int krow = g * 4 + (l15 >> 2);
          const int chunk = dt * 2 + ((l15 & 3) >> 1);
          const char* base = reinterpret_cast<const char*>(Vs) + (l15 & 1) * 8;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (lds_s16x4_ptr)(base + (size_t)v_slot<HD>(krow + (2 * s2) * 16, chunk) * 16));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (lds_s16x4_ptr)(base + (size_t)v_slot<HD>(krow + (2 * s2 + 1) * 16, chunk) * 16));
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 cat = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cat), pf[s2], o[dt], 0, 0, 0);
        }
      }
      float l_tot = psum + __shfl_xor(psum, 16, 64);
      l_tot += __shfl_xor(l_tot, 32, 64);
      if (q_ok) {
        const float inv = 1.f / l_tot;
        bf16_t* op = p.out + q_row * p.dmodel + h * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int d0 = dt * 16 + g * 4;
          *reinterpret_cast<uint2*>(op + d0) =
              make_uint2(pack_bf2(o[dt][0] * inv, o[dt][1] * inv), pack_bf2(o[dt][2] * inv, o[dt][3] * inv));
        }
      }
    }
    __syncthreads();                              // every wave is done with this head's K / V
  }
}

// ---------------------------------------------------------------------------------------------
// The same, with the (relative-position bias + shift mask) tile out of the window loop (round 6).  In attn_window_kernel every
// (window, head) loads its 16 bias values per lane again -- 16 KB per head and workgroup against 9.4 KB of Q, K and V: at
// Swin-B's stage 1 (65536 windows x 4 heads per layer) that is 4.3 GB of L2 reads per launch next to 0.6 GB of activations,
// and the launch runs at the L2's rate, not at HBM's.  The tile depends on the head and on the window's mask KIND only
// (interior / right edge / bottom edge / corner of the shifted frame, swin.py:243-285): a workgroup now owns HPW heads and
// a CHUNK of consecutive windows of ONE kind, holds its bias values in registers (16 x HPW floats per lane, loaded once) and
// walks the windows.  Same arithmetic per (window, head) as attn_window_kernel -- bit-identical results.
// Kinds of one image's nwy x nwx windows (shift > 0): 0 = wy < nwy-1 and wx < nwx-1, 1 = right column, 2 = bottom row,
// 3 = corner; shift == 0: every window is kind 0.
struct WinSched {
  int cnt[4];       // windows of each kind per image
  int first[5];     // first chunk of each kind in the chunk list (first[4] = number of chunks)
  int wpb;          // windows per chunk
};

template <int HD, int HPW>
__global__ void __launch_bounds__(256) attn_window_persist_kernel(const AttnArgs p, const int hsplit, const WinSched ws) {
  constexpr int NT = 256;
  constexpr int CH = HD / 8, KROW = CH + 1, DT = HD / 16;
  static_assert(HD == 32 && NT == 64 * CH, "window kernel: head dim 32, one K and one V chunk per thread");
  constexpr int NKP = 64;
  constexpr float LOG2E = 1.4426950408889634f;
  __shared__ __attribute__((aligned(16))) uint4 Ks[NKP * KROW];
  __shared__ __attribute__((aligned(16))) uint4 Vs[NKP * CH];
  __shared__ int Rw[2 * NKP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  int item = blockIdx.x;
  {   // XCD-contiguous ranges
    const int G = (int)gridDim.x, xcd = item & 7, idx = item >> 3;
    item = xcd * (G >> 3) + min(xcd, G & 7) + idx;
  }
  const int cg = item / hsplit, part = item - cg * hsplit;
  const int h0 = part * HPW;
  const int kind = cg >= ws.first[3] ? 3 : cg >= ws.first[2] ? 2 : cg >= ws.first[1] ? 1 : 0;
  const int cnt = ws.cnt[kind];
  const int j0 = (cg - ws.first[kind]) * ws.wpb;                 // first window of this chunk in the kind's list
  const int j1 = min(j0 + ws.wpb, p.batch * cnt);
  const int nwy = p.nw / p.nwx;

  const int qi = wave * 16 + l15;
  const bool q_ok = qi < p.n;
  const int skey = tid / CH, sc8 = tid - skey * CH;
  const bool s_ok = skey < p.n;
  const float cs = p.scale * LOG2E;

  // this lane's bias values of its HPW heads: once per workgroup
  float4 bias[HPW][4];
  {
    const float* bbase = p.bias_log2 + ((size_t)(p.shift > 0 ? kind : 0) * p.heads * p.n + (q_ok ? qi : 0)) * NKP + g * 4;
#pragma unroll
    for (int hl = 0; hl < HPW; ++hl)
#pragma unroll
      for (int t = 0; t < 4; ++t) bias[hl][t] = *reinterpret_cast<const float4*>(bbase + (size_t)(h0 + hl) * p.n * NKP + t * 16);
  }

  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  // window j of this kind -> (image, wy, wx) -> sequence index
  auto seq_of = [&](int j) __attribute__((always_inline)) -> int {
    const int b = j / cnt, r = j - b * cnt;
    int wy, wx;
    if (p.shift == 0 || kind == 0) {
      const int rowlen = p.shift == 0 ? p.nwx : p.nwx - 1;
      wy = r / rowlen; wx = r - wy * rowlen;
    } else if (kind == 1) { wy = r; wx = p.nwx - 1; }
    else if (kind == 2) { wy = nwy - 1; wx = r; }
    else { wy = nwy - 1; wx = p.nwx - 1; }
    return b * p.nw + wy * p.nwx + wx;
  };
  // The loads run ONE HEAD AHEAD across window boundaries as well: the index map of window j + 1 is computed while window j is
  // being multiplied (second buffer of Rw, published by window j's barriers), and the first head of window j + 1 is requested
  // under the arithmetic of window j's last head.
  u32x4 kreg, vreg, qreg;
  const bf16_t *srow, *qrow;
  int64_t q_row;
  auto bind = [&](const int* rw) __attribute__((always_inline)) {
    q_row = rw[q_ok ? qi : 0];
    srow = p.qkv + (int64_t)rw[s_ok ? skey : 0] * p.ld + sc8 * 8;
    qrow = p.qkv + q_row * p.ld + g * 8;
  };
  auto fetch = [&](int h) __attribute__((always_inline)) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    kreg = s_ok ? *reinterpret_cast<const u32x4*>(srow + p.dmodel + h * HD) : z;
    vreg = s_ok ? *reinterpret_cast<const u32x4*>(srow + 2 * p.dmodel + h * HD) : z;
    qreg = q_ok ? *reinterpret_cast<const u32x4*>(qrow + h * HD) : z;
  };
  if (j0 < j1) {
    if (tid < NKP) {
      int rg;
      Rw[tid] = tid < p.n ? (int)token_row<true>(p, seq_of(j0), tid, &rg) : 0;
    }
    __syncthreads();
    bind(Rw);
    fetch(h0);
  }
  int cur = 0;
  for (int j = j0; j < j1; ++j, cur ^= 1) {
    const int64_t q_row_cur = q_row;            // (bind() below moves q_row to the next window while this one is being stored)
    if (j + 1 < j1 && tid < NKP) {               // next window's index map: read behind this window's barriers
      int rg;
      Rw[(cur ^ 1) * NKP + tid] = tid < p.n ? (int)token_row<true>(p, seq_of(j + 1), tid, &rg) : 0;
    }
#pragma unroll
    for (int hl = 0; hl < HPW; ++hl) {
      const int h = h0 + hl;
      reinterpret_cast<u32x4*>(Ks)[k_slot<HD>(skey, sc8)] = kreg;
      reinterpret_cast<u32x4*>(Vs)[v_slot<HD>(skey, sc8)] = vreg;
      const bf16x8 qf = __builtin_bit_cast(bf16x8, qreg);
      __syncthreads();
      if (hl + 1 < HPW) fetch(h + 1);                 // in flight under this head's arithmetic
      else if (j + 1 < j1) {                          // ... the next window's first head under this window's last
        bind(Rw + (cur ^ 1) * NKP);
        fetch(h0);
      }
      if (wave * 16 < p.n) {
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          const bf16x8 kf = __builtin_bit_cast(bf16x8, Ks[k_slot<HD>(t * 16 + l15, g)]);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, acc[t], 0, 0, 0);
        }
        float sc[16];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float4 b4 = bias[hl][t];
          sc[t * 4 + 0] = fmaf(acc[t][0], cs, b4.x); sc[t * 4 + 1] = fmaf(acc[t][1], cs, b4.y);
          sc[t * 4 + 2] = fmaf(acc[t][2], cs, b4.z); sc[t * 4 + 3] = fmaf(acc[t][3], cs, b4.w);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
            if (t * 16 + g * 4 + rr >= p.n) sc[t * 4 + rr] = -__builtin_inff();
        float mloc = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) mloc = fmaxf(fmaxf(mloc, sc[i]), sc[i + 1]);
        mloc = fmaxf(mloc, sc[15]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          sc[i] = __builtin_amdgcn_exp2f(sc[i] - mloc);
          psum += sc[i];
        }
        bf16x8 pf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          uint4 pu;
          pu.x = pack_bf2(sc[8 * s2 + 0], sc[8 * s2 + 1]);
          pu.y = pack_bf2(sc[8 * s2 + 2], sc[8 * s2 + 3]);
          pu.z = pack_bf2(sc[8 * s2 + 4], sc[8 * s2 + 5]);
          pu.w = pack_bf2(sc[8 * s2 + 6], sc[8 * s2 + 7]);
          pf[s2] = __builtin_bit_cast(bf16x8, pu);
        }
        f32x4 o[DT];
#pragma unroll
        for (int i = 0; i < DT; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const int krow = g * 4 + (l15 >> 2);
            const int chunk = dt * 2 + ((l15 & 3) >> 1);
            const char* base = reinterpret_cast<const char*>(Vs) + (l15 & 1) * 8;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (lds_s16x4_ptr)(base + (size_t)v_slot<HD>(krow + (2 * s2) * 16, chunk) * 16));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (lds_s16x4_ptr)(base + (size_t)v_slot<HD>(krow + (2 * s2 + 1) * 16, chunk) * 16));
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            const s16x8 cat = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cat), pf[s2], o[dt], 0, 0, 0);
          }
        }
        float l_tot = psum + __shfl_xor(psum, 16, 64);
        l_tot += __shfl_xor(l_tot, 32, 64);
        if (q_ok) {
          const float inv = 1.f / l_tot;
          bf16_t* op = p.out + q_row_cur * p.dmodel + h * HD;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const int d0 = dt * 16 + g * 4;
            *reinterpret_cast<uint2*>(op + d0) =
                make_uint2(pack_bf2(o[dt][0] * inv, o[dt][1] * inv), pack_bf2(o[dt][2] * inv, o[dt][3] * inv));
          }
        }
      }
      __syncthreads();                              // every wave is done with this head's K / V (and with Rw on the last head)
    }
  }
}

template <int HD, int NW, int TQ>
static int launch_attn_stream(const AttnArgs& a, int64_t nseq, hipStream_t st) {
  const int nkp = (a.n + 63) / 64 * 64;
  const int krow = HD == 64 ? 8 : HD / 8 + 1;
  const size_t lds = (size_t)nkp * krow * 16 + (size_t)nkp * (HD / 8) * 16;
  static tfimm_once_t attr_done;
  static int cus = 256;
  if (attr_done.need()) {
    TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)attn_stream_kernel<HD, NW, TQ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
    attr_done.mark();
  }
  const int64_t items = nseq * a.heads;
  const int grid = (int)(items < cus ? items : cus);
  TFIMM_LAUNCH((attn_stream_kernel<HD, NW, TQ>), dim3((unsigned)grid), dim3(NW * 64), lds, st, a, nkp, (int)items);
  return 0;
}

template <int HD>
static size_t attn_resident_lds(int n, bool swin, bool tiles) {
  const int nkp = (n + 63) / 64 * 64;
  const int krow = HD == 64 ? 8 : HD / 8 + 1;
  size_t b = (size_t)nkp * krow * 16 + (size_t)nkp * (HD / 8) * 16;
  if (swin) b += (tiles ? 0 : (size_t)n * nkp * sizeof(float)) + 2 * (size_t)n * sizeof(int);
  return b;
}

template <int HD, bool SWIN, int NW, int TQ>
static int launch_attn_resident(const AttnArgs& a, int64_t nseq, hipStream_t st) {
  const int nkp = (a.n + 63) / 64 * 64;
  const size_t lds = attn_resident_lds<HD>(a.n, SWIN, a.bias_log2 != nullptr);
  static tfimm_once_t attr_done;
  if (attr_done.need()) {
    TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)attn_resident_kernel<HD, SWIN, NW, TQ>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done.mark();
  }
  TFIMM_LAUNCH((attn_resident_kernel<HD, SWIN, NW, TQ>), dim3((unsigned)(nseq * a.heads)), dim3(NW * 64), lds, st, a, nkp);
  return 0;
}

// waves x query tiles per wave that cover ceil(n / 16) tiles: <= 4 -> 4x1, <= 8 -> 8x1, <= 16 -> 8x2
template <int HD, bool SWIN>
static int launch_attn_resident_any(const AttnArgs& a, int64_t nseq, hipStream_t st) {
  const int tiles = (a.n + 15) / 16;
  if (tiles <= 4) return launch_attn_resident<HD, SWIN, 4, 1>(a, nseq, st);
  if (tiles <= 8) return launch_attn_resident<HD, SWIN, 8, 1>(a, nseq, st);
  return launch_attn_resident<HD, SWIN, 8, 2>(a, nseq, st);
}

}  // namespace

extern "C" int tfimm_hip_attention(const tfimm_attn_desc* dp, void* stream) {
  if (!dp) TFIMM_FAIL(TFIMM_EINVAL, "attention: null descriptor");
  const tfimm_attn_desc& d = *dp;
  if (!d.qkv || !d.out) TFIMM_FAIL(TFIMM_EINVAL, "attention: null pointer");
  if (d.batch <= 0 || d.n_tokens <= 0 || d.heads <= 0 || d.hd <= 0)
    TFIMM_FAIL(TFIMM_EINVAL, "attention: bad shape");
  if (d.hd > 128) TFIMM_FAIL(TFIMM_EUNSUP, "attention: head dim %d > 128 not built", d.hd);
  if (d.hd > 64 && d.window > 0) TFIMM_FAIL(TFIMM_EUNSUP, "attention: windows with head dim %d > 64 not built", d.hd);
  AttnArgs a;
  a.qkv = (const bf16_t*)d.qkv; a.out = (bf16_t*)d.out; a.rel_bias = d.rel_bias;
  a.bias_log2 = d.window > 0 ? d.bias_log2 : nullptr;
  a.batch = d.batch; a.n_tokens = d.n_tokens; a.heads = d.heads; a.hd = d.hd; a.scale = d.scale;
  a.window = d.window; a.shift = d.shift; a.res_h = d.res_h; a.res_w = d.res_w;
  a.dmodel = d.heads * d.hd; a.ld = 3 * a.dmodel;
  static const int attn_dbg = getenv("TFIMM_ATTN_DBG") ? atoi(getenv("TFIMM_ATTN_DBG")) : 0;
  a.dbg = attn_dbg;
  static const int attn_xcd = getenv("TFIMM_ATTN_XCD") ? atoi(getenv("TFIMM_ATTN_XCD")) : 1;
  a.xcd_map = attn_xcd;
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
  if (d.hd > 64) {
    // head dims 65..128 (vit_huge_patch14: 80): the streaming kernel with three / four MFMA k-steps over the head dimension
    // and six / eight output tiles per wave; the columns beyond hd are zero in K, Q and V
    if (d.hd <= 96) TFIMM_LAUNCH((attn_kernel<96, false>), grid, block, 0, st, a);
    else TFIMM_LAUNCH((attn_kernel<128, false>), grid, block, 0, st, a);
    return 0;
  }
  {
    // K + V^T (+ Swin bias tile) of one (sequence, head) resident in LDS (<= 80 KiB: two workgroups per CU)
    const bool tl = d.window > 0 && d.bias_log2 != nullptr;
    const size_t lds = d.hd <= 32 ? attn_resident_lds<32>(a.n, d.window > 0, tl) : attn_resident_lds<64>(a.n, d.window > 0, tl);
    static int use_resident = -1;
    if (use_resident < 0) {
      const char* e = getenv("TFIMM_ATTN_NO_RESIDENT");
      use_resident = (e && e[0] == '1') ? 0 : 1;
    }
    // global attention at head dim 64 with 9..16 query tiles (n = 129..256: ViT / DeiT / CaiT at 224) and enough items
    // to keep every CU's workgroup busy for several rounds: the persistent kernel (measured 5-9 % faster on ViT-B;
    // no gain at head dim 32 or with few heads, which stay on the kernel below)
    static const bool no_stream = getenv("TFIMM_ATTN_NO_STREAM") != nullptr;
    if (use_resident && !no_stream && d.window == 0 && a.vec && a.n > 128 && a.n <= 256 && d.hd == 64 &&
        nseq * d.heads >= 2048 && nseq * d.heads <= 0x7fffffffLL)
      return launch_attn_stream<64, 16, 1>(a, nseq, st);
    // Swin windows (head dim 32, <= 64 tokens, pre-combined bias tiles): a workgroup owns a window and a range of its heads
    static const bool no_window = getenv("TFIMM_ATTN_NO_WINDOW") != nullptr;
    if (use_resident && !no_window && tl && a.vec && d.hd == 32 && a.n <= 64 && nseq * d.heads <= 0x7fffffffLL) {
      // heads per workgroup: split a window's heads over workgroups until there are ~16 workgroups per CU (measured on Swin-B:
      // 2.00 ms of attention at a 1024-workgroup threshold, 1.93 at 4096)
      // (round 6) bias tiles out of the window loop: a workgroup owns HPW heads and a chunk of windows of one mask kind
      // (attn_window_persist_kernel); TFIMM_ATTN_WIN_V1=1 keeps the one-window-per-workgroup kernel (A/B)
      static const bool win_v1 = getenv("TFIMM_ATTN_WIN_V1") != nullptr;
      if (!win_v1) {
        const int hpw = (d.heads % 2 == 0) ? 2 : 1;
        const int hsplit2 = d.heads / hpw;
        const int nwy = a.nw / a.nwx;
        WinSched ws;
        if (d.shift > 0) {
          ws.cnt[0] = (nwy - 1) * (a.nwx - 1); ws.cnt[1] = nwy - 1; ws.cnt[2] = a.nwx - 1; ws.cnt[3] = 1;
        } else {
          ws.cnt[0] = a.nw; ws.cnt[1] = ws.cnt[2] = ws.cnt[3] = 0;
        }
        // windows per chunk: as many as leave ~4096 workgroups (16 per CU), at most 16
        static const int64_t win_wgs2 = getenv("TFIMM_ATTN_WIN_WGS") ? atoi(getenv("TFIMM_ATTN_WIN_WGS")) : 4096;
        int64_t wpb = nseq * hsplit2 / win_wgs2;
        wpb = wpb < 1 ? 1 : (wpb > 16 ? 16 : wpb);
        ws.wpb = (int)wpb;
        int64_t total = 0;
        for (int k = 0; k < 4; ++k) {
          ws.first[k] = (int)total;
          total += ((int64_t)d.batch * ws.cnt[k] + wpb - 1) / wpb;
        }
        ws.first[4] = (int)total;
        if (total * hsplit2 <= 0x7fffffffLL) {
          if (hpw == 2) TFIMM_LAUNCH((attn_window_persist_kernel<32, 2>), dim3((unsigned)(total * hsplit2)), dim3(256), 0, st, a, hsplit2, ws);
          else TFIMM_LAUNCH((attn_window_persist_kernel<32, 1>), dim3((unsigned)(total * hsplit2)), dim3(256), 0, st, a, hsplit2, ws);
          return 0;
        }
      }
      int hsplit = 1;
      static const int64_t win_wgs = getenv("TFIMM_ATTN_WIN_WGS") ? atoi(getenv("TFIMM_ATTN_WIN_WGS")) : 4096;
      while (hsplit < d.heads && nseq * hsplit < win_wgs && d.heads % (hsplit * 2) == 0) hsplit *= 2;
      TFIMM_LAUNCH((attn_window_kernel<32>), dim3((unsigned)(nseq * hsplit)), dim3(256), 0, st, a, hsplit);
      return 0;
    }
    if (use_resident && lds <= 80 * 1024 && a.n <= 256 && nseq * d.heads <= 0x7fffffffLL) {
      if (d.window > 0)
        return d.hd <= 32 ? launch_attn_resident_any<32, true>(a, nseq, st) : launch_attn_resident_any<64, true>(a, nseq, st);
      return d.hd <= 32 ? launch_attn_resident_any<32, false>(a, nseq, st) : launch_attn_resident_any<64, false>(a, nseq, st);
    }
  }
  if (d.window > 0) {
    if (d.hd <= 32) TFIMM_LAUNCH((attn_kernel<32, true>), grid, block, 0, st, a);
    else TFIMM_LAUNCH((attn_kernel<64, true>), grid, block, 0, st, a);
  } else {
    if (d.hd <= 32) TFIMM_LAUNCH((attn_kernel<32, false>), grid, block, 0, st, a);
    else TFIMM_LAUNCH((attn_kernel<64, false>), grid, block, 0, st, a);
  }
  return 0;
}
