// tfimm_hip_expand_dwconv: the front half of an inverted-residual (MBConv) block as ONE kernel (gfx950)
//     conv_pw (1x1, Cin -> C) + bn1 + act  ->  conv_dw (k x k depthwise, stride s) + bn2 + act  [-> SE squeeze sums]
// (efficientnet_blocks.py:438-445; InvertedResidual.call up to the squeeze-excite).  As two launches the expanded tensor --
// 6x the block's input, the largest tensor of the network -- is written by the GEMM and read back by the depthwise kernel, and
// the GEMM with K = 24 / 32 is one k-step followed by an activation over C outputs per pixel that its epilogue pays for with the
// MFMA pipe idle.  Here the expanded activations exist only in LDS:
//
//   * a workgroup (8 waves) owns an OTH x OTW tile of OUTPUT pixels of one image and walks the expanded channels in chunks of
//     32.  The input pixels the tile needs (its halo, IH x IW) are fetched once into LDS (64 bytes per pixel, K padded to 32,
//     16-byte chunks XOR-swizzled for the fragment reads) and serve every chunk;
//   * phase 1 (per chunk): E[pixel][32 channels] = act(W1^T . x + b1) for every halo pixel by v_mfma_f32_32x32x16_bf16 with
//     the weights as the A operand (host-packed fragments, 2 k-steps), so a lane ends up with 16 channels of ONE pixel; pixels
//     outside the image are zeroed (TF pads the EXPANDED tensor with zeros, not the expansion of zero pixels), rounded to bf16
//     exactly as the two-launch path rounds the tensor it stores, and written to LDS at a pitch of 72 bytes per pixel
//     (conflict-free 8-byte writes, no address arithmetic on the read side);
//   * phase 2 (per chunk): a thread owns a channel PAIR and one output column and marches down its rows; every halo row costs K
//     4-byte LDS reads at immediate offsets and feeds the K output rows it contributes to with packed FMAs (taps as float2 in
//     registers, read per chunk from an LDS copy).  bias + activation, 4-byte stores (a wave writes 64-byte runs of 4 pixels),
//     squeeze sums of the stored values through LDS atomics and one global atomic per (workgroup, channel).
//   Both phases are VALU-bound (the activation: two quarter-rate transcendentals per value for swish); two workgroups share a
//   CU so one's global loads / stores sit under the other's arithmetic.  HBM traffic: input halo once + output once.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct MbArgs {
  const bf16_t* x;
  const uint4* w1;
  const float* b1;
  const float* wdw;
  const float* b2;
  bf16_t* y;
  tfimm_sq_t* sums;
  int H, W, Cin, C, Cpad, pad_t, pad_l, OH, OW, tiles_x, act1, act2;
  int img_h, img_w;   // STEM: the padded 4-channel image x points at; H, W are then the convolution's output (= depthwise input) size
  int dbg;      // TFIMM_MB_DBG ablation bits: 1 no output stores, 2 no phase-2 activation, 4 no phase-1 activation
};

template <int K, int S, int OTH, int OTW, bool STEM = false>
struct MbGeom {
  static constexpr int NT = 512, NSLOT = 32;
  static constexpr int RG = NSLOT / OTW, RPT = OTH / RG;
  static_assert(NSLOT % OTW == 0 && OTH % RG == 0, "tile does not map onto 32 column slots");
  static constexpr int IH = (OTH - 1) * S + K, IW = (OTW - 1) * S + K, NPX = IH * IW, NBLK = (NPX + 31) / 32;
  static constexpr int JB = (NBLK + 7) / 8;
  static constexpr int NR = (RPT - 1) * S + K;
  static constexpr int XP = 64, EP = 72;
  // STEM: the expansion is a 3 x 3 / stride 2 convolution of the padded 4-channel image; LDS holds the image region the halo
  // needs ((2 IH + 1) x (2 IW + 1) pixels of 8 bytes) instead of the halo's own pixels
  static constexpr int XIH = 2 * IH + 1, XIW = 2 * IW + 1;
  static constexpr int X_BYTES = STEM ? (XIH * XIW * 8 + 15) / 16 * 16 : NPX * XP, E_BYTES = NPX * EP;
  static constexpr int KS = STEM ? 3 : 2;                      // MFMA k-steps of the expansion (K padded to 48 / 32)
  static constexpr int WD_FLOATS = (K * K + 1) * 32;
  static constexpr int WDN = (WD_FLOATS + NT - 1) / NT;         // tap / bias values each thread stages per chunk
  // PF: a chunk's weights are requested one chunk ahead (see the kernel).  Stride-1 tiles only: measured on EfficientNet-B4 the
  // stride-1 launches gain 7 % (518 -> 480 us), the stride-2 ones -- whose expansion phase is 4.4x their depthwise phase -- LOSE
  // 7 - 11 % with it (897 -> 998, 440 -> 473 us; tools/mb_diag.py, one box), so they keep loading at the top of the chunk
  static constexpr bool PF = S == 1;
  static constexpr int B1_FLOATS = PF ? 512 : 0;                // expansion bias of EVERY chunk, staged once (Cpad <= 512: host check)
  static constexpr int WD_LDS = PF ? WDN * NT : WD_FLOATS;      // floats of the tap region (PF: every thread writes its WDN values, no branch)
  static constexpr int LDS = X_BYTES + E_BYTES + WD_LDS * 4 + 64 * 8 + B1_FLOATS * 4;
  static_assert(LDS <= 80 * 1024, "two workgroups per CU");
};

// ACT >= 0: both activations are that TFIMM_ACT_* (its parameters fold into the instructions: no scalar registers, no class
// branches); ACT < 0: read act1 / act2 from the arguments
template <int K, int S, int OTH, int OTW, int ACT, bool STEM = false>
__global__ __launch_bounds__(512, 4) void expand_dw_kernel(MbArgs p) {
  using G = MbGeom<K, S, OTH, OTW, STEM>;
  extern __shared__ __attribute__((aligned(16))) unsigned char mb_smem[];
  unsigned char* Xs = mb_smem;
  unsigned char* Es = mb_smem + G::X_BYTES;
  float* Wd = reinterpret_cast<float*>(Es + G::E_BYTES);
  tfimm_sq_t* lsum = reinterpret_cast<tfimm_sq_t*>(Wd + G::WD_LDS);         // 2 x 32 fixed-point squeeze sums
  float* B1s = reinterpret_cast<float*>(lsum + 64);                          // b1[Cpad]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.y;
  const int ty = blockIdx.x / p.tiles_x, tx = blockIdx.x - ty * p.tiles_x;
  const int oy0 = ty * OTH, ox0 = tx * OTW;
  const int gy0 = oy0 * S - p.pad_t, gx0 = ox0 * S - p.pad_l;

  // ---- phase 0: the input halo, once per tile ------------------------------------------------------------------------
  if (STEM) {
    // image rows 2 (gy0 + iy) + ky, columns 2 (gx0 + ix) + kx: the region starts at (2 gy0, 2 gx0)
    const bf16_t* xb = p.x + (size_t)b * p.img_h * p.img_w * 4;
    for (int i = tid; i < G::XIH * G::XIW; i += G::NT) {
      const int y = i / G::XIW, xx = i - y * G::XIW;
      const int gy = 2 * gy0 + y, gx = 2 * gx0 + xx;
      uint2 v = make_uint2(0u, 0u);
      if ((unsigned)gy < (unsigned)p.img_h && (unsigned)gx < (unsigned)p.img_w)
        v = *reinterpret_cast<const uint2*>(xb + ((size_t)gy * p.img_w + gx) * 4);
      *reinterpret_cast<uint2*>(Xs + i * 8) = v;
    }
    if (tid < 64) lsum[tid] = 0;
    if (G::PF && tid < p.Cpad) B1s[tid] = p.b1[tid];
  } else {
    const int nch = p.Cin >> 3;
    const bf16_t* xb = p.x + (size_t)b * p.H * p.W * p.Cin;
#pragma unroll
    for (int i0 = 0; i0 < G::NPX * 4; i0 += G::NT) {
      const int i = i0 + tid;
      if (i < G::NPX * 4) {
        const int px = i >> 2, c = i & 3;
        const int iy = px / G::IW, ix = px - iy * G::IW;
        const int gy = gy0 + iy, gx = gx0 + ix;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (c < nch && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
          v = *reinterpret_cast<const uint4*>(xb + ((size_t)gy * p.W + gx) * p.Cin + c * 8);
        *reinterpret_cast<uint4*>(Xs + px * G::XP + ((c ^ ((px >> 2) & 3)) << 4)) = v;
      }
    }
    if (tid < 64) lsum[tid] = 0;
    if (G::PF && tid < p.Cpad) B1s[tid] = p.b1[tid];
  }
  __syncthreads();          // the halo is read by other threads than the ones that staged it
  // which of this lane's halo pixels (one per block it expands) lie inside the image
  uint32_t vbits = 0;
#pragma unroll
  for (int j = 0; j < G::JB; ++j) {
    const int px = (wave + 8 * j) * 32 + l31;
    const int iy = px / G::IW, ix = px - iy * G::IW;
    const bool ok = px < G::NPX && (unsigned)(gy0 + iy) < (unsigned)p.H && (unsigned)(gx0 + ix) < (unsigned)p.W;
    vbits |= (ok ? 1u : 0u) << j;
  }
  const ActParams a1 = make_act(ACT >= 0 ? ACT : p.act1), a2 = make_act(ACT >= 0 ? ACT : p.act2);
  const int nchunks = p.Cpad >> 5;

  // Weights of a chunk -- the expansion fragments (KS x 16 bytes per lane) and the taps + bias of the depthwise layer (WDN
  // floats per thread) -- are requested ONE CHUNK AHEAD, at the start of the depthwise phase: in front of that phase's output
  // stores in program order, so that (VMEM retiring in issue order) they do not wait for the stores' acknowledgements, and a
  // whole phase early.  Loaded at the top of their own chunk they cost 7 - 11 % of the launch (tools/mb_diag.py: 890 -> 811,
  // 514 -> 478, 440 -> 390 us with the loads removed).  The expansion bias of every chunk sits in LDS since phase 0.
#ifndef TFIMM_MB_ABLATE
#define TFIMM_MB_ABLATE 0      // probe builds (tools/mb_diag.py): 1 = every chunk multiplies with chunk 0's weights (timing only)
#endif
  uint4 af_n[G::KS];
  float wd_n[G::WDN];
  auto prefetch = [&](int c) __attribute__((always_inline)) {
    const int cw = (TFIMM_MB_ABLATE & 1) ? 0 : min(c, nchunks - 1);       // (past the last chunk: the last one again, never used)
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) af_n[ks] = p.w1[(size_t)(cw * G::KS + ks) * 64 + lane];
#pragma unroll
    for (int u = 0; u < G::WDN; ++u) {
      const int i = min(u * G::NT + tid, G::WD_FLOATS - 1);
      const int t = i >> 5, ch = cw * 32 + (i & 31);
      const float* src = t < K * K ? p.wdw + (size_t)t * p.Cpad + ch : p.b2 + ch;
      wd_n[u] = *src;
    }
  };
  if (G::PF) prefetch(0);
  // As many stores that write nothing (offsets beyond a zero-record descriptor) as a depthwise phase issues stand behind the first
  // prefetch: on both paths to the top of a chunk -- from the prologue and around the loop -- the prefetched weights are then
  // followed by the same number of younger VMEM operations, and the compiler's wait for them is `vmcnt(RPT_FULL)`: the previous
  // chunk's output stores stay in flight.  (It falls back to `vmcnt(0)` when the paths differ; that is also why a last chunk
  // with the half-chunk mapping is peeled out of the loop below instead of being a second branch inside it.)
  const __amdgpu_buffer_rsrc_t rs_none = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, 0, 0x00020000);
  constexpr int RPT_FULL = OTH / ((G::NT / 16) / OTW);
  auto empty_stores = [&](int n0, int n1) __attribute__((always_inline)) {
#pragma unroll
    for (int e = n0; e < n1; ++e) __builtin_amdgcn_raw_buffer_store_b32((unsigned)e, rs_none, (int)(0x7fffff00u + 64u * e), 0, 0);
  };
  if (G::PF) empty_stores(0, RPT_FULL);
  constexpr bool HALF_OK = (G::NT / 8) % OTW == 0 && OTH % ((G::NT / 8) / OTW) == 0;
  // a last chunk of at most 16 channels runs with the half-chunk mapping of the depthwise phase: peeled, behind the loop
  const int n_loop = (HALF_OK && p.C - (nchunks - 1) * 32 <= 16) ? nchunks - 1 : nchunks;
  auto chunk = [&](const int cc, auto half_tag) __attribute__((always_inline)) {
    // squeeze sums of the previous chunk: one global atomic per channel, then re-arm that half of the buffer
    if (p.sums && cc > 0 && tid < 32) {
      const int h = ((cc - 1) & 1) * 32 + tid, ch = (cc - 1) * 32 + tid;
      if (ch < p.C) sq_add(p.sums + (size_t)b * p.C + ch, lsum[h]);
      lsum[h] = 0;
    }
    // ---- phase 1: expand + activation into LDS ------------------------------------------------------------------------
    bf16x8 af[G::KS];
    f32x4 bq[4];
    if (G::PF) {
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) af[ks] = __builtin_bit_cast(bf16x8, af_n[ks]);
#pragma unroll
      for (int u = 0; u < G::WDN; ++u) Wd[u * G::NT + tid] = wd_n[u];      // (read in phase 2, behind the barrier below; the region holds WDN NT floats)
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4*>(B1s + cc * 32 + q * 8 + hi * 4);
    } else {
      const int ccw = (TFIMM_MB_ABLATE & 1) ? 0 : cc;
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) af[ks] = __builtin_bit_cast(bf16x8, p.w1[(size_t)(ccw * G::KS + ks) * 64 + lane]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4*>(p.b1 + ccw * 32 + q * 8 + hi * 4);
        bq[q] = f32x4{b4.x, b4.y, b4.z, b4.w};
      }
#pragma unroll
      for (int i0 = 0; i0 < G::WD_FLOATS; i0 += G::NT) {
        const int i = i0 + tid;
        if (i < G::WD_FLOATS) {
          const int t = i >> 5, ch = ccw * 32 + (i & 31);
          Wd[i] = t < K * K ? p.wdw[(size_t)t * p.Cpad + ch] : p.b2[ch];
        }
      }
    }
    const int nq = min(4, (p.C - cc * 32 + 7) >> 3);
#pragma unroll
    for (int j = 0; j < G::JB; ++j) {
      const int blk = wave + 8 * j;
      if (blk < G::NBLK) {
        const int px = blk * 32 + l31;
        const int pxr = min(px, G::NPX - 1);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        if (STEM) {
          // k = 4 tap + channel: this lane's half of k-step ks holds taps 4 ks + 2 hi and + 1 (ky = tap / 3, kx = tap % 3;
          // taps 9..11 do not exist), 8 bytes each out of the image region
          const int py = pxr / G::IW, pxx = pxr - py * G::IW;
          const unsigned char* xa = Xs + ((2 * py) * G::XIW + 2 * pxx) * 8;
#pragma unroll
          for (int ks = 0; ks < G::KS; ++ks) {
            uint2 t[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              t[e] = make_uint2(0u, 0u);
              if (ks < 2) {                                                  // taps 0..7: both halves exist
                const int tap = ks * 4 + hi * 2 + e;
                t[e] = *reinterpret_cast<const uint2*>(xa + ((tap / 3) * G::XIW + tap % 3) * 8);
              } else if (e == 0) {                                           // k-step 2: tap 8 (hi == 0) only
                const uint2 v = *reinterpret_cast<const uint2*>(xa + (2 * G::XIW + 2) * 8);
                t[e] = hi ? make_uint2(0u, 0u) : v;
              }
            }
            const uint4 u = make_uint4(t[0].x, t[0].y, t[1].x, t[1].y);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], __builtin_bit_cast(bf16x8, u), acc, 0, 0, 0);
          }
        } else {
          const int sw = (pxr >> 2) & 3;
          const unsigned char* xa = Xs + pxr * G::XP;
          const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(xa + ((hi ^ sw) << 4));
          const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(xa + (((2 + hi) ^ sw) << 4));
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], x0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[G::KS > 1 ? 1 : 0], x1, acc, 0, 0, 0);
        }
        const uint32_t keep = ((vbits >> j) & 1u) ? 0xffffffffu : 0u;
        unsigned char* ea = Es + px * G::EP + hi * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (2 * h < nq) {            // wave-uniform: a last chunk of 8 / 16 / 24 channels skips the activation of its padding
            tfimm_f32x2 v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int q = 2 * h + (e >> 1), r = (e & 1) * 2;
              v[e] = tfimm_f32x2{acc[q * 4 + r] + bq[q][r], acc[q * 4 + r + 1] + bq[q][r + 1]};
            }
            // p.dbg is a RUN-TIME zero in product builds (the host passes 0 unless built with -DTFIMM_PROBE_HOOKS), on purpose:
            // with the three switches of this kernel folded away at compile time hipcc allocates 14-48 registers more than the
            // 128 the launch bounds give it and spills them (expand_dw 3.8 -> 4.9 ms per EfficientNet-B4 forward, round 4);
            // scheduling barriers in their place do not bring the old allocation back, the branches do
            if (!(p.dbg & 4)) act8p(v, a1);
            if (px < G::NPX) {
#pragma unroll
              for (int qq = 0; qq < 2; ++qq) {
                const uint32_t u0 = pack_bf2(v[qq * 2][0], v[qq * 2][1]) & keep;
                const uint32_t u1 = pack_bf2(v[qq * 2 + 1][0], v[qq * 2 + 1][1]) & keep;
                *reinterpret_cast<uint2*>(ea + (2 * h + qq) * 16) = make_uint2(u0, u1);
              }
            }
          }
        }
      }
    }
    __syncthreads();
    if (G::PF) prefetch(cc + 1);
    // ---- phase 2: depthwise taps out of LDS ------------------------------------------------------------------------------
    // A last chunk of at most 16 channels (48 = 32 + 16, 144 = 4 x 32 + 16) would leave half of the 16 channel-pair columns
    // idle: it runs with 8 channel pairs x 64 pixel slots instead, every thread marching half as many rows.
    auto phase2 = [&](auto hc) __attribute__((always_inline)) {
      constexpr bool HALF = decltype(hc)::value;
      constexpr int NCPL = HALF ? 8 : 16;                       // channel pairs handled side by side
      constexpr int RGL = (G::NT / NCPL) / OTW;                 // row groups
      constexpr int RPTL = OTH / RGL;                           // output rows per thread
      constexpr int NRL = (RPTL - 1) * S + K;
      const int cpl = tid % NCPL, slotl = tid / NCPL;
      const int coll = slotl % OTW, rgl = slotl / OTW;
      const uint32_t* epl = reinterpret_cast<const uint32_t*>(Es + ((rgl * RPTL * S) * G::IW + coll * S) * G::EP + cpl * 4);
      const tfimm_f32x2* wll = reinterpret_cast<const tfimm_f32x2*>(Wd) + cpl;
      const int oxl = ox0 + coll, oybl = oy0 + rgl * RPTL;
      tfimm_f32x2 w[K * K];
#pragma unroll
      for (int t = 0; t < K * K; ++t) w[t] = wll[t * 16];
      const tfimm_f32x2 bias2 = wll[K * K * 16];
      tfimm_f32x2 acc[RPTL];
#pragma unroll
      for (int r = 0; r < RPTL; ++r) acc[r] = bias2;
#pragma unroll
      for (int j = 0; j < NRL; ++j) {
        tfimm_f32x2 v[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const uint32_t raw = epl[(j * G::IW + kx) * (G::EP / 4)];
          v[kx] = tfimm_f32x2{__uint_as_float(raw << 16), __uint_as_float(raw & 0xffff0000u)};
        }
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          if (j - ky >= 0 && (j - ky) % S == 0 && (j - ky) / S < RPTL) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
              acc[(j - ky) / S] = __builtin_elementwise_fma(v[kx], w[ky * K + kx], acc[(j - ky) / S]);
          }
        }
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // keeps hipcc from hoisting every row's LDS reads to the top
      }
      const int c0 = cc * 32 + cpl * 2;
      const bool cok = c0 < p.C && oxl < p.OW;
      tfimm_f32x2 tot = {0.f, 0.f};
      // Output stores through a buffer descriptor over THIS image (OH OW C elements, < 2 GiB: checked by the host): one 32-bit
      // byte offset per thread, stepped by a uniform row pitch -- a channel pair / column outside the tensor gets an offset
      // beyond the descriptor, rows >= OH run past its end by themselves, and the store is dropped.  The 64-bit row pointers
      // of the first version did not fit into the 128 registers of four waves per SIMD next to the depthwise accumulators:
      // hipcc spilled them and reloaded each in front of its store, and a scratch reload is a VMEM load -- `s_waitcnt vmcnt(0)`
      // in front of every store waited for the write acknowledgement of the store before it (gfx9 retires VMEM in issue
      // order), 12 round trips per thread and chunk.  No branches around the stores either, so they stay countable.
      const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
          p.y + (size_t)b * p.OH * p.OW * p.C, 0, (int)((unsigned)p.OH * (unsigned)p.OW * (unsigned)p.C * 2u), 0x00020000);
      const unsigned row_pitch = (unsigned)p.OW * (unsigned)p.C * 2u;
      const unsigned yoff0 = cok ? (unsigned)(((oybl * p.OW + oxl) * p.C + c0) * 2) : 0x7fffff00u;
#pragma unroll
      for (int r0 = 0; r0 < RPTL; r0 += 4) {
        tfimm_f32x2 v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (r0 + e < RPTL) ? acc[(r0 + e < RPTL) ? r0 + e : 0] : tfimm_f32x2{0.f, 0.f};
        if (!(p.dbg & 2)) act8p(v, a2);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (r0 + e < RPTL) {
            const uint32_t pk = pack_bf2(v[e][0], v[e][1]);
            if (!(TFIMM_PROBE(p.dbg) & 1)) __builtin_amdgcn_raw_buffer_store_b32(pk, rs_y, (int)(yoff0 + (unsigned)(r0 + e) * row_pitch), 0, 0);
            // the squeeze sees the stored (bf16-rounded) activations, as in tfimm_hip_dwconv
            const uint32_t seen = (cok && oybl + r0 + e < p.OH) ? pk : 0u;
            tot += tfimm_f32x2{__uint_as_float(seen << 16), __uint_as_float(seen & 0xffff0000u)};
          }
        }
      }
      if (p.sums) {
        sq_add(&lsum[(cc & 1) * 32 + cpl * 2], sq_from_float(tot[0]));
        sq_add(&lsum[(cc & 1) * 32 + cpl * 2 + 1], sq_from_float(tot[1]));
      }
    };
    phase2(half_tag);
    __syncthreads();
  };
  for (int cc = 0; cc < n_loop; ++cc) chunk(cc, std::false_type{});
  if (HALF_OK && n_loop < nchunks) chunk(n_loop, std::integral_constant<bool, HALF_OK>{});
  if (p.sums && tid < 32) {
    const int cl = nchunks - 1;
    const int ch = cl * 32 + tid;
    if (ch < p.C) sq_add(p.sums + (size_t)b * p.C + ch, lsum[(cl & 1) * 32 + tid]);
  }
}

template <int K, int S, int OTH, int OTW, int ACT, bool STEM = false>
int launch_expand_dw_act(const MbArgs& a0, int B, hipStream_t st) {
  using G = MbGeom<K, S, OTH, OTW, STEM>;
  MbArgs a = a0;
  a.tiles_x = (a.OW + OTW - 1) / OTW;
  const int tiles_y = (a.OH + OTH - 1) / OTH;
  auto fn = expand_dw_kernel<K, S, OTH, OTW, ACT, STEM>;
  static tfimm_once_t ready;
  if (ready.need()) {
    TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
    ready.mark();
  }
  TFIMM_LAUNCH(fn, dim3((unsigned)(a.tiles_x * tiles_y), (unsigned)B), dim3(G::NT), (size_t)G::LDS, st, a);
  return 0;
}

template <int K, int S, int OTH, int OTW>
int launch_expand_dw(const MbArgs& a, int B, hipStream_t st) {
  if (a.act1 == a.act2 && a.act1 == TFIMM_ACT_SWISH) return launch_expand_dw_act<K, S, OTH, OTW, TFIMM_ACT_SWISH>(a, B, st);
  if (a.act1 == a.act2 && a.act1 == TFIMM_ACT_RELU6) return launch_expand_dw_act<K, S, OTH, OTW, TFIMM_ACT_RELU6>(a, B, st);
  return launch_expand_dw_act<K, S, OTH, OTW, -1>(a, B, st);
}

}  // namespace

extern "C" int tfimm_hip_expand_dwconv(const tfimm_expand_dw_desc* d, void* stream) {
  if (!d) TFIMM_FAIL(TFIMM_EINVAL, "expand_dwconv: null descriptor");
  if (!d->x || !d->w1 || !d->b1 || !d->wdw || !d->b2 || !d->y) TFIMM_FAIL(TFIMM_EINVAL, "expand_dwconv: null pointer");
  if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->OH <= 0 || d->OW <= 0)
    TFIMM_FAIL(TFIMM_EINVAL, "expand_dwconv: bad shape");
  if (d->stem) {
    // x is the zero-bordered 4-channel image; the expansion is the 3 x 3 / stride 2 stem convolution without padding of its own
    if (d->Cin != 4 || d->img_h <= 0 || d->img_w <= 0 || 2 * (d->H - 1) + 3 > d->img_h || 2 * (d->W - 1) + 3 > d->img_w)
      TFIMM_FAIL(TFIMM_EINVAL, "expand_dwconv: stem needs Cin = 4 and an image of at least %d x %d pixels", 2 * d->H + 1, 2 * d->W + 1);
    if (d->k != 3 || d->stride != 1) TFIMM_FAIL(TFIMM_EUNSUP, "expand_dwconv: stem flavour is built for a 3 x 3 / stride 1 depthwise layer");
  } else if (d->Cin <= 0 || (d->Cin & 7) || d->Cin > 32)
    TFIMM_FAIL(TFIMM_EUNSUP, "expand_dwconv: Cin=%d (multiples of 8 up to 32)", d->Cin);
  if ((d->C & 1) || d->Cpad != (d->C + 31) / 32 * 32)
    TFIMM_FAIL(TFIMM_EINVAL, "expand_dwconv: C=%d must be even and Cpad=%d its multiple-of-32 ceiling", d->C, d->Cpad);
  if (d->B > 65535) TFIMM_FAIL(TFIMM_EUNSUP, "expand_dwconv: batch %d > 65535", d->B);
  if (d->Cpad > 512 && d->stride == 1)
    TFIMM_FAIL(TFIMM_EUNSUP, "expand_dwconv: C=%d (at most 512 expanded channels at stride 1: their bias is staged in LDS)", d->C);
  if ((int64_t)d->OH * d->OW * d->C * 2 > 0x7fffff00LL) TFIMM_FAIL(TFIMM_EUNSUP, "expand_dwconv: one image of the output exceeds 2 GiB");
  if ((((uintptr_t)d->x | (uintptr_t)d->w1 | (uintptr_t)d->b1) & 15) || ((uintptr_t)d->y & 3))
    TFIMM_FAIL(TFIMM_EINVAL, "expand_dwconv: x / w1 / b1 must be 16-byte aligned, y 4-byte aligned");
  if (d->pad_t < 0 || d->pad_l < 0 || d->pad_t >= d->k || d->pad_l >= d->k)
    TFIMM_FAIL(TFIMM_EINVAL, "expand_dwconv: padding (%d, %d) for k=%d", d->pad_t, d->pad_l, d->k);
  if ((d->OH - 1) * d->stride - d->pad_t >= d->H || (d->OW - 1) * d->stride - d->pad_l >= d->W)
    TFIMM_FAIL(TFIMM_EINVAL, "expand_dwconv: output %dx%d does not fit input %dx%d", d->OH, d->OW, d->H, d->W);
  MbArgs a;
  a.x = (const bf16_t*)d->x; a.w1 = (const uint4*)d->w1; a.b1 = d->b1; a.wdw = d->wdw; a.b2 = d->b2;
  a.y = (bf16_t*)d->y; a.sums = reinterpret_cast<tfimm_sq_t*>(d->sum_out);
  a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.C = d->C; a.Cpad = d->Cpad; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
  a.OH = d->OH; a.OW = d->OW; a.tiles_x = 0; a.act1 = d->act1; a.act2 = d->act2;
  a.img_h = d->img_h; a.img_w = d->img_w;
#ifdef TFIMM_PROBE_HOOKS
  static const int dbg = getenv("TFIMM_MB_DBG") ? atoi(getenv("TFIMM_MB_DBG")) : 0;     // ablation switches: probe builds only
  a.dbg = dbg;
#else
  a.dbg = 0;            // (a run-time zero the kernel still branches on: see the comment there)
#endif
  hipStream_t st = (hipStream_t)stream;
  if (d->stem) {
    if (a.act1 == a.act2 && a.act1 == TFIMM_ACT_SWISH) return launch_expand_dw_act<3, 1, 12, 32, TFIMM_ACT_SWISH, true>(a, d->B, st);
    if (a.act1 == a.act2 && a.act1 == TFIMM_ACT_RELU6) return launch_expand_dw_act<3, 1, 12, 32, TFIMM_ACT_RELU6, true>(a, d->B, st);
    return launch_expand_dw_act<3, 1, 12, 32, -1, true>(a, d->B, st);
  }
  if (d->k == 3 && d->stride == 1) return launch_expand_dw<3, 1, 12, 32>(a, d->B, st);
  // (8 x 16 outputs per tile at stride 2: 6 x 16 balances the expansion phase over the waves better -- 14 pixel blocks instead of
  //  18 on 8 waves -- and is 12 % slower, 4 x 16 33 %: the halo overhead decides; tools/mb_diag.py)
  if (d->k == 3 && d->stride == 2) return launch_expand_dw<3, 2, 8, 16>(a, d->B, st);
  if (d->k == 5 && d->stride == 2) return launch_expand_dw<5, 2, 6, 16>(a, d->B, st);
  TFIMM_FAIL(TFIMM_EUNSUP, "expand_dwconv: k=%d stride=%d (3 or 5, stride 1 or 2)", d->k, d->stride);
}
