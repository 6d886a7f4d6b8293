// CaiT attention operators (gfx950): talking-heads attention, class attention, token-axis concat.
// See include/tfimm_hip.h for the reference call sites (tfimm/architectures/cait.py).
//
// tfimm_hip_talking_heads_attention
// ---------------------------------
//   attn = softmax_j( proj_l( scale * q k^T ) );  out = proj_w(attn) v      (cait.py:239-256)
// proj_l / proj_w are Dense(H -> H) layers over the HEAD axis, applied at every (query, key)
// position, so the H score maps of one position must meet before the softmax and again before the
// P.V product.  The kernel keeps them together in registers:
//   * one workgroup = 4 waves = 64 query rows of one image, ALL heads; a wave owns 16 queries and
//     walks the keys in tiles of 16.  For a tile it computes S_h^T = K_h . Q_h^T for every head h
//     (v_mfma_f32_16x16x32_bf16, K as the "a" operand): a lane then holds, for its query (lane & 15)
//     and its 4 keys (4 * (lane >> 4) + r), the scores of all H heads in H x 4 registers -- the
//     head mixing is H x H scalar-weight FMAs per register, weights in SGPRs.
//   * the softmax runs over MIXED logits, so a one-pass online softmax would need the P.V partial
//     sums of every (mixed head, value head) pair.  Instead the keys are walked twice: pass 1
//     gathers max / sum of every (mixed head, query) (QK^T + first mixing only), pass 2 recomputes
//     the logits, normalises, applies the second mixing and multiplies with V.  QK^T is cheap next
//     to the mixing (H^2 VALU FMAs per position against 2 * hd MFMA MACs).
//   * P (accumulator layout: query = lane & 15, keys 4g .. 4g+3) is exactly the "b" operand of
//     v_mfma_f32_16x16x16_bf16; the matching V^T "a" operand comes out of the row-major V block
//     with one transposing LDS read (ds_read_b64_tr_b16): O^T = V^T . P^T needs no shuffle and the
//     staging no transposition.  Output accumulators are 4 * hd/16 registers per value head, so value heads are
//     processed in groups of <= 8 (H = 16: pass 2 runs twice).
//   * K block and V block [32 keys][D] of the image are staged per workgroup with plain 16-byte
//     copies (8 threads per row: no index divisions); heads sit at a stride padded to whole 32-wide
//     k-steps with zeroed pads, so fragment reads need no masking.  The workgroup's Q rows live in
//     LDS when they fit, else Q fragments come from L1/L2.
//   Measured on the first version (B=256, N=196, H=4): VALU 88 % busy, 68 % of the LDS cycles bank
//   conflicts -- index divisions and 2-byte transposing stores in the staging, masks on every
//   fragment; all three are gone in this layout.
#include "common.h"

#include <cstdlib>

namespace {

typedef __attribute__((ext_vector_type(4))) short bf16x4_t;

// The two H x H head-mixing layers travel BY VALUE in the kernel argument segment and are copied
// to LDS once per workgroup; the mixing loops read them from there (broadcast ds_read, re-read for
// every key tile).  Both alternatives cost the second wave per SIMD: through global pointers hipcc
// hoists all 2 H^2 weights into VGPRs for the whole key loop, as SGPRs (kernarg scalar loads) they
// overflow the scalar file and every use becomes a v_readlane from a spill register.
struct ThaWeights {
  float wl[256], bl[16];   // proj_l kernel [H_in][H_out] (row stride H), bias
  float ww[256], bw[16];   // proj_w
};

struct ThaArgs {
  const bf16_t* qkv;
  bf16_t* out;
  int batch, n, heads;
  float scale;
  int ld, dmodel;
  int qchunks;
  int kstr;       // LDS row stride (elements) of the K / Q blocks (heads padded to whole 32-wide k-steps)
  int vstr;       // LDS row stride of the V block
  int q_in_lds;
  const float* wdev;   // DEVICE copy of the mixing layers [wl H*H | bl H | ww H*H | bw H] (tfimm_tha_desc.proj_dev), or null
};

constexpr int THA_KB = 32;        // keys staged per step

#ifdef TFIMM_THA_DBG
// Probe build only (tools/probes/build_tha_dbg.sh; profiles/NOTES_r04.md section 1): integrity checks of everything the
// workgroup keeps in LDS, 16 words per workgroup:
//   0 HW_ID  1 LDS_ALLOC  2 GPR_ALLOC  3 XCC_ID  4 K/V chunk staged by ANOTHER wave != its source, right behind the barrier
//   5 own source chunk loaded a second time != first load  6 Wm  7 zero pads of Ks  8 Qs vs global  9 own staged chunk changed
//   while the block was consumed, or poisoned columns no fragment read touches (Ks / Qs columns >= H*HP, Vs columns >= D) changed
//   10 St vs the registers that wrote it
//   11 first bad LDS byte offset + 1   12/13 s_memrealtime at entry / exit (low words)   14/15 the first 8 bytes of the bad 16-byte slot
__device__ unsigned tha_dbg_buf[16384 * 16];
__device__ float tha_dbg_st[1024 * 512];      // the softmax statistics (St) of workgroups 0 .. 1023 as pass 2 reads them
__device__ unsigned tha_dbg_ck[1024 * 4];
__device__ float tha_dbg_run[256 * 2048];     // workgroups 0 .. 255: every lane's (m_run, l_run) per mixed head when the key loop of pass 1 ends     // wrap-around word sums of everything staged: K in pass 1, K in pass 2, V in pass 2
__device__ __forceinline__ uint4 tha_vload(const bf16_t* p) {
  typedef __attribute__((ext_vector_type(4))) unsigned u4v;
  const u4v v = *reinterpret_cast<const volatile u4v*>(p);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
#define THA_DBG_ONLY(...) __VA_ARGS__
#else
#define THA_DBG_ONLY(...)
#endif

template <int H, int HG, int DT, bool QLDS>
__global__ void __launch_bounds__(256) tha_kernel(const ThaArgs p, const ThaWeights w) {
  constexpr int HD = DT * 16;
  constexpr int KS = (HD + 31) / 32;
  constexpr int HP = KS * 32;       // head stride of the K / Q blocks in LDS: head dim padded to whole k-steps
  constexpr int CPH = HD / 8;       // 16-byte chunks per head row
  constexpr float LOG2E = 1.4426950408889634f;
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
  extern __shared__ __attribute__((aligned(16))) char smem_tha[];
  const int D = p.dmodel;
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem_tha);                     // [THA_KB][kstr], heads at stride HP
  bf16_t* Vs = Ks + THA_KB * p.kstr;                                     // [THA_KB][vstr], row-major V
  // Everything the packed fp32 arithmetic below takes from LDS is stored TWICE, as a (v, v) pair, and used as a whole pair.
  // With one copy hipcc loads two neighbours with one ds_read2_b32 and picks the second with an operand select
  // (v_pk_fma_f32 ... op_sel:[0,1,0]) -- and on gfx950 a packed fp32 operation whose LOW result takes the HIGH half of a
  // source returns the product of the wrong half in lanes 48..63 whenever a wave of another workgroup issues MFMAs on the same
  // SIMD (tools/tha_coresident_probe.py ldsret reproduces it on a six-instruction kernel; profiles/NOTES_r04.md section 1).
  // tools/isa_lint.py keeps such instructions out of every product kernel.
  float* St = reinterpret_cast<float*>(Vs + THA_KB * p.vstr);            // [4 waves][H][16][4]: (max, max, 1 / sum, 1 / sum)
  tfimm_f32x2* Wm = reinterpret_cast<tfimm_f32x2*>(St + 4 * H * 16 * 4); // wl [H][H], bl [H], ww [H][H], bw [H], every value as a pair
  bf16_t* Qs = reinterpret_cast<bf16_t*>(Wm + 2 * (H * H + H));          // [64][kstr] (q_in_lds)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int qc = blockIdx.x % p.qchunks;
  const int img = blockIdx.x / p.qchunks;
  const int64_t row0 = (int64_t)img * p.n;
  const int q = qc * 64 + wave * 16 + l15;
  const bool q_ok = q < p.n;
  const int CH = D / 8;             // 16-byte chunks per token row of one of q / k / v
#ifdef TFIMM_THA_DBG
  unsigned* const dbg = tha_dbg_buf + (size_t)(blockIdx.x < 16384 ? blockIdx.x : 16383) * 16;
  const int lds_total = (int)(reinterpret_cast<char*>(Qs) - smem_tha) + (QLDS ? 64 * p.kstr * 2 : 0);
  for (int i = tid; i < lds_total / 4; i += 256) reinterpret_cast<unsigned*>(smem_tha)[i] = 0xDEADBEEFu;
  if (tid == 0) {
    dbg[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    dbg[1] = __builtin_amdgcn_s_getreg((31 << 11) | 6);
    dbg[2] = __builtin_amdgcn_s_getreg((31 << 11) | 5);
    dbg[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    for (int i = 4; i < 12; ++i) dbg[i] = 0;
    if (blockIdx.x < 1024) for (int i = 0; i < 4; ++i) tha_dbg_ck[blockIdx.x * 4 + i] = 0;
    dbg[12] = (unsigned)__builtin_amdgcn_s_memrealtime();
  }
  __syncthreads();
  auto dbg_bad = [&](int what, const void* where) {
    atomicAdd(&dbg[what], 1u);
    if (atomicCAS(&dbg[11], 0u, (unsigned)(reinterpret_cast<const char*>(where) - smem_tha) + 1u) == 0u) {
      const unsigned* w4 = reinterpret_cast<const unsigned*>((size_t)where & ~(size_t)15);   // the 16 bytes around it, as they are now
      dbg[14] = w4[0]; dbg[15] = w4[1];
    }
  };
  constexpr int DBG_NC = (H * HD / 8 + 7) / 8;
  uint4 dbg_k[DBG_NC], dbg_v[DBG_NC];
  bool dbg_have = false, dbg_with_v = false;
#endif

  if (p.wdev) {
    // from device memory, in Wm's own order (plans pass the device copy: no 2 KiB by-value struct per launch)
    for (int id = tid; id < 2 * (H * H + H); id += 256) {
      const float v = p.wdev[id];
      const float e = (id >= H * H && id < H * H + H) ? v * LOG2E : v;
      Wm[id] = tfimm_f32x2{e, e};
    }
  } else {
    for (int id = tid; id < H * H; id += 256) {
      Wm[id] = tfimm_f32x2{w.wl[id], w.wl[id]};
      Wm[H * H + H + id] = tfimm_f32x2{w.ww[id], w.ww[id]};
    }
    if (tid < H) {
      Wm[H * H + tid] = tfimm_f32x2{w.bl[tid] * LOG2E, w.bl[tid] * LOG2E};
      Wm[2 * H * H + H + tid] = tfimm_f32x2{w.bw[tid], w.bw[tid]};
    }
  }
  // staging map (no divisions): 8 threads per row, thread (row = tid >> 3) walks chunks (tid & 7) + 8 j;
  // chunk c of a token row belongs to head c / CPH and lands at column head * HP + (c % CPH) * 8
  const int srow = tid >> 3, sc0 = tid & 7;
  if (HP != HD) {   // the pad columns of every head stay zero for the whole kernel
    for (int id = tid; id < THA_KB * H; id += 256) {
      const int r = id / H, h = id - r * H;
#pragma unroll
      for (int e = HD; e < HP; e += 8) *reinterpret_cast<uint4*>(&Ks[r * p.kstr + h * HP + e]) = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  if (QLDS) {
    for (int rr = srow; rr < 64; rr += 32) {
      const int t = qc * 64 + rr;
      for (int c = sc0; c < CH; c += 8) {
        uint4 u = make_uint4(0u, 0u, 0u, 0u);
        if (t < p.n) u = *reinterpret_cast<const uint4*>(p.qkv + (row0 + t) * p.ld + c * 8);
        const int h = c / CPH;
        *reinterpret_cast<uint4*>(&Qs[rr * p.kstr + h * HP + (c - h * CPH) * 8]) = u;
      }
      if (HP != HD) {
        for (int h = sc0; h < H; h += 8)
#pragma unroll
          for (int e = HD; e < HP; e += 8) *reinterpret_cast<uint4*>(&Qs[rr * p.kstr + h * HP + e]) = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  }
  // (these LDS offsets are made opaque once per key tile: fragments, softmax statistics and mixing
  // weights are then re-read from LDS for every tile instead of being hoisted and kept live across the
  // key loop.  Offsets, not pointers: an opaque POINTER loses its address space and turns every read
  // into a flat load)
  int qs_off = (wave * 16 + l15) * p.kstr;
  int st_off = (wave * H * 16 + l15) * 4;
  int wm_off = 0;
  // Q fragment of head h, k-step ks for this lane's query: d = ks*32 + g*8 .. +8 (zero beyond HD)
  auto q_frag = [&](int h, int ks) -> bf16x8 {
    const int d0 = ks * 32 + g * 8;
    if (QLDS) return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&Qs[qs_off + h * HP + d0]));
    // from L1/L2: lanes whose k-group lies beyond the head read a valid address and are masked to zero
    const int dc = d0 < HD ? d0 : 0;
    const unsigned keep = (d0 < HD && q_ok) ? 0xffffffffu : 0u;
    uint4 u = *reinterpret_cast<const uint4*>(p.qkv + (row0 + (q_ok ? q : 0)) * p.ld + h * HD + dc);
    u.x &= keep; u.y &= keep; u.z &= keep; u.w &= keep;
    return __builtin_bit_cast(bf16x8, u);
  };

  auto stage = [&](int kb, bool with_v) {
#ifdef TFIMM_THA_DBG
    if (dbg_have) {   // the block this thread staged last time, now that every fragment read of it is over
      int j = 0;
      for (int c = sc0; c < CH; c += 8, ++j) {
        const int h = c / CPH;
        const uint4* kq = reinterpret_cast<const uint4*>(&Ks[srow * p.kstr + h * HP + (c - h * CPH) * 8]);
        const uint4 a = *kq;
        if (a.x != dbg_k[j].x || a.y != dbg_k[j].y || a.z != dbg_k[j].z || a.w != dbg_k[j].w) dbg_bad(9, kq);
        if (dbg_with_v) {
          const uint4* vq = reinterpret_cast<const uint4*>(&Vs[srow * p.vstr + c * 8]);
          const uint4 b = *vq;
          if (b.x != dbg_v[j].x || b.y != dbg_v[j].y || b.z != dbg_v[j].z || b.w != dbg_v[j].w) dbg_bad(9, vq);
        }
      }
    }
#endif
    __syncthreads();   // previous block fully consumed (and Qs / Wm written, first time)
    const int t = kb + srow;
    const bf16_t* kp = p.qkv + (row0 + (t < p.n ? t : 0)) * p.ld + D;
    for (int c = sc0; c < CH; c += 8) {
      uint4 ku = make_uint4(0u, 0u, 0u, 0u), vu = make_uint4(0u, 0u, 0u, 0u);
      if (t < p.n) {
        ku = *reinterpret_cast<const uint4*>(kp + c * 8);
        if (with_v) vu = *reinterpret_cast<const uint4*>(kp + D + c * 8);
      }
      const int h = c / CPH;
      *reinterpret_cast<uint4*>(&Ks[srow * p.kstr + h * HP + (c - h * CPH) * 8]) = ku;
      if (with_v) *reinterpret_cast<uint4*>(&Vs[srow * p.vstr + c * 8]) = vu;
      THA_DBG_ONLY(dbg_k[(c - sc0) >> 3] = ku; dbg_v[(c - sc0) >> 3] = vu;
                   if (blockIdx.x < 1024) {
                     atomicAdd(&tha_dbg_ck[blockIdx.x * 4 + (with_v ? 1 : 0)], ku.x + ku.y + ku.z + ku.w);
                     if (with_v) atomicAdd(&tha_dbg_ck[blockIdx.x * 4 + 2], vu.x + vu.y + vu.z + vu.w);
                   })
    }
    __syncthreads();
#ifdef TFIMM_THA_DBG
    dbg_have = true; dbg_with_v = with_v;
    {
      // (4) what ANOTHER wave staged (the thread 64 further on), right behind the barrier, against a fresh load of its source:
      //     a barrier that lets this wave through early, or LDS writes that are not visible yet, show here
      // (5) this thread's own source chunks loaded a second time against the registers of the first load
      const int orow = ((tid + 64) & 255) >> 3;
      const int ot = kb + orow;
      const bf16_t* okp = p.qkv + (row0 + (ot < p.n ? ot : 0)) * p.ld + D;
      int j = 0;
      for (int c = sc0; c < CH; c += 8, ++j) {
        const int h = c / CPH;
        uint4 eku = make_uint4(0u, 0u, 0u, 0u), evu = eku, rku = eku, rvu = eku;
        if (ot < p.n) {
          eku = tha_vload(okp + c * 8);
          if (with_v) evu = tha_vload(okp + D + c * 8);
        }
        if (t < p.n) {
          rku = tha_vload(kp + c * 8);
          if (with_v) rvu = tha_vload(kp + D + c * 8);
        }
        const uint4* kq = reinterpret_cast<const uint4*>(&Ks[orow * p.kstr + h * HP + (c - h * CPH) * 8]);
        const uint4 a = *kq;
        if (a.x != eku.x || a.y != eku.y || a.z != eku.z || a.w != eku.w) dbg_bad(4, kq);
        if (rku.x != dbg_k[j].x || rku.y != dbg_k[j].y || rku.z != dbg_k[j].z || rku.w != dbg_k[j].w) dbg_bad(5, kq);
        if (with_v) {
          const uint4* vq = reinterpret_cast<const uint4*>(&Vs[orow * p.vstr + c * 8]);
          const uint4 b = *vq;
          if (b.x != evu.x || b.y != evu.y || b.z != evu.z || b.w != evu.w) dbg_bad(4, vq);
          if (rvu.x != dbg_v[j].x || rvu.y != dbg_v[j].y || rvu.z != dbg_v[j].z || rvu.w != dbg_v[j].w) dbg_bad(5, vq);
        }
      }
    }
#endif
  };

  // mixed logits (log2 units) of key tile t of the staged block, all H mixed heads
  tfimm_f32x2 cs2 = {p.scale * LOG2E, p.scale * LOG2E};
  asm volatile("" : "+v"(cs2));      // a real register pair (no operand select on a scalar)
  // mixed[hp] = (lo, hi): the logits of keys 4g, 4g+1 and of keys 4g+2, 4g+3
  auto logits = [&](int kb, int t, tfimm_f32x2 (*mixed)[2]) __attribute__((always_inline)) {
    asm volatile("" : "+v"(qs_off), "+v"(st_off), "+v"(wm_off));
    const tfimm_f32x2* wm = Wm + wm_off;
#pragma unroll
    for (int hp = 0; hp < H; ++hp) {
      const tfimm_f32x2 b = wm[H * H + hp];
      mixed[hp][0] = b;
      mixed[hp][1] = b;
    }
    // fragments of head h + 1 are requested before head h is multiplied (one head of lookahead hides
    // the LDS latency); the scheduling barrier keeps hipcc from hoisting ALL heads' fragments to the
    // top, which costs 16 H registers and with them the second wave per SIMD
    bf16x8 kf[2][KS], qf[2][KS];
    auto frags = [&](int h, int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        kf[buf][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&Ks[(t * 16 + l15) * p.kstr + h * HP + ks * 32 + g * 8]));
        qf[buf][ks] = q_frag(h, ks);
      }
    };
    frags(0, 0);
#pragma unroll
    for (int h = 0; h < H; ++h) {
      if (h + 1 < H) frags(h + 1, (h + 1) & 1);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[h & 1][ks], qf[h & 1][ks], acc, 0, 0, 0);
      const tfimm_f32x2 a_lo = tfimm_f32x2{acc[0], acc[1]} * cs2, a_hi = tfimm_f32x2{acc[2], acc[3]} * cs2;
#pragma unroll
      for (int hp = 0; hp < H; ++hp) {
        const tfimm_f32x2 w2 = wm[h * H + hp];
        mixed[hp][0] = __builtin_elementwise_fma(w2, a_lo, mixed[hp][0]);
        mixed[hp][1] = __builtin_elementwise_fma(w2, a_hi, mixed[hp][1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kb + THA_KB > p.n) {   // last block: keys beyond the sequence take no part in the softmax
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (kb + t * 16 + g * 4 + r >= p.n) {
#pragma unroll
          for (int hp = 0; hp < H; ++hp) mixed[hp][r >> 1][r & 1] = -1e30f;
        }
    }
  };

  THA_DBG_ONLY(float dbg_st[H][2];)
  // ---- pass 1: max / sum of every (mixed head, query), over this lane's keys first
  {
    float m_run[H], l_run[H];
#pragma unroll
    for (int hp = 0; hp < H; ++hp) { m_run[hp] = -1e30f; l_run[hp] = 0.f; }
    for (int kb = 0; kb < p.n; kb += THA_KB) {
      stage(kb, false);
#pragma unroll 1
      for (int t = 0; t < THA_KB / 16; ++t) {
        tfimm_f32x2 mixed[H][2];
        logits(kb, t, mixed);
#pragma unroll
        for (int hp = 0; hp < H; ++hp) {
          const f32x4 v = {mixed[hp][0][0], mixed[hp][0][1], mixed[hp][1][0], mixed[hp][1][1]};
          const float m_new = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), m_run[hp]);
          l_run[hp] = l_run[hp] * __builtin_amdgcn_exp2f(m_run[hp] - m_new) + __builtin_amdgcn_exp2f(v[0] - m_new) +
                      __builtin_amdgcn_exp2f(v[1] - m_new) + __builtin_amdgcn_exp2f(v[2] - m_new) +
                      __builtin_amdgcn_exp2f(v[3] - m_new);
          m_run[hp] = m_new;
        }
      }
    }
    THA_DBG_ONLY(if (blockIdx.x < 256) for (int hp = 0; hp < H; ++hp) {
                   tha_dbg_run[blockIdx.x * 2048 + (tid * H + hp) * 2] = m_run[hp];
                   tha_dbg_run[blockIdx.x * 2048 + (tid * H + hp) * 2 + 1] = l_run[hp]; })
    // combine the four key groups (g) of a query, publish (max, 1 / sum)
#pragma unroll
    for (int hp = 0; hp < H; ++hp) {
      float m = fmaxf(m_run[hp], __shfl_xor(m_run[hp], 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float l = l_run[hp] * __builtin_amdgcn_exp2f(m_run[hp] - m);
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      // (Leftover of a REFUTED hypothesis, harmless: it was suspected that hipcc sinks the last add into the `if (g == 0)` block
      // and that its ds_bpermute_b32 then executes under the narrowed EXEC mask.  The victim kernel built for it showed 0 wrong
      // lanes and forcing the wait changed nothing (profiles/NOTES_r04.md section 1, "dead ends").  The cause of the round-3
      // co-residency defect was the packed-fp32 op_sel form -- see the St / Wm comment further up and DESIGN.md section 3,
      // rule 1.  The empty statement only pins the add in front of the branch.)
      asm volatile("" : "+v"(m), "+v"(l));
      if (g == 0) {
        const float il = 1.f / l;
        *reinterpret_cast<f32x4*>(&St[((wave * H + hp) * 16 + l15) * 4]) = f32x4{m, m, il, il};
      }
      THA_DBG_ONLY(dbg_st[hp][0] = m; dbg_st[hp][1] = 1.f / l;)
    }
  }

  // ---- pass 2: probabilities, second mixing, P.V -- value heads in groups of HG
  for (int hq0 = 0; hq0 < H; hq0 += HG) {
    f32x4 o[HG][DT];
#pragma unroll
    for (int i = 0; i < HG; ++i)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) o[i][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < p.n; kb += THA_KB) {
      stage(kb, true);
#pragma unroll 1
      for (int t = 0; t < THA_KB / 16; ++t) {
        tfimm_f32x2 pr[H][2];
        logits(kb, t, pr);
#pragma unroll
        for (int hp = 0; hp < H; ++hp) {
          const tfimm_f32x2 m2 = *reinterpret_cast<const tfimm_f32x2*>(&St[st_off + hp * 64]);
          const tfimm_f32x2 il2 = *reinterpret_cast<const tfimm_f32x2*>(&St[st_off + hp * 64 + 2]);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const tfimm_f32x2 d = pr[hp][u] - m2;
            pr[hp][u] = tfimm_f32x2{__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])} * il2;
          }
        }
        // V^T fragments by transposing LDS reads: lane (l15, g) supplies the address of the 8-byte piece
        // (key 4g + l15/4, d 4*(l15%4)..+3) and receives keys 4g..4g+3 at d = l15 -- the MFMA "a" operand
        const char* vbase = reinterpret_cast<const char*>(Vs) + ((t * 16 + g * 4 + (l15 >> 2)) * p.vstr + (l15 & 3) * 4) * 2;
#pragma unroll
        for (int i = 0; i < HG; ++i) {
          const int hq = hq0 + i;
          const tfimm_f32x2 b2 = Wm[wm_off + 2 * H * H + H + hq];
          tfimm_f32x2 a_lo = b2, a_hi = b2;
#pragma unroll
          for (int hp = 0; hp < H; ++hp) {
            const tfimm_f32x2 w2 = Wm[wm_off + H * H + H + hp * H + hq];
            a_lo = __builtin_elementwise_fma(w2, pr[hp][0], a_lo);
            a_hi = __builtin_elementwise_fma(w2, pr[hp][1], a_hi);
          }
          // keys beyond the sequence carry the bias b, but their V rows are staged as zeros
          const uint2 pu = make_uint2(pack_bf2(a_lo[0], a_lo[1]), pack_bf2(a_hi[0], a_hi[1]));
          const bf16x4_t pf = __builtin_bit_cast(bf16x4_t, pu);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const s16x4 vf = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(vbase + (hq * HD + dt * 16) * 2));
            o[i][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(bf16x4_t, vf), pf, o[i][dt], 0, 0, 0);
          }
        }
      }
    }
    // lane (q, g) holds d = dt*16 + g*4 + r of each value head
    if (q_ok) {
      bf16_t* op = p.out + (row0 + q) * D;
#pragma unroll
      for (int i = 0; i < HG; ++i)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const f32x4 v = o[i][dt];
          *reinterpret_cast<uint2*>(op + (hq0 + i) * HD + dt * 16 + g * 4) =
              make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        }
    }
  }
#ifdef TFIMM_THA_DBG
  __syncthreads();
  // mixing layers
  if (p.wdev)
    for (int id = tid; id < 2 * (H * H + H); id += 256) {
      const float v = p.wdev[id];
      const float e = (id >= H * H && id < H * H + H) ? v * LOG2E : v;
      if (__float_as_uint(Wm[id][0]) != __float_as_uint(e) || __float_as_uint(Wm[id][1]) != __float_as_uint(e)) dbg_bad(6, &Wm[id]);
    }
  // zero pads of the K block, and the columns no fragment read touches (still poisoned)
  for (int id = tid; id < THA_KB * H; id += 256) {
    const int r = id / H, h = id - r * H;
    for (int e = HD; e < HP; ++e)
      if (Ks[r * p.kstr + h * HP + e] != 0) dbg_bad(7, &Ks[r * p.kstr + h * HP + e]);
  }
  for (int id = tid; id < THA_KB * (p.kstr - H * HP) / 2; id += 256) {
    const int per = (p.kstr - H * HP) / 2, r = id / per, e = id - r * per;
    const unsigned* q = reinterpret_cast<const unsigned*>(&Ks[r * p.kstr + H * HP]) + e;
    if (*q != 0xDEADBEEFu) dbg_bad(9, q);
  }
  for (int id = tid; id < THA_KB * (p.vstr - D) / 2; id += 256) {
    const int per = (p.vstr - D) / 2, r = id / per, e = id - r * per;
    const unsigned* q = reinterpret_cast<const unsigned*>(&Vs[r * p.vstr + D]) + e;
    if (*q != 0xDEADBEEFu) dbg_bad(9, q);
  }
  if (QLDS) {
    for (int id = tid; id < 64 * (p.kstr - H * HP) / 2; id += 256) {
      const int per = (p.kstr - H * HP) / 2, r = id / per, e = id - r * per;
      const unsigned* q = reinterpret_cast<const unsigned*>(&Qs[r * p.kstr + H * HP]) + e;
      if (*q != 0xDEADBEEFu) dbg_bad(9, q);
    }
    for (int rr = srow; rr < 64; rr += 32) {
      const int t = qc * 64 + rr;
      for (int c = sc0; c < CH; c += 8) {
        uint4 u = make_uint4(0u, 0u, 0u, 0u);
        if (t < p.n) u = *reinterpret_cast<const uint4*>(p.qkv + (row0 + t) * p.ld + c * 8);
        const int h = c / CPH;
        const uint4* qq = reinterpret_cast<const uint4*>(&Qs[rr * p.kstr + h * HP + (c - h * CPH) * 8]);
        const uint4 a = *qq;
        if (a.x != u.x || a.y != u.y || a.z != u.z || a.w != u.w) dbg_bad(8, qq);
      }
      for (int h = sc0; h < H; h += 8)
        for (int e = HD; e < HP; ++e)
          if (Qs[rr * p.kstr + h * HP + e] != 0) dbg_bad(8, &Qs[rr * p.kstr + h * HP + e]);
    }
  }
  if (g == 0) {
    for (int hp = 0; hp < H; ++hp) {
      const float* sp = &St[((wave * H + hp) * 16 + l15) * 4];
      if (__float_as_uint(sp[0]) != __float_as_uint(dbg_st[hp][0]) || __float_as_uint(sp[2]) != __float_as_uint(dbg_st[hp][1]))
        dbg_bad(10, sp);
    }
  }
  if (blockIdx.x < 1024)
    for (int i = tid; i < 4 * H * 16 * 2; i += 256) tha_dbg_st[blockIdx.x * 512 + i] = St[(i >> 1) * 4 + (i & 1) * 2];
  if (tid == 0) dbg[13] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
}

template <int H, int HG, int DT, bool QLDS>
int launch_tha_q(const ThaArgs& a, const ThaWeights& w, size_t lds, hipStream_t st) {
  static tfimm_once_t attr_done;
  if (attr_done.need()) {
    TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)tha_kernel<H, HG, DT, QLDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done.mark();
  }
  const int64_t nblocks = (int64_t)a.batch * a.qchunks;
  if (nblocks > 0x7fffffffLL) TFIMM_FAIL(TFIMM_EINVAL, "talking_heads_attention: grid too large");
  TFIMM_LAUNCH((tha_kernel<H, HG, DT, QLDS>), dim3((unsigned)nblocks), dim3(256), lds, st, a, w);
  return 0;
}

template <int H, int HG, int DT>
int launch_tha(const ThaArgs& a, const ThaWeights& w, size_t lds, hipStream_t st) {
  return a.q_in_lds ? launch_tha_q<H, HG, DT, true>(a, w, lds, st) : launch_tha_q<H, HG, DT, false>(a, w, lds, st);
}

template <int DT>
int launch_tha_heads(const ThaArgs& a, const ThaWeights& w, size_t lds, hipStream_t st) {
  switch (a.heads) {
    case 1: return launch_tha<1, 1, DT>(a, w, lds, st);
    case 2: return launch_tha<2, 2, DT>(a, w, lds, st);
    case 3: return launch_tha<3, 3, DT>(a, w, lds, st);
    case 4: return launch_tha<4, 4, DT>(a, w, lds, st);
    case 6: return launch_tha<6, 6, DT>(a, w, lds, st);
    case 8: return launch_tha<8, 8, DT>(a, w, lds, st);
    case 16: return launch_tha<16, 8, DT>(a, w, lds, st);
    default: TFIMM_FAIL(TFIMM_EUNSUP, "talking_heads_attention: %d heads not built (1, 2, 3, 4, 6, 8, 16)", a.heads);
  }
}

// ---------------------------------------------------------------------------------------------
// Catch-all talking-heads attention (any head dim / head count, unaligned rows): one workgroup per
// query row, the H x N score / probability maps of that row in LDS, plain fp32 loops.  Serves the
// reference's miniature test configuration (embed_dim 4, 2 heads); every published CaiT takes the
// MFMA kernel above.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) tha_generic_kernel(const ThaArgs p, const ThaWeights w, int hd) {
  extern __shared__ float tg[];
  const int H = p.heads, N = p.n;
  float* s0 = tg;                 // [H][N] scores, later mixed probabilities
  float* s1 = tg + (size_t)H * N; // [H][N] mixed logits / probabilities
  const int tid = threadIdx.x;
  const int img = blockIdx.x / N, i = blockIdx.x - img * N;
  const int64_t row0 = (int64_t)img * N;
  const bf16_t* qrow = p.qkv + (row0 + i) * p.ld;
  for (int id = tid; id < H * N; id += 256) {
    const int h = id / N, j = id - h * N;
    const bf16_t* krow = p.qkv + (row0 + j) * p.ld + p.dmodel + h * hd;
    float acc = 0.f;
    for (int d = 0; d < hd; ++d) acc += (p.scale * bf2f(qrow[h * hd + d])) * bf2f(krow[d]);
    s0[id] = acc;
  }
  __syncthreads();
  for (int id = tid; id < H * N; id += 256) {
    const int hp = id / N, j = id - hp * N;
    float acc = w.bl[hp];
    for (int h = 0; h < H; ++h) acc += s0[h * N + j] * w.wl[h * H + hp];
    s1[id] = acc;
  }
  __syncthreads();
  // softmax over j, one wave per mixed head
  for (int hp = tid >> 6; hp < H; hp += 4) {
    const int lane = tid & 63;
    float m = -1e30f;
    for (int j = lane; j < N; j += 64) m = fmaxf(m, s1[hp * N + j]);
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < N; j += 64) {
      const float e = __expf(s1[hp * N + j] - m);
      s1[hp * N + j] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int j = lane; j < N; j += 64) s1[hp * N + j] *= inv;
  }
  __syncthreads();
  for (int id = tid; id < H * N; id += 256) {
    const int hq = id / N, j = id - hq * N;
    float acc = w.bw[hq];
    for (int hp = 0; hp < H; ++hp) acc += s1[hp * N + j] * w.ww[hp * H + hq];
    s0[id] = acc;
  }
  __syncthreads();
  for (int id = tid; id < p.dmodel; id += 256) {
    const int hq = id / hd;
    const bf16_t* vcol = p.qkv + row0 * p.ld + 2 * p.dmodel + id;
    float acc = 0.f;
    for (int j = 0; j < N; ++j) acc += s0[hq * N + j] * bf2f(vcol[(int64_t)j * p.ld]);
    p.out[(row0 + i) * p.dmodel + id] = (bf16_t)f2bf(acc);
  }
}

// ---------------------------------------------------------------------------------------------
// Class attention: ONE query (the class token) per (image, head) against all tokens.
// One wave per (image, head): scores into LDS (lane = key), softmax by wave reductions, then
// lane = channel for the weighted sum of V.  q is already scaled (host folds scale into the q layer).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) class_attn_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kv,
                                                        bf16_t* __restrict__ out, int n, int heads, int hd, int ldq,
                                                        int ldkv, int ldo) {
  extern __shared__ float cls_s[];   // [n] scores, then probabilities
  const int lane = threadIdx.x;
  const int h = blockIdx.x % heads;
  const int img = blockIdx.x / heads;
  const int dmodel = heads * hd;
  const bf16_t* qp = q + (int64_t)img * ldq + h * hd;
  const bf16_t* kbase = kv + (int64_t)img * n * ldkv + h * hd;
  float mx = -1e30f;
  for (int j = lane; j < n; j += 64) {
    const bf16_t* kp = kbase + (int64_t)j * ldkv;
    float s = 0.f;
    for (int d = 0; d < hd; ++d) s += bf2f(qp[d]) * bf2f(kp[d]);
    cls_s[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < n; j += 64) {
    const float e = __expf(cls_s[j] - mx);
    cls_s[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __syncthreads();
  const float inv = 1.f / sum;
  for (int d = lane; d < hd; d += 64) {
    const bf16_t* vp = kbase + dmodel + d;
    float acc = 0.f;
    for (int j = 0; j < n; ++j) acc += cls_s[j] * bf2f(vp[(int64_t)j * ldkv]);
    out[(int64_t)img * ldo + h * hd + d] = (bf16_t)f2bf(acc * inv);
  }
}

__global__ void copy_rows_scalar_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int64_t total,
                                        int src_rows, int dst_rows, int dst_row0, int d) {
  const int64_t per_img = (int64_t)src_rows * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / per_img;
    dst[(b * dst_rows + dst_row0) * d + (i - b * per_img)] = src[i];
  }
}

// dst[b][dst_row0 + r][:] = src[b][r][:], 16-byte vectors
__global__ void copy_rows_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t total, int src_rows,
                                 int dst_rows, int dst_row0, int d8) {
  const int64_t per_img = (int64_t)src_rows * d8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / per_img;
    const int64_t r = i - b * per_img;
    dst[(b * dst_rows + dst_row0) * d8 + r] = src[i];
  }
}

}  // namespace

#ifdef TFIMM_THA_DBG
extern "C" __attribute__((visibility("default"))) int tfimm_hip_dbg_tha_read(void* host_dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(tha_dbg_buf), bytes, 0, hipMemcpyDeviceToHost);
}
extern "C" __attribute__((visibility("default"))) int tfimm_hip_dbg_tha_read3(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(tha_dbg_run), bytes, 0, hipMemcpyDeviceToHost);
}
extern "C" __attribute__((visibility("default"))) int tfimm_hip_dbg_tha_read2(void* st_dst, size_t st_bytes, void* ck_dst, size_t ck_bytes) {
  int rc = (int)hipMemcpyFromSymbol(st_dst, HIP_SYMBOL(tha_dbg_st), st_bytes, 0, hipMemcpyDeviceToHost);
  if (rc) return rc;
  return (int)hipMemcpyFromSymbol(ck_dst, HIP_SYMBOL(tha_dbg_ck), ck_bytes, 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int tfimm_hip_talking_heads_attention(const tfimm_tha_desc* dp, void* stream) {
  if (!dp) TFIMM_FAIL(TFIMM_EINVAL, "talking_heads_attention: null descriptor");
  const tfimm_tha_desc& d = *dp;
  if (!d.qkv || !d.out || !d.proj_l_w || !d.proj_l_b || !d.proj_w_w || !d.proj_w_b)
    TFIMM_FAIL(TFIMM_EINVAL, "talking_heads_attention: null pointer");
  if (d.batch <= 0 || d.n_tokens <= 0 || d.heads <= 0 || d.hd <= 0)
    TFIMM_FAIL(TFIMM_EINVAL, "talking_heads_attention: bad shape");
  ThaArgs a;
  a.qkv = (const bf16_t*)d.qkv; a.out = (bf16_t*)d.out;
  a.batch = d.batch; a.n = d.n_tokens; a.heads = d.heads; a.scale = d.scale;
  a.dmodel = d.heads * d.hd; a.ld = 3 * a.dmodel;
  a.qchunks = (d.n_tokens + 63) / 64;
  a.kstr = 0; a.vstr = 0; a.q_in_lds = 0;
  a.wdev = d.proj_dev;
  if (d.heads > 16) TFIMM_FAIL(TFIMM_EUNSUP, "talking_heads_attention: %d heads > 16", d.heads);
  ThaWeights w;
  for (int i = 0; i < d.heads * d.heads; ++i) { w.wl[i] = d.proj_l_w[i]; w.ww[i] = d.proj_w_w[i]; }
  for (int i = 0; i < d.heads; ++i) { w.bl[i] = d.proj_l_b[i]; w.bw[i] = d.proj_w_b[i]; }
  hipStream_t st = (hipStream_t)stream;
  const bool heads_built = d.heads <= 4 || d.heads == 6 || d.heads == 8 || d.heads == 16;
  if ((d.hd != 32 && d.hd != 48) || !heads_built || ((uintptr_t)d.qkv & 15) || ((uintptr_t)d.out & 7)) {
    const size_t lds = (size_t)2 * d.heads * d.n_tokens * 4;
    if (lds > 160 * 1024) TFIMM_FAIL(TFIMM_EUNSUP, "talking_heads_attention: hd=%d heads=%d n=%d has no kernel", d.hd, d.heads, d.n_tokens);
    static tfimm_once_t attr_done;
    if (attr_done.need()) {
      TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)tha_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_done.mark();
    }
    const int64_t nb = (int64_t)d.batch * d.n_tokens;
    if (nb > 0x7fffffffLL) TFIMM_FAIL(TFIMM_EINVAL, "talking_heads_attention: grid too large");
    TFIMM_LAUNCH(tha_generic_kernel, dim3((unsigned)nb), dim3(256), lds, st, a, w, d.hd);
    return 0;
  }
  // row strides: >= row length + 8 and == 72 (mod 128) elements, i.e. 36 dwords (mod 64 banks), which keeps
  // the 16-row ds_read_b128 fragment reads conflict free
  const int hp_ = (d.hd + 31) / 32 * 32;
  a.kstr = ((d.heads * hp_ + 8 - 72 + 127) / 128) * 128 + 72;
  a.vstr = ((a.dmodel + 8 - 72 + 127) / 128) * 128 + 72;
  const size_t base = (size_t)THA_KB * a.kstr * 2 + (size_t)THA_KB * a.vstr * 2 + (size_t)4 * d.heads * 16 * 4 * 4 +   // St: pairs
                      (size_t)2 * (d.heads * d.heads + d.heads) * 8;                                                      // Wm: pairs
  const size_t qbytes = (size_t)64 * a.kstr * 2;
  a.q_in_lds = (base + qbytes <= 160 * 1024) ? 1 : 0;   // else Q fragments are re-read from L1/L2 per key tile
  size_t lds = base + (a.q_in_lds ? qbytes : 0);
  if (lds > 160 * 1024) TFIMM_FAIL(TFIMM_EUNSUP, "talking_heads_attention: embed dim %d needs %zu bytes of LDS", a.dmodel, lds);
  return d.hd == 32 ? launch_tha_heads<2>(a, w, lds, st) : launch_tha_heads<3>(a, w, lds, st);
}

extern "C" int tfimm_hip_class_attention(const void* q, const void* kv, void* out, int B, int n_tokens, int heads,
                                         int hd, int ldq, int ldkv, int ldo, void* stream) {
  if (!q || !kv || !out) TFIMM_FAIL(TFIMM_EINVAL, "class_attention: null pointer");
  if (B <= 0 || n_tokens <= 0 || heads <= 0 || hd <= 0 || ldq < heads * hd || ldkv < 2 * heads * hd || ldo < heads * hd)
    TFIMM_FAIL(TFIMM_EINVAL, "class_attention: bad shape");
  if ((size_t)n_tokens * 4 > 64 * 1024) TFIMM_FAIL(TFIMM_EUNSUP, "class_attention: %d tokens", n_tokens);
  const int64_t nblocks = (int64_t)B * heads;
  if (nblocks > 0x7fffffffLL) TFIMM_FAIL(TFIMM_EINVAL, "class_attention: grid too large");
  TFIMM_LAUNCH(class_attn_kernel, dim3((unsigned)nblocks), dim3(64), (size_t)n_tokens * 4, (hipStream_t)stream,
               (const bf16_t*)q, (const bf16_t*)kv, (bf16_t*)out, n_tokens, heads, hd, ldq, ldkv, ldo);
  return 0;
}

extern "C" int tfimm_hip_copy_rows(const void* src, void* dst, int B, int src_rows, int dst_rows, int dst_row0, int d,
                                   void* stream) {
  if (!src || !dst) TFIMM_FAIL(TFIMM_EINVAL, "copy_rows: null pointer");
  if (B <= 0 || src_rows <= 0 || d <= 0 || dst_row0 < 0 || dst_row0 + src_rows > dst_rows)
    TFIMM_FAIL(TFIMM_EINVAL, "copy_rows: bad shape");
  if ((d & 7) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) {   // element-wise catch-all
    const int64_t total = (int64_t)B * src_rows * d;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    TFIMM_LAUNCH(copy_rows_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                 (bf16_t*)dst, total, src_rows, dst_rows, dst_row0, d);
    return 0;
  }
  const int d8 = d / 8;
  const int64_t total = (int64_t)B * src_rows * d8;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  TFIMM_LAUNCH(copy_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src,
               (uint4*)dst, total, src_rows, dst_rows, dst_row0, d8);
  return 0;
}
