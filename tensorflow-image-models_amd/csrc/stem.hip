// ResNet stem as one kernel: 7x7 stride-2 RGB convolution (+ folded BN bias, ReLU) and the 3x3 stride-2 max pooling
// behind it (resnet.py:505-512, 538-540, 572-576), gfx950.
//
// The unfused path (GEMM on the pixel-pair view, then tfimm_hip_maxpool) writes the 112x112x64 convolution output to
// HBM and reads it back (411 MB + 411 MB at batch 256), and its operand gather fetches every input pixel ~15 times
// through the texture path in 64-byte pieces that straddle cache lines.  Here a workgroup walks down one image (or a
// band of it): the input rows it needs live in an LDS ring (each row fetched once), the MFMA operand of an output
// pixel is read from that ring by address arithmetic (no im2col), two convolution rows at a time are kept in LDS as
// bf16 next to the last row of the previous step, and one pooled row per step is written from there.  Only the pooled
// 56x56x64 tensor goes to HBM.
//
// GEMM view of one convolution row y: D[n][ox] = sum_k W[n][k] X[ox][k], k = ky * 32 + pair * 8 + e (the K order of
// pack_conv on the pair view: 7 kernel rows x 4 pixel pairs x (2 pixels x 4 stored channels)), X[ox][ky, pair, :] =
// the 16 bytes at padded input row 2y + ky, pixel pair ox + pair.  v_mfma_f32_16x16x32_bf16 with the weights as the
// A operand: a lane ends up with 4 consecutive channels of one pixel.  Wave w of the 4 owns convolution row
// 2s + w/2 of step s and the channel half w%2; its 2 x 7 weight fragments stay in registers for the whole kernel.
// Measured at batch 256, 224x224 (MI355X): 132 us against 262 us (convolution) + 138 us (pooling) unfused.
#include "common.h"

#include <hip/hip_runtime.h>

#include <cstdlib>

namespace {

constexpr int kThreads = 256;          // 4 waves: (convolution row of the step) x (channel half)
constexpr int kRowB = 1856;            // LDS bytes per input row: 116 pixel pairs
constexpr int kRowPairs = kRowB / 16;
constexpr int kGroup = 4;              // input rows per fetch group
constexpr int kRing = 16;              // input rows resident: the three groups a step reads + the one in flight
constexpr int kPxB = 144;              // LDS bytes per convolution-output pixel: 64 bf16 channels + 16 (bank spread)
constexpr int kConvW = 112;            // widest convolution row
constexpr int kConvRowB = kConvW * kPxB;
constexpr int kConvRing = 3;           // convolution rows resident: the two of a step + the last of the previous step
constexpr int kLdsBytes = kRing * kRowB + kConvRing * kConvRowB;   // 78080: two workgroups per CU
constexpr int kFetch = kGroup * kRowPairs;                         // 16-byte slots per fetch group (464)

typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

struct StemArgs {
  const void* x;         // kInPairs: [B][Hp][Wp2] pixel pairs (16 bytes: 2 pixels x 4 bf16 channels); else the raw image
  const bf16_t* wt;      // [64][ldw], k = ky * 32 + pair * 8 + e
  const float* bias;     // [64]
  uint4* out;            // [B][PH][PW][8] x 16 bytes (64 bf16 channels per pixel)
  int B, Hp, Wp2, OH, OW, PH, PW, ldw;
  int steps;             // = PH: a step is two convolution rows = one pooled row
  int bands, steps_per_band, items;   // items = B * bands
  int dbg;               // TFIMM_STEM_DBG (profiling only): 1 skip pooling, 2 skip the MFMA phase, 4 skip the row prefetch
  // raw input modes (IN != kInPairs): x is the caller's [B][H][W][3] image (bf16 or float32) and the zero border /
  // 4th channel / bf16 rounding of tfimm_hip_cast_input_pad happen while the LDS ring is filled
  int H, W, pad_t, pad_l;
};

enum { kInPairs = 0, kInBf16Rgb = 1, kInF32Rgb = 2 };
typedef __attribute__((ext_vector_type(4))) unsigned int stem_u32x4;

// One 16-byte ring slot (a pixel pair) on its way from HBM: the loaded values stay untouched in registers until they
// are packed for the LDS store a step later (any arithmetic on them earlier would make hipcc wait for the load
// during the MFMA phase).  Raw modes load from clamped coordinates and zero what lies outside by a mask computed
// from the coordinates alone.
template <int IN> struct StemPre;
template <> struct StemPre<kInPairs> { stem_u32x4 v; };
template <> struct StemPre<kInBf16Rgb> { uint32_t w0, w1, w2, mask; };     // 12 bytes = two adjacent pixels, + validity / shift
struct __attribute__((packed, aligned(2))) StemBf16Pair { uint32_t w0, w1, w2; };
template <> struct StemPre<kInF32Rgb> { uint32_t a0, a1, a2, b0, b1, b2, mask; };

template <int IN>
__device__ __forceinline__ StemPre<IN> stem_fetch(const StemArgs& p, int b, int r, int pr) {
  StemPre<IN> o;
  if constexpr (IN == kInPairs) {
    const stem_u32x4* xb = reinterpret_cast<const stem_u32x4*>(p.x) + (size_t)b * p.Hp * p.Wp2;
    o.v = (r < p.Hp && pr < p.Wp2) ? xb[(size_t)r * p.Wp2 + pr] : stem_u32x4{0u, 0u, 0u, 0u};
  } else {
    const int y = r - p.pad_t, x0 = 2 * pr - p.pad_l, x1 = x0 + 1;
    const bool yok = (unsigned)y < (unsigned)p.H;
    o.mask = ((yok && (unsigned)x0 < (unsigned)p.W) ? 1u : 0u) | ((yok && (unsigned)x1 < (unsigned)p.W) ? 2u : 0u);
    const size_t rowbase = ((size_t)b * p.H + (size_t)min(max(y, 0), p.H - 1)) * p.W;
    const size_t e0 = (rowbase + min(max(x0, 0), p.W - 1)) * 3, e1 = (rowbase + min(max(x1, 0), p.W - 1)) * 3;
    if constexpr (IN == kInBf16Rgb) {
      // ONE 12-byte load of the two adjacent pixels xb, xb + 1 (xb = x0 clamped into the row); at the row ends the
      // pair straddles the border and the valid pixel is the other one of the two loaded (bits 2, 3 of mask).
      // Separate 2-byte loads would be merged by hipcc and un-merged with shifts right behind the load = a wait.
      const int xb = min(max(x0, 0), p.W - 2);
      const StemBf16Pair w = *reinterpret_cast<const StemBf16Pair*>(reinterpret_cast<const uint16_t*>(p.x) + (rowbase + xb) * 3);
      o.w0 = w.w0; o.w1 = w.w1; o.w2 = w.w2;
      o.mask |= (x0 < xb ? 4u : 0u) | (x0 > xb ? 8u : 0u);
    } else {
      const uint32_t* q = reinterpret_cast<const uint32_t*>(p.x);
      o.a0 = q[e0]; o.a1 = q[e0 + 1]; o.a2 = q[e0 + 2];
      o.b0 = q[e1]; o.b1 = q[e1 + 1]; o.b2 = q[e1 + 2];
    }
  }
  return o;
}

template <int IN>
__device__ __forceinline__ stem_u32x4 stem_pack(const StemPre<IN>& r) {
  if constexpr (IN == kInPairs) {
    return r.v;
  } else if constexpr (IN == kInBf16Rgb) {
    const uint32_t p0x = r.w0, p0y = r.w1 & 0xffffu;                         // first loaded pixel: (c0 | c1 << 16, c2)
    const uint32_t p1x = (r.w1 >> 16) | (r.w2 << 16), p1y = r.w2 >> 16;       // second
    const bool ma = r.mask & 1u, mb = r.mask & 2u, lo = r.mask & 4u, hi = r.mask & 8u;
    return stem_u32x4{ma ? (hi ? p1x : p0x) : 0u, ma ? (hi ? p1y : p0y) : 0u,
                      mb ? (lo ? p0x : p1x) : 0u, mb ? (lo ? p0y : p1y) : 0u};
  } else {
    // the rounding of tfimm_hip_cast_input (round to nearest even)
    const uint32_t a0 = f2bf(__uint_as_float(r.a0)), a1 = f2bf(__uint_as_float(r.a1)), a2 = f2bf(__uint_as_float(r.a2));
    const uint32_t b0 = f2bf(__uint_as_float(r.b0)), b1 = f2bf(__uint_as_float(r.b1)), b2 = f2bf(__uint_as_float(r.b2));
    const bool ma = r.mask & 1u, mb = r.mask & 2u;
    return stem_u32x4{ma ? (a0 | (a1 << 16)) : 0u, ma ? a2 : 0u, mb ? (b0 | (b1 << 16)) : 0u, mb ? b2 : 0u};
  }
}

// Two workgroups share a CU (78 KB of LDS each) and are in different phases of their steps most of the time, so
// one's MFMA phase overlaps the other's pooling / barriers.
template <int IN>
__global__ void __launch_bounds__(kThreads, 2) stem_pool_kernel(const StemArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sIn = smem;
  char* const sConv = smem + kRing * kRowB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mg = wave >> 1, nh = wave & 1;
  const int lm = lane & 15, lg = lane >> 4;

  // this wave's weights (A operand: lane = channel lm of the block, k group lg) and bias (the lane's 4 channels)
  bf16x8 wf[2][7];
  f32x4 bias4[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = nh * 32 + nb * 16;
#pragma unroll
    for (int ks = 0; ks < 7; ++ks)
      wf[nb][ks] = *reinterpret_cast<const bf16x8*>(p.wt + (size_t)(n + lm) * p.ldw + ks * 32 + lg * 8);
    const float4 b = *reinterpret_cast<const float4*>(p.bias + n + 4 * lg);
    bias4[nb] = f32x4{b.x, b.y, b.z, b.w};
  }
  // The weights must have landed before the work loop: otherwise hipcc's waits for them inside the loop (counted
  // against the loads issued since) would also wait for each step's row prefetch in the middle of the MFMA phase.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) asm volatile("" : "+v"(wf[nb][ks]));
    asm volatile("" : "+v"(bias4[nb]));
  }
  const int mblocks = (p.OW + 15) >> 4;

  for (int item = blockIdx.x; item < p.items; item += gridDim.x) {
    const int b = item / p.bands, band = item - b * p.bands;
    const int s_begin = band * p.steps_per_band;
    const int s_end = min(p.steps, s_begin + p.steps_per_band);
    const int s_first = s_begin > 0 ? s_begin - 1 : 0;     // a band below the top recomputes the step above it: its
                                                            // last convolution row is the pooling window's top row
    // input rows 4g .. 4g+3 -> ring slots 4 (g % 4) ..; rows / pairs beyond the image are zero
    auto fetch = [&](int g, int idx) __attribute__((always_inline)) {
      const int j = idx / kRowPairs, pr = idx - j * kRowPairs;
      return stem_fetch<IN>(p, b, kGroup * g + j, pr);
    };
    auto slot_ptr = [&](int g, int idx) -> stem_u32x4* {
      return reinterpret_cast<stem_u32x4*>(sIn + (kGroup * (g & 3)) * kRowB + idx * 16);    // idx = j * kRowPairs + pr
    };
    // (every wave is past the previous item's last read of the input ring: that item ended with a barrier + pooling)
    for (int g = s_first; g < s_first + 3; ++g) {
      for (int idx = tid; idx < kFetch; idx += kThreads) *slot_ptr(g, idx) = stem_pack<IN>(fetch(g, idx));
    }

    for (int s = s_first; s < s_end; ++s) {
      // step s: convolution rows 2s, 2s+1 from input rows 4s .. 4s+8 (groups s, s+1, s+2) -> pooled row s.
      // The next step's new rows (group s + 3) are requested now and written to the ring after the arithmetic.
      const bool pf = !(TFIMM_PROBE(p.dbg) & 4);
      StemPre<IN> pre0 = {}, pre1 = {};
      if (pf) pre0 = fetch(s + 3, tid);
      if (pf && tid + kThreads < kFetch) pre1 = fetch(s + 3, tid + kThreads);
      __syncthreads();     // groups s .. s+2 are in the ring; the previous step's pooling is done with the conv ring

      const int y = 2 * s + mg;                       // this wave's convolution row (may be >= OH: computed, never pooled)
      int rowoff[7];
#pragma unroll
      for (int ks = 0; ks < 7; ++ks) rowoff[ks] = ((2 * y + ks) & (kRing - 1)) * kRowB + (lm + lg) * 16;
      char* const crow = sConv + (y % kConvRing) * kConvRowB + lm * kPxB + (nh * 32 + 4 * lg) * 2;

      // NBLK (1 or 2) pixel blocks of 16 at a time: with two, four independent accumulator chains
      auto load_blocks = [&](bf16x8 (*dst)[7], int mb, int nblk) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (h >= nblk) break;
#pragma unroll
          for (int ks = 0; ks < 7; ++ks)
            dst[h][ks] = *reinterpret_cast<const bf16x8*>(sIn + rowoff[ks] + (mb + h) * 256);
        }
      };
      auto store_block = [&](const f32x4* acc, int mb) __attribute__((always_inline)) {
        uint2 w[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const f32x4 v = acc[nb];
          const tfimm_f32x2 lo = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)}, hi = {fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
          // (sign cleared: a -0.0 would win the unsigned maximum of the pooling below)
          w[nb].x = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, tfimm_bf16x2)) & 0x7fff7fffu;
          w[nb].y = __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, tfimm_bf16x2)) & 0x7fff7fffu;
        }
        *reinterpret_cast<uint2*>(crow + mb * 16 * kPxB) = w[0];
        *reinterpret_cast<uint2*>(crow + mb * 16 * kPxB + 32) = w[1];
      };
      auto multiply = [&](bf16x8 (*xf)[7], int mb, int nblk) __attribute__((always_inline)) {
        f32x4 acc[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[h][nb] = bias4[nb];          // accumulate on top of the bias
#pragma unroll
        for (int ks = 0; ks < 7; ++ks)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (h >= nblk) break;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
              acc[h][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb][ks], xf[h][ks], acc[h][nb], 0, 0, 0);
          }
        store_block(acc[0], mb);
        if (nblk > 1) store_block(acc[1], mb + 1);
      };
      const int mb_end = (TFIMM_PROBE(p.dbg) & 2) ? 0 : mblocks;
      bf16x8 xf[2][2][7];          // [buffer][block of the pair][kernel row]
      if (mb_end > 0) load_blocks(xf[0], 0, mb_end > 1 ? 2 : 1);
#pragma unroll 1
      for (int mb = 0; mb < mb_end; mb += 4) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int m0 = mb + 2 * u;
          if (m0 >= mb_end) break;
          const int left = mb_end - m0 - 2;                 // blocks after this pair
          if (left >= 2) load_blocks(xf[u ^ 1], m0 + 2, 2);
          else if (left == 1) load_blocks(xf[u ^ 1], m0 + 2, 1);
          if (left >= 0) multiply(xf[u], m0, 2);
          else multiply(xf[u], m0, 1);
        }
      }
      // the prefetched rows: slot group (s + 3) % 4 held group s - 1, which no wave reads any more (barrier above)
      *slot_ptr(s + 3, tid) = stem_pack<IN>(pre0);
      if (tid + kThreads < kFetch) *slot_ptr(s + 3, tid + kThreads) = stem_pack<IN>(pre1);
      __syncthreads();     // this step's two convolution rows are in LDS

      if (s >= s_begin && !(TFIMM_PROBE(p.dbg) & 1)) {
        // pooled row s: 3x3 window, stride 2, one zero row / column of padding on top / left (values are >= 0 after
        // the ReLU, so padding never wins and bf16 order is unsigned-integer order)
        const int py = s;
        const int cy0 = 2 * py;
        const int ry[3] = {cy0 > 0 ? cy0 - 1 : 0, cy0, cy0 + 1 < p.OH ? cy0 + 1 : cy0};
        for (int it = tid; it < p.PW * 8; it += kThreads) {
          const int c8 = it & 7, px = it >> 3;
          // window rows / columns outside the convolution output are replaced by the nearest one inside, which is
          // in the window too (the maximum does not mind the duplicate): nine unconditional reads
          const int cx0 = 2 * px;
          const int cx[3] = {cx0 > 0 ? cx0 - 1 : 0, cx0, cx0 + 1 < p.OW ? cx0 + 1 : cx0};
          u16x8 v[9];
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const char* const row = sConv + (ry[a] % kConvRing) * kConvRowB + c8 * 16;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[a * 3 + c] = *reinterpret_cast<const u16x8*>(row + cx[c] * kPxB);
          }
          u16x8 m = v[0];
#pragma unroll
          for (int e = 1; e < 9; ++e) m = __builtin_elementwise_max(m, v[e]);
          p.out[(((size_t)b * p.PH + py) * p.PW + px) * 8 + c8] = __builtin_bit_cast(uint4, m);
        }
      }
    }
  }
}

}  // namespace

extern "C" int tfimm_hip_stem_conv_pool(const tfimm_stem_desc* d, void* stream) {
  if (!d || !d->x || !d->wt || !d->bias || !d->out || d->batch <= 0 || d->OH <= 0 || d->OW <= 0 || d->OW > kConvW ||
      d->Wp2 <= 0 || d->Wp2 > kRowPairs || d->ldw < 224 || (d->ldw & 7) || d->Hp < 2 * (d->OH - 1) + 7 ||
      d->Wp2 < d->OW + 3 || ((uintptr_t)d->wt & 15) || ((uintptr_t)d->bias & 15) || ((uintptr_t)d->out & 15) ||
      d->in_dtype < 0 || d->in_dtype > 2)
    TFIMM_FAIL(TFIMM_EINVAL, "stem_conv_pool: bad arguments (7x7 stride-2 stem, 64 channels, OW <= 112, Wp/2 <= 116)");
  if (d->in_dtype == 0 ? ((uintptr_t)d->x & 15) != 0
                       : (d->H <= 0 || d->W < 2 || d->pad_t < 0 || d->pad_l < 0 || ((uintptr_t)d->x & (d->in_dtype == 1 ? 1 : 3))))
    TFIMM_FAIL(TFIMM_EINVAL, "stem_conv_pool: bad input description");
  StemArgs a;
  a.x = d->x; a.wt = (const bf16_t*)d->wt; a.bias = d->bias; a.out = (uint4*)d->out;
  a.H = d->H; a.W = d->W; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
  a.B = d->batch; a.Hp = d->Hp; a.Wp2 = d->Wp2; a.OH = d->OH; a.OW = d->OW; a.ldw = d->ldw;
  a.PH = (d->OH - 1) / 2 + 1;
  a.PW = (d->OW - 1) / 2 + 1;
  a.steps = a.PH;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  // bands per image: two workgroups per CU at a time; a band below the top pays one extra step.  Pick the split with
  // the shortest makespan in steps.
  int best_bands = 1;
  long best = -1;
  for (int nb = 1; nb <= a.steps; ++nb) {
    const int spb = (a.steps + nb - 1) / nb;
    const int real = (a.steps + spb - 1) / spb;          // bands that actually get steps
    if (real != nb) continue;
    const long rounds = ((long)a.B * nb + 2 * cus - 1) / (2 * cus);
    const long cost = rounds * (spb + (nb > 1 ? 1 : 0));
    if (best < 0 || cost < best) { best = cost; best_bands = nb; }
  }
  a.bands = best_bands;
  a.steps_per_band = (a.steps + best_bands - 1) / best_bands;
  a.items = a.B * a.bands;
  static const int dbg = getenv("TFIMM_STEM_DBG") ? atoi(getenv("TFIMM_STEM_DBG")) : 0;
  a.dbg = dbg;
  static tfimm_once_t attr_set;
  if (attr_set.need()) {
    TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)stem_pool_kernel<kInPairs>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)stem_pool_kernel<kInBf16Rgb>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)stem_pool_kernel<kInF32Rgb>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    attr_set.mark();
  }
  const int grid = a.items < 2 * cus ? a.items : 2 * cus;
  const dim3 gd((unsigned)grid), bd(kThreads);
  if (d->in_dtype == 0) TFIMM_LAUNCH(stem_pool_kernel<kInPairs>, gd, bd, (size_t)kLdsBytes, (hipStream_t)stream, a);
  else if (d->in_dtype == 1) TFIMM_LAUNCH(stem_pool_kernel<kInBf16Rgb>, gd, bd, (size_t)kLdsBytes, (hipStream_t)stream, a);
  else TFIMM_LAUNCH(stem_pool_kernel<kInF32Rgb>, gd, bd, (size_t)kLdsBytes, (hipStream_t)stream, a);
  return 0;
}
