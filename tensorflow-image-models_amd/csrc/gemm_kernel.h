// bf16 MFMA GEMM / implicit-GEMM convolution kernel template (gfx950).
//
// One kernel family serves every dense contraction on the tfimm forward path
// (reference call sites listed in include/tfimm_hip.h):  out = epi(A . Wt^T).
//
// Design:
//   * block tile BM x BN x 64, 64-lane waves arranged WAVES_M x WAVES_N, each wave owns a
//     (BM/WAVES_M) x (BN/WAVES_N) sub-tile built from v_mfma_f32_32x32x16_bf16.
//   * operands are swapped (a = weight rows, b = activation rows) so every lane ends up
//     with 4 CONSECUTIVE output channels of one output row per accumulator quad -> packed
//     8-byte bf16 stores, per-channel bias as a register quad.
//   * A (activations) is fetched either as dense rows or gathered straight from the NHWC
//     image (im2col-free): a 16-byte chunk = 8 input channels of one filter tap
//     (Cin % 8 == 0) or two horizontally adjacent 4-channel pixels (padded-RGB stems and
//     patch embeddings).
//   * global -> VGPR -> LDS staging, double-buffered LDS.  The next tile's global loads are
//     issued before the current tile's MFMAs and only consumed (masked + written to LDS)
//     after them, so HBM/L2 latency hides under the matrix work.  Out-of-range chunks are
//     fetched from a safe in-bounds address and zeroed by a register select at write time
//     (a `ok ? *p : 0` ternary makes clang select a *stack* zero and go through scratch).
//   * 16-byte LDS slots are XOR-swizzled (slot ^= (row>>1)&7): the ds_read_b128 fragment
//     reads are bank-conflict free for 128-byte rows.
//   * block id -> tile map is XCD-aware: each of the 8 XCDs (private L2) walks a contiguous
//     range of M-tiles over all N-tiles, so an activation panel is pulled through one L2.
#pragma once
#include "common.h"

namespace tfimm_gemm {

constexpr int BK = 64;  // K elements per LDS tile (128-byte rows)

// kernel flavours (template parameter KMODE)
enum {
  K_DENSE = 0,         // dense rows, 16-byte aligned
  K_CONV = 1,          // NHWC gather, Cin % 8 == 0
  K_CONV_C4 = 2,       // NHWC gather, Cin == 4
  K_DENSE_SCALAR = 3,  // dense rows, arbitrary K / lda / alignment (element loads)
  K_DENSE_SCALE = 4,   // dense rows * per-(image, k) SE gate
  K_CONV_SCALAR = 5,   // NHWC gather, arbitrary Cin (element loads; odd-shaped test models)
  K_NUM = 6
};

struct GemmArgs {
  const bf16_t* a;
  const bf16_t* wt;
  const float* bias;
  const bf16_t* residual;
  void* out;
  const float* a_scale;
  int M, N, K;
  int lda, ldw, ldr, ldc;
  int out_f32, act, act_after_res, res_mod;
  int remap_in, remap_out, remap_off;
  int B, H, W, Cin, KH, KW, KWp, stride, pad_t, pad_l, OH, OW;
  int stride_w;   // horizontal stride (== stride unless the pixel-pair view of an RGB stem is used)
  int cpitch;     // elements between consecutive pixels of the conv input (Cin unless a channel slice is convolved)
  int rows_per_image;
  int res_vec;    // residual rows can be read as aligned 8-byte quads
  int out_vec;    // output rows can be written as aligned quads
  int res_vec16;  // ... as aligned 16-byte octets (LDS-staged epilogue)
  int out_vec16;
  int tiles_m, tiles_n;
};

typedef void (*gemm_fn)(const GemmArgs);

__device__ __forceinline__ int lds_slot(int row, int chunk) {
  return row * 8 + (chunk ^ ((row >> 1) & 7));
}
__device__ __forceinline__ uint4 sel4(bool ok, const uint4& v) {
  return make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int KMODE>
__global__ void __launch_bounds__(WAVES_M* WAVES_N * 64) gemm_kernel(const GemmArgs p) {
  constexpr int NTHR = WAVES_M * WAVES_N * 64;
  constexpr int ROWS_PER_PASS = NTHR / 8;
  constexpr int A_ITERS = BM / ROWS_PER_PASS;
  constexpr int B_ITERS = BN / ROWS_PER_PASS;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr bool DENSE = (KMODE == K_DENSE || KMODE == K_DENSE_SCALAR || KMODE == K_DENSE_SCALE);
  static_assert(A_ITERS >= 1 && B_ITERS >= 1 && TM >= 1 && TN >= 1, "tile too small");

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint4* lds = reinterpret_cast<uint4*>(smem_raw);
  constexpr int A_SLOTS = BM * 8, B_SLOTS = BN * 8;
  uint4* ldsA0 = lds;                 // [A buf0][A buf1]
  uint4* ldsB0 = lds + 2 * A_SLOTS;   // [B buf0][B buf1]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- XCD-aware block -> tile map (bijective for any grid size) ----
  int tile;
  {
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, i = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  const int mt = tile / p.tiles_n, nt = tile - mt * p.tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread loader state ----
  const int lc = tid & 7;   // 16-byte chunk within the 64-wide K tile
  const int lr = tid >> 3;  // first row handled by this thread
  const bf16_t* a_ptr[A_ITERS];
  int a_iy0[A_ITERS], a_ix0[A_ITERS], a_pix[A_ITERS];
  bool a_ok[A_ITERS];
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) {
    const int m = m0 + lr + i * ROWS_PER_PASS;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    if (DENSE) {
      a_ptr[i] = p.a + (size_t)mm * p.lda;
      a_iy0[i] = a_ix0[i] = a_pix[i] = 0;
    } else {
      const int ohw = p.OH * p.OW;
      const int b = mm / ohw;
      const int rem = mm - b * ohw;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      a_iy0[i] = a_ok[i] ? oy * p.stride - p.pad_t : -(1 << 28);
      a_ix0[i] = ox * p.stride - p.pad_l;
      a_pix[i] = b * p.H * p.W;
      a_ptr[i] = p.a;
    }
  }
  const bf16_t* b_ptr[B_ITERS];
  bool b_ok[B_ITERS];
#pragma unroll
  for (int i = 0; i < B_ITERS; ++i) {
    const int n = n0 + lr + i * ROWS_PER_PASS;
    b_ok[i] = n < p.N;
    b_ptr[i] = p.wt + (size_t)(b_ok[i] ? n : 0) * p.ldw;
  }

  uint4 ra[A_ITERS], rb[B_ITERS];

  // validity of this thread's chunk for k-tile kt (recomputed at write time: integer ops only)
  auto a_valid = [&](int kt, int i, bool* ok1) __attribute__((always_inline)) -> bool {
    *ok1 = false;
    if (DENSE) {
      return a_ok[i] && (kt * BK + lc * 8) < p.K;
    } else if (KMODE == K_CONV_SCALAR) {
      return a_ok[i];
    } else if (KMODE == K_CONV) {
      const int kg = kt * BK + lc * 8;
      const int tap = kg / p.Cin;
      const int ky = tap / p.KW, kx = tap - ky * p.KW;
      const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
      return kg < p.K && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    } else {
      const int kc = kt * 8 + lc;
      const int half = p.KWp >> 1;
      const int ky = kc / half;
      const int kx0 = (kc - ky * half) * 2;
      const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx0;
      const bool rok = ky < p.KH && (unsigned)iy < (unsigned)p.H;
      *ok1 = rok && (kx0 + 1) < p.KW && (unsigned)(ix + 1) < (unsigned)p.W;
      return rok && kx0 < p.KW && (unsigned)ix < (unsigned)p.W;
    }
  };

  auto load_tile = [&](int kt) __attribute__((always_inline)) {
    const int kg = kt * BK + lc * 8;
    {  // B (weights): aligned, zero padded to ldw
      const int kb = (kg < p.ldw) ? kg : 0;
#pragma unroll
      for (int i = 0; i < B_ITERS; ++i) rb[i] = *reinterpret_cast<const uint4*>(b_ptr[i] + kb);
    }
    if (KMODE == K_DENSE || KMODE == K_DENSE_SCALE) {
      const int ka = (kg < p.K) ? kg : 0;
#pragma unroll
      for (int i = 0; i < A_ITERS; ++i) ra[i] = *reinterpret_cast<const uint4*>(a_ptr[i] + ka);
    } else if (KMODE == K_DENSE_SCALAR) {
#pragma unroll
      for (int i = 0; i < A_ITERS; ++i) {
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = kg + e;
          const bool ok = k < p.K;
          uint32_t v = (uint32_t)a_ptr[i][ok ? k : 0];
          v = ok ? v : 0u;
          w[e >> 1] |= v << ((e & 1) * 16);
        }
        ra[i] = make_uint4(w[0], w[1], w[2], w[3]);
      }
    } else if (KMODE == K_CONV) {
      const int tap = kg / p.Cin;
      const int ci = kg - tap * p.Cin;
      const int ky = tap / p.KW, kx = tap - ky * p.KW;
      const bool kok = kg < p.K;
#pragma unroll
      for (int i = 0; i < A_ITERS; ++i) {
        const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
        const bool ok = kok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const size_t off = ok ? ((size_t)(a_pix[i] + iy * p.W + ix)) * p.cpitch + ci : (size_t)0;
        ra[i] = *reinterpret_cast<const uint4*>(p.a + off);
      }
    } else if (KMODE == K_CONV_SCALAR) {
#pragma unroll
      for (int i = 0; i < A_ITERS; ++i) {
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = kg + e;
          const int tap = k / p.Cin;
          const int ci = k - tap * p.Cin;
          const int ky = tap / p.KW, kx = tap - ky * p.KW;
          const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
          const bool ok = k < p.K && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
          const size_t off = ok ? ((size_t)(a_pix[i] + iy * p.W + ix)) * p.cpitch + ci : (size_t)0;
          uint32_t v = (uint32_t)p.a[off];
          v = ok ? v : 0u;
          w[e >> 1] |= v << ((e & 1) * 16);
        }
        ra[i] = make_uint4(w[0], w[1], w[2], w[3]);
      }
    } else {  // K_CONV_C4: chunk = two adjacent 4-channel pixels of one filter row
#pragma unroll
      for (int i = 0; i < A_ITERS; ++i) {
        bool ok1;
        const bool ok0 = a_valid(kt, i, &ok1);
        const int kc = kt * 8 + lc;
        const int half = p.KWp >> 1;
        const int ky = kc / half;
        const int kx0 = (kc - ky * half) * 2;
        const size_t pix = (size_t)(a_pix[i] + (a_iy0[i] + ky) * p.W + a_ix0[i] + kx0);
        const uint2* base = reinterpret_cast<const uint2*>(p.a);
        const uint2 v0 = base[ok0 ? pix : (size_t)0];
        const uint2 v1 = base[ok1 ? pix + 1 : (size_t)0];
        ra[i] = make_uint4(v0.x, v0.y, v1.x, v1.y);
      }
    }
  };

  auto store_tile = [&](int buf, int kt) __attribute__((always_inline)) {
    uint4* dA = ldsA0 + buf * A_SLOTS;
    uint4* dB = ldsB0 + buf * B_SLOTS;
    const int kg = kt * BK + lc * 8;
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      bool ok1;
      const bool ok0 = a_valid(kt, i, &ok1);
      uint4 v = ra[i];
      if (KMODE == K_CONV_C4) {
        v = make_uint4(ok0 ? v.x : 0u, ok0 ? v.y : 0u, ok1 ? v.z : 0u, ok1 ? v.w : 0u);
      } else if (KMODE == K_DENSE_SCALAR || KMODE == K_CONV_SCALAR) {
        v = sel4(a_ok[i], v);  // per-element masking already applied at load
      } else {
        v = sel4(ok0, v);
      }
      if (KMODE == K_DENSE_SCALE) {
        // SqueezeExcite gate folded into the projection conv: A[m][k] *= gate[image(m)][k]
        const int m = m0 + lr + i * ROWS_PER_PASS;
        const int img = (a_ok[i] ? m : 0) / p.rows_per_image;
        const float* g = p.a_scale + (size_t)img * p.K + (kg < p.K ? kg : 0);
        const float4 g0 = reinterpret_cast<const float4*>(g)[0];
        const float4 g1 = reinterpret_cast<const float4*>(g)[1];
        float f[8];
        unpack8(v, f);
        f[0] *= g0.x; f[1] *= g0.y; f[2] *= g0.z; f[3] *= g0.w;
        f[4] *= g1.x; f[5] *= g1.y; f[6] *= g1.z; f[7] *= g1.w;
        v = sel4(ok0, pack8(f));
      }
      dA[lds_slot(lr + i * ROWS_PER_PASS, lc)] = v;
    }
    const bool kok = kg < p.ldw;
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) dB[lds_slot(lr + i * ROWS_PER_PASS, lc)] = sel4(b_ok[i] && kok, rb[i]);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (p.K + BK - 1) / BK;
  const int frow = lane & 31;  // fragment row within a 32-row MFMA tile
  const int fhi = lane >> 5;   // which 8-wide K half of the 16-deep MFMA step

  load_tile(0);
  store_tile(0, 0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = (kt + 1) < nk;
    if (more) load_tile(kt + 1);

    const uint4* sA = ldsA0 + cur * A_SLOTS;
    const uint4* sB = ldsB0 + cur * B_SLOTS;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = __builtin_bit_cast(bf16x8, sA[lds_slot(wm * WTM + i * 32 + frow, ks * 2 + fhi)]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[j] = __builtin_bit_cast(bf16x8, sB[lds_slot(wn * WTN + j * 32 + frow, ks * 2 + fhi)]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          // swapped operands: D[row = n][col = m]
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }

    if (more) store_tile(cur ^ 1, kt + 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds, for output row m (col of D), channels n = nb + 8*q + 4*fhi + e
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WTM + i * 32 + frow;
    if (m >= p.M) continue;
    const int rm = p.res_mod > 0 ? (m % p.res_mod) : m;
    const int om = p.remap_in > 0 ? (m / p.remap_in) * p.remap_out + (m % p.remap_in) + p.remap_off : m;
    const bf16_t* res_row = p.residual ? p.residual + (size_t)rm * p.ldr : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * WTN + j * 32 + q * 8 + fhi * 4;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][q * 4 + e];
        const bool full = (n + 3) < p.N;
        if (p.bias) {
          if (full) {
            const float4 bq = *reinterpret_cast<const float4*>(p.bias + n);
            v[0] += bq.x; v[1] += bq.y; v[2] += bq.z; v[3] += bq.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < p.N) v[e] += p.bias[n + e];
          }
        }
        if (!p.act_after_res && p.act != TFIMM_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
        }
        if (res_row) {
          if (full && p.res_vec) {
            const uint2 rq = *reinterpret_cast<const uint2*>(res_row + n);
            v[0] += bf2f(rq.x & 0xffffu); v[1] += bf2f(rq.x >> 16);
            v[2] += bf2f(rq.y & 0xffffu); v[3] += bf2f(rq.y >> 16);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < p.N) v[e] += bf2f(res_row[n + e]);
          }
        }
        if (p.act_after_res && p.act != TFIMM_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act);
        }
        if (p.out_f32) {
          float* o = reinterpret_cast<float*>(p.out) + (size_t)om * p.ldc + n;
          if (full && p.out_vec) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < p.N) o[e] = v[e];
          }
        } else {
          bf16_t* o = reinterpret_cast<bf16_t*>(p.out) + (size_t)om * p.ldc + n;
          if (full && p.out_vec) {
            *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < p.N) o[e] = (bf16_t)f2bf(v[e]);
          }
        }
      }
    }
  }
}

// one entry per tile shape; each lives in its own translation unit (gemm_inst.hip -DTILE_ID=n)
struct TileCfg {
  int bm, bn, threads;
  gemm_fn fn[K_NUM];  // nullptr = flavour not built for this tile
};

}  // namespace tfimm_gemm

#define TFIMM_GEMM_TILES(X) \
  X(0, 128, 128, 2, 2)      \
  X(1, 128, 64, 2, 2)       \
  X(2, 64, 64, 2, 2)        \
  X(3, 256, 128, 4, 2)      \
  X(4, 128, 256, 2, 4)      \
  X(5, 64, 128, 2, 2)
#define TFIMM_GEMM_NUM_TILES 6
