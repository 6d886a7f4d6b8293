// bf16 MFMA GEMM / implicit-GEMM convolution, LDS-DMA variant (gfx950).
//
// Same math and fused epilogue as gemm_kernel.h, different data movement:
//   * operand tiles go global -> LDS with `buffer_load_dwordx4 ... lds` (LDS-DMA): no VGPR
//     round trip, no ds_write pass, and the buffer descriptor's bounds check zero-fills every
//     out-of-range chunk (tile tails, conv padding taps) for free -- an invalid lane simply
//     gets an offset past num_records.
//   * the DMA writes LDS linearly (wave-uniform base + lane*16 B), so the XOR swizzle that
//     keeps the ds_read_b128 fragment reads conflict free is applied to the per-lane SOURCE
//     address instead (lane l fetches the chunk that belongs in physical slot l).
//   * tiles up to 256x256x64 with 8 waves (128x64 per wave): one K-tile of MFMA work
//     (~2k cycles) is long enough to hide the HBM/L2 latency of the next tile's DMA, which is
//     issued right after the barrier that publishes the current tile (2-stage LDS ring, one
//     barrier per K-tile, raw s_barrier + explicit vmcnt so nothing drains early).
//   * epilogue: accumulators are staged through LDS (fp32, 64 output rows per pass) so that
//     global stores and residual loads are full 16-byte-per-lane row segments.
#pragma once
#include "gemm_kernel.h"

namespace tfimm_gemm {

constexpr unsigned kOobOffset = 0x7fffff00u;  // >= any num_records we accept (tensors <= 2 GiB)

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

struct GemmDmaArgs {
  GemmArgs g;
  unsigned a_bytes, w_bytes;  // buffer sizes for the descriptors' bounds checks
};

typedef void (*gemm_dma_fn)(const GemmDmaArgs);

template <int BM, int BN, int WAVES_M, int WAVES_N, int KMODE>
__global__ void __launch_bounds__(WAVES_M* WAVES_N * 64) gemm_dma_kernel(const GemmDmaArgs pa) {
  const GemmArgs& p = pa.g;
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int A_INSTR = BM / 8 / NW;  // 1-KiB DMA instructions per wave per K-tile
  constexpr int B_INSTR = BN / 8 / NW;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(A_INSTR >= 1 && B_INSTR >= 1 && TM >= 1 && TN >= 1, "tile/wave mismatch");
  static_assert(KMODE == K_DENSE || KMODE == K_CONV, "LDS-DMA flavours: dense rows or Cin % 8 == 0 gather");
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  int tile;
  {
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, i = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
  }
  const int mt = tile / p.tiles_n, nt = tile - mt * p.tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;

  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(p.a, pa.a_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(p.wt, pa.w_bytes);

  // ---- per-lane DMA source state: lane fills physical 16-byte slot `lane` of each 1-KiB piece
  const int lrow = lane >> 3;   // row within the 8-row piece
  const int lpc = lane & 7;     // physical chunk
  unsigned a_off[A_INSTR];      // dense: byte offset of (row, chunk) at k = 0, or kOobOffset
  int a_iy0[A_INSTR], a_ix0[A_INSTR], a_pix[A_INSTR];
  int a_chunk[A_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int r = (wave * A_INSTR + j) * 8 + lrow;
    const int chunk = lpc ^ ((r >> 1) & 7);
    a_chunk[j] = chunk;
    const int m = m0 + r;
    const bool ok = m < p.M;
    if (KMODE == K_DENSE) {
      a_off[j] = ok ? (unsigned)(((size_t)m * p.lda + chunk * 8) * 2) : kOobOffset;
      a_iy0[j] = a_ix0[j] = a_pix[j] = 0;
    } else {
      const int mm = ok ? m : 0;
      const int ohw = p.OH * p.OW;
      const int b = mm / ohw;
      const int rem = mm - b * ohw;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      a_iy0[j] = ok ? oy * p.stride - p.pad_t : -(1 << 28);
      a_ix0[j] = ox * p.stride - p.pad_l;
      a_pix[j] = b * p.H * p.W;
      a_off[j] = 0;
    }
  }
  unsigned b_off[B_INSTR];
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    const int r = (wave * B_INSTR + j) * 8 + lrow;
    const int chunk = lpc ^ ((r >> 1) & 7);
    const int n = n0 + r;
    b_off[j] = (n < p.N) ? (unsigned)(((size_t)n * p.ldw + chunk * 8) * 2) : kOobOffset;
  }

  auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
    char* sa = smem + stage * STAGE;
    char* sb = sa + A_BYTES;
    const int kbytes = kt * 128;
    // ---- B (weights): rows are zero padded to a multiple of 64 -> always in range in k
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(sb + (wave * B_INSTR + j) * 1024), 16,
                                               (int)b_off[j], kbytes, 0, 0);
    }
    // ---- A (activations)
    if (KMODE == K_DENSE) {
#pragma unroll
      for (int j = 0; j < A_INSTR; ++j) {
        const bool kok = (kt * BK + a_chunk[j] * 8) < p.K;
        const unsigned off = kok ? a_off[j] : kOobOffset;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sa + (wave * A_INSTR + j) * 1024), 16,
                                                 (int)off, kbytes, 0, 0);
      }
    } else if (KMODE == K_CONV) {
#pragma unroll
      for (int j = 0; j < A_INSTR; ++j) {
        const int kg = kt * BK + a_chunk[j] * 8;
        const int tap = kg / p.Cin;
        const int ci = kg - tap * p.Cin;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        const int iy = a_iy0[j] + ky, ix = a_ix0[j] + kx;
        const bool ok = kg < p.K && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const unsigned off = ok ? (unsigned)((((size_t)(a_pix[j] + iy * p.W + ix)) * p.cpitch + ci) * 2) : kOobOffset;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sa + (wave * A_INSTR + j) * 1024), 16,
                                                 (int)off, 0, 0, 0);
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (p.K + BK - 1) / BK;
  const int frow = lane & 31;
  const int fhi = lane >> 5;

  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt have landed
    tfimm_lds_reuse_barrier();                         // ... everyone's; and stage cur^1 is free again
    asm volatile("" ::: "memory");
    if (kt + 1 < nk) issue(kt + 1, cur ^ 1);

    const uint4* sA = reinterpret_cast<const uint4*>(smem + cur * STAGE);
    const uint4* sB = reinterpret_cast<const uint4*>(smem + cur * STAGE + A_BYTES);
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
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue through LDS: pass `i` handles row-tile i of every wave row (WAVES_M * 32 rows)
  tfimm_lds_reuse_barrier();   // all MFMA reads of the last stage are done: LDS is free
  constexpr int CROW = BN + 4;                       // fp32 row stride (+16 B: conflict-free b128 writes)
  float* sC = reinterpret_cast<float*>(smem);
  constexpr int PASS_ROWS = WAVES_M * 32;
  constexpr int NTHR = NW * 64;
  constexpr int CHUNKS_PER_ROW = BN / 8;             // 8 outputs (16 B of bf16) per thread per chunk
  static_assert(PASS_ROWS * CROW * 4 <= 2 * STAGE, "epilogue staging does not fit in LDS");
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    // write this wave's 32 x WTN block: lane holds row frow, 4 consecutive n per quad
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = wn * WTN + j * 32 + q * 8 + fhi * 4;
        float4 v = make_float4(acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]);
        *reinterpret_cast<float4*>(&sC[(wm * 32 + frow) * CROW + col]) = v;
      }
    __syncthreads();
    // read back row-contiguous: thread handles 8 consecutive outputs of one row
    for (int id = tid; id < PASS_ROWS * CHUNKS_PER_ROW; id += NTHR) {
      const int pr = id / CHUNKS_PER_ROW;            // row within the pass
      const int c8 = id - pr * CHUNKS_PER_ROW;
      const int wrow = pr >> 5;                      // which wave row
      const int m = m0 + wrow * WTM + i * 32 + (pr & 31);
      const int n = n0 + c8 * 8;
      if (m >= p.M || n >= p.N) continue;
      float v[8];
      {
        const float4 lo = *reinterpret_cast<const float4*>(&sC[pr * CROW + c8 * 8]);
        const float4 hi = *reinterpret_cast<const float4*>(&sC[pr * CROW + c8 * 8 + 4]);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
      }
      const bool full = (n + 7) < p.N;
      if (p.bias) {
        if (full) {
          const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n);
          const float4 b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
          v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (n + e < p.N) v[e] += p.bias[n + e];
        }
      }
      if (!p.act_after_res && p.act != TFIMM_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], p.act);
      }
      if (p.residual) {
        const int rm = p.res_mod > 0 ? (m % p.res_mod) : m;
        const bf16_t* rr = p.residual + (size_t)rm * p.ldr + n;
        if (full && p.res_vec16) {
          float r8[8];
          unpack8(*reinterpret_cast<const uint4*>(rr), r8);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += r8[e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (n + e < p.N) v[e] += bf2f(rr[e]);
        }
      }
      if (p.act_after_res && p.act != TFIMM_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], p.act);
      }
      const int om = p.remap_in > 0 ? (m / p.remap_in) * p.remap_out + (m % p.remap_in) + p.remap_off : m;
      if (p.out_f32) {
        float* o = reinterpret_cast<float*>(p.out) + (size_t)om * p.ldc + n;
        if (full && p.out_vec) {
          *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (n + e < p.N) o[e] = v[e];
        }
      } else {
        bf16_t* o = reinterpret_cast<bf16_t*>(p.out) + (size_t)om * p.ldc + n;
        if (full && p.out_vec16) {
          *reinterpret_cast<uint4*>(o) = pack8(v);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (n + e < p.N) o[e] = (bf16_t)f2bf(v[e]);
        }
      }
    }
    __syncthreads();
  }
}

struct DmaTileCfg {
  int bm, bn, threads;
  gemm_dma_fn fn[2];  // K_DENSE, K_CONV
};

}  // namespace tfimm_gemm

// DMA tile shapes: id, BM, BN, WAVES_M, WAVES_N
#define TFIMM_GEMM_DMA_TILES(X) \
  X(0, 256, 256, 2, 4)          \
  X(1, 256, 128, 4, 2)          \
  X(2, 128, 128, 2, 2)          \
  X(3, 256, 64, 4, 2)           \
  X(4, 128, 64, 2, 2)           \
  X(5, 128, 256, 2, 4)
#define TFIMM_GEMM_DMA_NUM_TILES 6
