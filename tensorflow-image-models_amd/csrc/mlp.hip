// tfimm_hip_mlp_fused: a transformer MLP block in ONE launch (gfx950),
//
//     out = residual + fc2( act( fc1( LayerNorm(x) ) ) )
//
// i.e. norm2 -> mlp.fc1 -> GELU -> mlp.fc2 -> += shortcut of a Swin block (swin.py:322-325, layers/transformers.py:208-214)
// and norm -> fc1 -> GELU -> fc2 (-> LayerScale) -> += shortcut of a ConvNeXt block (convnext.py:226-232), for NARROW rows
// (C = 128 channels, 4 C = 512 hidden).  As two GEMM launches the 4C-wide hidden tensor is written and read back -- at
// Swin-B's first stage 1.6 GB per block against 0.4 GB for x and the result -- and the GELU epilogue of fc1 runs with the
// matrix pipe idle.  Here the hidden tensor never exists:
//   * a workgroup owns 256 rows (8 waves x 32 rows).  Their x rows are brought into LDS once (64 KiB, 256-byte rows,
//     16-byte chunk c of row r at physical chunk c ^ (r & 15)); every wave normalises ITS rows in registers (two-pass
//     statistics in fp32, (x - mean) * rstd rounded to bf16: gamma and beta live in W1' = gamma * W1, b1' = beta . W1 + b1) and
//     keeps them as the MFMA B operand of GEMM 1 (8 k-steps x 4 VGPRs) for the whole tile;
//   * the hidden axis is walked in GROUPS of 64 channels.  Per group and wave: GEMM 1 (32 rows x 64 hidden, K = C:
//     16 MFMAs, weights from a 2-stage LDS ring), + b1' and the exact-erf GELU in the accumulator layout, packed to bf16 --
//     and those packed accumulators ARE the B operand of GEMM 2 (one row per lane, 8 consecutive k per lane up to the
//     order of the k values, which the host bakes into W2: pack.chain_k_order, the register chain of
//     gemm_chain_kernel.h), 16 more MFMAs into the 32 x 128 output accumulators that live across all 8 groups;
//   * all eight waves walk the groups in step (one workgroup barrier per group: the ring stage of group g + 1 is requested
//     behind it).  The two waves of a SIMD are in the SAME phase on purpose: measured on gfx950
//     (tools/probes/valu_mfma_overlap_probe.hip, profiles/r03_valu_mfma_overlap_probe.txt) a wave's v_pk_fma_f32 stream
//     runs at 8.5 cycles per instruction alone, 4.5 when the OTHER wave of the SIMD is in the VALU too, and 24.5 when
//     the other wave issues back-to-back MFMAs (whose rate does not change) -- a matrix instruction's operand reads go
//     through the vector register ports.  A variant with the two waves half a group out of phase (one multiplies while
//     the other runs the activation) was 2.2x SLOWER per tile than this one (profiles/r03_mlp_pingpong_stamps.txt);
//   * the output accumulators START as residual + b2 (the residual rows are requested at the top of the tile and
//     unpacked just before the first GEMM 2: their latency is behind a whole group of work), so the epilogue is 16 stores.
// The intermediate is rounded to bf16 exactly once, like the two-launch path rounds the tensor it stores.
#include <algorithm>
#include "gemm_stream_kernel.h"

using namespace tfimm_gemm;

namespace {

struct MlpArgs {
  const bf16_t* x;          // [M][C]
  const bf16_t* w1;         // [4C][C] bf16, gamma folded
  const float* b1;          // [4C] beta . W1 + b1
  const bf16_t* w2;         // [C][4C] bf16, K axis in pack.chain_k_order
  const float* b2;          // [C]
  const bf16_t* residual;   // [M][C]
  bf16_t* out;              // [M][C]
  int M, act;
  int res_is_x;             // the shortcut is x itself: its rows are taken from the x tile in LDS
  float eps;
  unsigned x_bytes, w1_bytes, w2_bytes, b1_bytes, b2_bytes, out_bytes;
  int n_tiles;
#ifdef MLP_STAMPS
  long long* stamps;        // tools/probes/mlp_stamps_probe.hip: s_memtime of workgroup 0, waves 0 and 4, around every step
#endif
};

template <int C>
struct MlpGeom {
  static constexpr int BM = 256, NW = 8, HG = 64;          // rows per workgroup, waves, hidden channels per group
  static constexpr int NG = 4 * C / HG;                    // groups
  static constexpr int X_BYTES = BM * C * 2;               // 64 KiB at C = 128
  static constexpr int W1_BYTES = HG * C * 2, W2_BYTES = C * HG * 2;
  static constexpr int TAB_BYTES = 3072;                   // b1 [4C] fp32 (2 KiB) | b2 [C] fp32 in a 1-KiB piece
  static constexpr int OFF_W1 = X_BYTES, OFF_W2 = OFF_W1 + 2 * W1_BYTES, OFF_TAB = OFF_W2 + 2 * W2_BYTES;
  static constexpr int STG_PITCH = 80;                     // staging rows of the output transpose: 64 bytes + 16 of padding
  static constexpr int STG_BYTES = 32 * STG_PITCH;         // per wave: 32 rows x 32 channels
  static constexpr int OFF_STG = OFF_TAB + TAB_BYTES;
  static constexpr int LDS_BYTES = OFF_STG + NW * STG_BYTES;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

typedef __attribute__((ext_vector_type(4))) unsigned int mlp_u32x4;

// GEMM 1 of one hidden group: acc[j] = W1'[group rows j*32 ..][:] . xhat^T, 16 MFMAs.  a[ks] = LDS byte address of this lane's
// fragment (row frow, k-step ks) of the j = 0 half, the j = 1 half is 8 KiB further.  Reads run one batch (2 k-steps) ahead.
__device__ __forceinline__ void mlp_gemm1(f32x16& acc0, f32x16& acc1, const bf16x8 (&xf)[8], const unsigned (&a)[8]) {
  mlp_u32x4 t0, t1, t2, t3, t4, t5, t6, t7;
  asm volatile(
      "ds_read_b128 %2, %18\n\tds_read_b128 %3, %18 offset:8192\n\tds_read_b128 %4, %19\n\tds_read_b128 %5, %19 offset:8192\n\t"
      "ds_read_b128 %6, %20\n\tds_read_b128 %7, %20 offset:8192\n\tds_read_b128 %8, %21\n\tds_read_b128 %9, %21 offset:8192\n\t"
      "s_waitcnt lgkmcnt(4)\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %2, %10, 0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %10, 0\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %4, %11, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %5, %11, %1\n\t"
      "ds_read_b128 %2, %22\n\tds_read_b128 %3, %22 offset:8192\n\tds_read_b128 %4, %23\n\tds_read_b128 %5, %23 offset:8192\n\t"
      "s_waitcnt lgkmcnt(4)\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %6, %12, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %7, %12, %1\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %8, %13, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %9, %13, %1\n\t"
      "ds_read_b128 %6, %24\n\tds_read_b128 %7, %24 offset:8192\n\tds_read_b128 %8, %25\n\tds_read_b128 %9, %25 offset:8192\n\t"
      "s_waitcnt lgkmcnt(4)\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %2, %14, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %14, %1\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %4, %15, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %5, %15, %1\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %6, %16, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %7, %16, %1\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %8, %17, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %9, %17, %1\n\t"
      "s_nop 15\n\ts_nop 7"      // hipcc pads no MFMA -> VALU read hazard behind an asm block: the accumulators are settled when it ends
      : "=&v"(acc0), "=&v"(acc1), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
      : "v"(xf[0]), "v"(xf[1]), "v"(xf[2]), "v"(xf[3]), "v"(xf[4]), "v"(xf[5]), "v"(xf[6]), "v"(xf[7]),
        "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7])
      : "memory");
}

// GEMM 2 of one hidden group: acc[b] += W2'[rows b*32 ..][the group's 64 k, chain order] . h^T, 16 MFMAs.  a[s] = LDS byte address
// of this lane's fragment (row frow, k-step s) of output block 0, block b is 4 KiB * b further.
__device__ __forceinline__ void mlp_gemm2(f32x16 (&acc)[4], const bf16x8 (&hf)[4], const unsigned (&a)[4]) {
  mlp_u32x4 t0, t1, t2, t3, t4, t5, t6, t7;
  asm volatile(
      "ds_read_b128 %4, %16\n\tds_read_b128 %5, %16 offset:4096\n\tds_read_b128 %6, %16 offset:8192\n\tds_read_b128 %7, %16 offset:12288\n\t"
      "ds_read_b128 %8, %17\n\tds_read_b128 %9, %17 offset:4096\n\tds_read_b128 %10, %17 offset:8192\n\tds_read_b128 %11, %17 offset:12288\n\t"
      "s_waitcnt lgkmcnt(4)\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %4, %12, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %5, %12, %1\n\t"
      "v_mfma_f32_32x32x16_bf16 %2, %6, %12, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %7, %12, %3\n\t"
      "ds_read_b128 %4, %18\n\tds_read_b128 %5, %18 offset:4096\n\tds_read_b128 %6, %18 offset:8192\n\tds_read_b128 %7, %18 offset:12288\n\t"
      "s_waitcnt lgkmcnt(4)\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %8, %13, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %9, %13, %1\n\t"
      "v_mfma_f32_32x32x16_bf16 %2, %10, %13, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %11, %13, %3\n\t"
      "ds_read_b128 %8, %19\n\tds_read_b128 %9, %19 offset:4096\n\tds_read_b128 %10, %19 offset:8192\n\tds_read_b128 %11, %19 offset:12288\n\t"
      "s_waitcnt lgkmcnt(4)\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %4, %14, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %5, %14, %1\n\t"
      "v_mfma_f32_32x32x16_bf16 %2, %6, %14, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %7, %14, %3\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %8, %15, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %9, %15, %1\n\t"
      "v_mfma_f32_32x32x16_bf16 %2, %10, %15, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %11, %15, %3\n\t"
      "s_nop 15\n\ts_nop 7"
      : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4),
        "=&v"(t5), "=&v"(t6), "=&v"(t7)
      : "v"(hf[0]), "v"(hf[1]), "v"(hf[2]), "v"(hf[3]), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3])
      : "memory");
}

template <int C>
__global__ void __launch_bounds__(512) mlp_fused_kernel(const MlpArgs p) {
  using G = MlpGeom<C>;
  static_assert(C == 128, "row swizzle and piece mapping are written for 256-byte rows");
  constexpr int BM = G::BM, NW = G::NW, NG = G::NG;
  constexpr int KS1 = C / 16;          // MFMA k-steps of GEMM 1 (8)
  constexpr int NB2 = C / 32;          // output accumulator blocks (4)
  constexpr int XP = BM * C * 2 / 1024 / NW;     // x DMA pieces per wave (8)
  static_assert(KS1 == 8 && NB2 == 4 && XP == 8, "the hand-scheduled GEMMs and the counted waits assume these");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sX = smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fhi = lane >> 5;

  int t_first, t_hi, t_step;
  {
    const int nb = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int q = p.n_tiles >> 3, r = p.n_tiles & 7;
    const int t_lo = xcd * q + (xcd < r ? xcd : r);
    t_hi = t_lo + q + (xcd < r ? 1 : 0);
    t_step = nb >> 3;
    t_first = t_lo + j;
  }
  if (t_first >= t_hi) return;

  const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w1 = make_rsrc(p.w1, p.w1_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w2 = make_rsrc(p.w2, p.w2_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_o = make_rsrc(p.out, p.out_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_r = make_rsrc(p.residual, p.out_bytes);

  // ---- LDS-DMA source mapping (the DMA writes LDS linearly: the swizzle goes on the SOURCE chunk).  Every per-piece offset
  // is derived AT THE REQUEST from one lane constant and wave-uniform terms (the opaque() keeps hipcc from hoisting a
  // dozen precomputed offsets into registers this kernel does not have):
  // 256-byte rows (x, W1): a 1-KiB piece = 4 rows; lane -> (row r4 = lane >> 4, physical chunk pc16 = lane & 15) of piece P reads
  //   logical chunk pc16 ^ (row & 15) = (pc16 ^ r4) ^ ((P & 3) << 2):   offset = P * 1024 + (xlane ^ ((P & 3) << 6))
  // 128-byte rows of W2 (1024-byte pitch): a piece = 8 rows; lane -> (row r8 = lane >> 3, chunk pc8 = lane & 7) of piece P reads
  //   logical chunk pc8 ^ ((row >> 1) & 7) = (pc8 ^ (r8 >> 1)) ^ ((P & 1) << 2):   offset = P * 8192 + (w2lane ^ ((P & 1) << 6))
  const unsigned xlane = (unsigned)((lane >> 4) * 256 + (((lane & 15) ^ (lane >> 4)) * 16));
  const unsigned w2lane = (unsigned)((lane >> 3) * (4 * C * 2) + (((lane & 7) ^ (lane >> 4)) * 16));
  auto opaque = [](unsigned v) __attribute__((always_inline)) { asm volatile("" : "+v"(v)); return v; };
  auto issue_x = [&](int tile, bool valid) __attribute__((always_inline)) {
    const unsigned xl = opaque(xlane);
    const int row0 = tile * BM + (lane >> 4);
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      const int P = wave * XP + j;                                 // piece = rows 4 P .. 4 P + 3 of the tile
      const int m = row0 + P * 4;
      const unsigned off = (valid && m < p.M) ? (unsigned)(tile * BM * C * 2 + P * 1024) + (xl ^ (unsigned)((j & 3) << 6)) : kOobOffset;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(sX + P * 1024), 16, (int)off, 0, 0, 0);
    }
  };
  // W1 rows of hidden group g -> W1 stage st (2 pieces per wave)
  auto issue_w1 = [&](int g, int st, bool valid) __attribute__((always_inline)) {
    char* const s1 = smem + G::OFF_W1 + st * G::W1_BYTES;
    const unsigned xl = opaque(xlane);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int P = wave * 2 + j;                                  // hidden channels 4 P .. 4 P + 3 of the group
      const unsigned off = valid ? (unsigned)(g * G::W1_BYTES + P * 1024) + (xl ^ (unsigned)((P & 3) << 6)) : kOobOffset;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w1, (lds_ptr_t)(s1 + P * 1024), 16, (int)off, 0, 0, 0);
    }
  };
  // the group's 64 k of every W2 row -> W2 stage st (2 pieces per wave)
  auto issue_w2 = [&](int g, int st, bool valid) __attribute__((always_inline)) {
    char* const s2 = smem + G::OFF_W2 + st * G::W2_BYTES;
    const unsigned wl = opaque(w2lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int P = wave * 2 + j;                                  // output channels 8 P .. 8 P + 7
      const unsigned off = valid ? (unsigned)(P * 8 * (4 * C * 2) + g * G::HG * 2) + (wl ^ (unsigned)((P & 1) << 6)) : kOobOffset;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w2, (lds_ptr_t)(s2 + P * 1024), 16, (int)off, 0, 0, 0);
    }
  };
  const ActParams actp = make_act(p.act);

  // LDS byte addresses of this lane's MFMA A fragments: k-step 0 of stage 0; k-step ks is ^ (ks << 5) (the k-step moves the
  // logical chunk by 2 ks, and the swizzle is an XOR on the chunk bits), stage st is + st * 16 KiB
  const unsigned a10 = (unsigned)(size_t)(lds_ptr_t)(smem + G::OFF_W1 + frow * 256 + ((fhi ^ (frow & 15)) * 16));
  const unsigned a20 = (unsigned)(size_t)(lds_ptr_t)(smem + G::OFF_W2 + lds_slot(frow, fhi) * 16);
  const unsigned ax0 = (unsigned)(size_t)(lds_ptr_t)(sX + (wave * 32 + frow) * 256 + ((fhi ^ (frow & 15)) * 16));
  const unsigned tab_addr = (unsigned)(size_t)(lds_ptr_t)(smem + G::OFF_TAB);

  bf16x8 xf[KS1], hf[4];
  f32x16 acc1[2], acc2[NB2];

  auto gemm1 = [&](int g) __attribute__((always_inline)) {
    unsigned a[KS1];
    const unsigned base = opaque(a10 + (unsigned)((g & 1) * G::W1_BYTES));
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) a[ks] = base ^ (unsigned)(ks << 5);
#ifndef MLP_PROBE_NO_GEMM
    mlp_gemm1(acc1[0], acc1[1], xf, a);
#endif
  };
  auto gemm2 = [&](int g) __attribute__((always_inline)) {
    unsigned a[4];
    const unsigned base = opaque(a20 + (unsigned)((g & 1) * G::W2_BYTES));
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = base ^ (unsigned)(s << 5);
#ifndef MLP_PROBE_NO_GEMM
    mlp_gemm2(acc2, hf, a);
#endif
  };
  // V(g): + b1', activation, bf16 -- in the accumulator layout (lane: row frow, channels j*32 + q*8 + fhi*4 + 0..3), packed to
  // the B operand of GEMM 2
  auto vstep = [&](int g) __attribute__((always_inline)) {
    const unsigned ta = opaque(tab_addr) + (unsigned)((g * G::HG + fhi * 4) * 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      mlp_u32x4 tb[4];
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\t"
                   "ds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(tb[0]), "=&v"(tb[1]), "=&v"(tb[2]), "=&v"(tb[3]) : "v"(ta + (unsigned)(j * 128)) : "memory");
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        tfimm_f32x2 v[4];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int q = qq * 2 + h2;
          const float4 b4 = __builtin_bit_cast(float4, tb[q]);
          v[h2 * 2 + 0] = tfimm_f32x2{acc1[j][q * 4 + 0], acc1[j][q * 4 + 1]} + tfimm_f32x2{b4.x, b4.y};
          v[h2 * 2 + 1] = tfimm_f32x2{acc1[j][q * 4 + 2], acc1[j][q * 4 + 3]} + tfimm_f32x2{b4.z, b4.w};
        }
#ifndef MLP_PROBE_NO_ACT
        act8p(v, actp);
#endif
        hf[j * 2 + qq] = __builtin_bit_cast(bf16x8, pack8p(v));
      }
    }
  };
  // residual rows of this wave in the accumulator layout: 8 bytes per lane and accumulator quad (row frow, channels
  // b*32 + q*8 + fhi*4 ..), requested at the top of a tile ...
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  u32x2 rr[NB2 * 4];
  auto row_offset = [&](int m0) __attribute__((always_inline)) {
    const int m = m0 + wave * 32 + frow;
    return (m < p.M) ? (unsigned)((size_t)m * C * 2) + (unsigned)(fhi * 8) : kOobOffset;
  };
  auto request_residual = [&](int m0) __attribute__((always_inline)) {
    if (p.res_is_x) {
      // chunk cc = 4 b + q of this lane's row, its half fhi: address = ar ^ (cc << 4) (the row swizzle is an XOR on the chunk bits)
      const unsigned ar = opaque((unsigned)(size_t)(lds_ptr_t)(sX + (wave * 32 + frow) * 256 + ((frow & 15) << 4) + fhi * 8));
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        unsigned a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = ar ^ (unsigned)((h * 8 + i) << 4);
        asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %9\n\tds_read_b64 %2, %10\n\tds_read_b64 %3, %11\n\t"
                     "ds_read_b64 %4, %12\n\tds_read_b64 %5, %13\n\tds_read_b64 %6, %14\n\tds_read_b64 %7, %15\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(rr[h * 8 + 0]), "=&v"(rr[h * 8 + 1]), "=&v"(rr[h * 8 + 2]), "=&v"(rr[h * 8 + 3]), "=&v"(rr[h * 8 + 4]),
                       "=&v"(rr[h * 8 + 5]), "=&v"(rr[h * 8 + 6]), "=&v"(rr[h * 8 + 7])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
      }
      return;
    }
    const unsigned rowoff = row_offset(m0);
#pragma unroll
    for (int i = 0; i < NB2 * 4; ++i)
      rr[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_r, (int)(rowoff == kOobOffset ? kOobOffset : rowoff + (unsigned)(i * 16)), 0, 0));
  };
  // ... and turned into the initial value of the output accumulators, + b2, just before the first GEMM 2
  auto init_acc2 = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < NB2; ++b) {
      mlp_u32x4 tb[4];
      const unsigned ta = opaque(tab_addr) + 2048u + (unsigned)((b * 32 + fhi * 4) * 4);
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\t"
                   "ds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(tb[0]), "=&v"(tb[1]), "=&v"(tb[2]), "=&v"(tb[3]) : "v"(ta) : "memory");
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b4 = __builtin_bit_cast(float4, tb[q]);
        const u32x2 r = rr[b * 4 + q];
        acc2[b][q * 4 + 0] = __uint_as_float(r[0] << 16) + b4.x;
        acc2[b][q * 4 + 1] = __uint_as_float(r[0] & 0xffff0000u) + b4.y;
        acc2[b][q * 4 + 2] = __uint_as_float(r[1] << 16) + b4.z;
        acc2[b][q * 4 + 3] = __uint_as_float(r[1] & 0xffff0000u) + b4.w;
      }
    }
  };
  // Output: one row per lane in the accumulators, 16-byte row segments in HBM -- transposed through a wave-private staging
  // block, 32 channels (one accumulator block) at a time: 4 x (4 ds_write_b64, 2 ds_read_b128, 2 stores of 16 bytes; a store
  // instruction covers 64 contiguous bytes of 16 rows -- straight from the accumulator layout it would be 8 bytes of 64 rows,
  // and 16 such stores per lane took ~5700 cycles of a ~73000-cycle tile).  All LDS traffic is explicit: the next tile's
  // weights are in flight, and in front of an LDS access it can see hipcc waits for every outstanding DMA.
  auto store_tile = [&](int m0) __attribute__((always_inline)) {
    const unsigned stg = opaque((unsigned)(size_t)(lds_ptr_t)(smem + G::OFF_STG + wave * G::STG_BYTES));
    const unsigned wa = stg + (unsigned)(frow * G::STG_PITCH + fhi * 8);                  // + q * 16
    const unsigned ra = stg + (unsigned)((lane >> 2) * G::STG_PITCH + (lane & 3) * 16);   // + 16 rows for the second read
    const int mrow = m0 + wave * 32 + (lane >> 2);
    const unsigned o0 = (mrow < p.M) ? (unsigned)((size_t)mrow * C * 2) + (unsigned)((lane & 3) * 16) : kOobOffset;
    const unsigned o1 = (mrow + 16 < p.M) ? (unsigned)((size_t)(mrow + 16) * C * 2) + (unsigned)((lane & 3) * 16) : kOobOffset;
#pragma unroll
    for (int b = 0; b < NB2; ++b) {
      u32x2 o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        o[q][0] = pack_bf2(acc2[b][q * 4 + 0], acc2[b][q * 4 + 1]);
        o[q][1] = pack_bf2(acc2[b][q * 4 + 2], acc2[b][q * 4 + 3]);
      }
      mlp_u32x4 v0, v1;
      asm volatile("ds_write_b64 %2, %4\n\tds_write_b64 %2, %5 offset:16\n\tds_write_b64 %2, %6 offset:32\n\t"
                   "ds_write_b64 %2, %7 offset:48\n\ts_waitcnt lgkmcnt(0)\n\t"
                   "ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1280\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1) : "v"(wa), "v"(ra), "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]) : "memory");
      __builtin_amdgcn_raw_buffer_store_b128(v0, rsrc_o, (int)(o0 == kOobOffset ? kOobOffset : o0 + (unsigned)(b * 64)), 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(v1, rsrc_o, (int)(o1 == kOobOffset ? kOobOffset : o1 + (unsigned)(b * 64)), 0, 0);
    }
  };
  // this wave's rows out of the x tile, normalised: xhat = (x - mean) * rstd in bf16 = the B fragments of GEMM 1
  auto normalise = [&]() __attribute__((always_inline)) {
    mlp_u32x4 u[KS1];
    {
      unsigned xa[KS1];
      const unsigned base = opaque(ax0);
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) xa[ks] = base ^ (unsigned)(ks << 5);
      asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
                   "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7])
                   : "v"(xa[0]), "v"(xa[1]), "v"(xa[2]), "v"(xa[3]), "v"(xa[4]), "v"(xa[5]), "v"(xa[6]), "v"(xa[7]) : "memory");
    }
    tfimm_f32x2 s2 = {0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      tfimm_f32x2 f[4];
      unpack8p(__builtin_bit_cast(uint4, u[ks]), f);
      s2 += (f[0] + f[1]) + (f[2] + f[3]);
    }
    float s = s2.x + s2.y;
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.f / (float)C);
    tfimm_f32x2 q2 = {0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      tfimm_f32x2 f[4];
      unpack8p(__builtin_bit_cast(uint4, u[ks]), f);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const tfimm_f32x2 t = f[e] - mean; q2 += t * t; }
    }
    float q = q2.x + q2.y;
    q += __shfl_xor(q, 32, 64);
    const float rstd = rsqrtf(q * (1.f / (float)C) + p.eps);
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      tfimm_f32x2 f[4];
      unpack8p(__builtin_bit_cast(uint4, u[ks]), f);
#pragma unroll
      for (int e = 0; e < 4; ++e) f[e] = (f[e] - mean) * rstd;
      xf[ks] = __builtin_bit_cast(bf16x8, pack8p(f));
    }
  };
#define MLP_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define MLP_BARRIER() tfimm_lds_reuse_barrier()    /* with this wave's LDS reads complete (common.h) */

  // ---- prologue: tables (waves 0-2 one piece each: b1 = 2 pieces, b2 = the valid half of a third), first x tile, W1(0)
  if (wave < 3) {
    const __amdgpu_buffer_rsrc_t rsrc_t = wave < 2 ? make_rsrc(p.b1, p.b1_bytes) : make_rsrc(p.b2, p.b2_bytes);
    const unsigned off = (unsigned)(((wave & 1) * 256 + lane * 4) * 4);     // beyond b2's 512 bytes: out of range -> zeros
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_t, (lds_ptr_t)(smem + G::OFF_TAB + wave * 1024), 16, (int)(wave == 2 ? lane * 16 : off), 0, 0, 0);
  }
  issue_x(t_first, true);
  issue_w1(0, 0, true);
  issue_w2(0, 0, true);

#ifdef MLP_STAMPS
  int n_stamp = 0;
#define MLP_STAMP()                                                                                  \
  do {                                                                                               \
    if (blockIdx.x == 0 && (wave & 3) == 0 && n_stamp < 2048) {                                      \
      const long long t_ = (long long)__builtin_amdgcn_s_memtime();                                  \
      if (lane == 0) p.stamps[(wave >> 2) * 2048 + n_stamp] = t_;                                    \
      ++n_stamp;                                                                                     \
    }                                                                                                \
  } while (0)
#else
#define MLP_STAMP() do {} while (0)
#endif

  // Requests and waits (every wave ITS pieces; the barrier makes them everyone's): the barrier of group g stands behind the last
  // read of ring stage (g + 1) & 1 (group g - 1) and in front of the first read of stage g & 1:
  //     behind barrier g:   W1(g+1), W2(g+1) -> stage (g + 1) & 1   (g = NG - 1: the next tile's group 0);   g = 1: also the next
  //                         tile's x rows (every wave normalised its rows before that barrier)
  //     before barrier g:   this wave's pieces of group g have landed.  Newer than them are only the 8 x pieces (g = 2) and,
  //                         at the top of a tile, the 8 stores of the previous tile
  bool first_tile = true;
  for (int tile = t_first; tile < t_hi; tile += t_step) {
    const int m0 = tile * BM;
    const bool has_next = tile + t_step < t_hi;
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
      MLP_STAMP();
      if (g == 0) {
        if (first_tile) MLP_WAIT(0); else MLP_WAIT(8);
      } else if (g == 2) {
        MLP_WAIT(8);
      } else {
        MLP_WAIT(0);
      }
      MLP_BARRIER();
      if (g + 1 < NG) {
        issue_w1(g + 1, (g + 1) & 1, true);
        issue_w2(g + 1, (g + 1) & 1, true);
      } else {
        issue_w1(0, 0, has_next);
        issue_w2(0, 0, has_next);
      }
      if (g == 1) issue_x(tile + t_step, has_next);
      MLP_STAMP();
      if (g == 0) {
        request_residual(m0);
        normalise();
      }
      gemm1(g);
      MLP_STAMP();
      vstep(g);
      MLP_STAMP();
      if (g == 0) init_acc2();
      gemm2(g);
    }
    MLP_STAMP();
    store_tile(m0);
    first_tile = false;
  }
  MLP_WAIT(0);
#undef MLP_WAIT
#undef MLP_BARRIER
#undef MLP_STAMP
}

int mlp_num_cu() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

}  // namespace

#ifdef MLP_STAMPS
long long* tfimm_mlp_stamps = nullptr;
#endif

extern "C" int tfimm_hip_mlp_fused(const tfimm_mlp_desc* dp, void* stream) {
  if (!dp) TFIMM_FAIL(TFIMM_EINVAL, "mlp_fused: null descriptor");
  const tfimm_mlp_desc& d = *dp;
  if (!d.x || !d.w1 || !d.b1 || !d.w2 || !d.b2 || !d.residual || !d.out) TFIMM_FAIL(TFIMM_EINVAL, "mlp_fused: null pointer");
  if (d.M <= 0) TFIMM_FAIL(TFIMM_EINVAL, "mlp_fused: M = %lld", (long long)d.M);
  if (d.C != 128 || d.hidden != 4 * d.C) TFIMM_FAIL(TFIMM_EUNSUP, "mlp_fused: built for C = 128, hidden = 512 (got %d, %d)", d.C, d.hidden);
  if (((uintptr_t)d.x | (uintptr_t)d.w1 | (uintptr_t)d.b1 | (uintptr_t)d.w2 | (uintptr_t)d.b2 | (uintptr_t)d.residual |
       (uintptr_t)d.out) & 15)
    TFIMM_FAIL(TFIMM_EINVAL, "mlp_fused: pointers must be 16-byte aligned");
  // a buffer descriptor addresses 2 GiB: larger tensors (Swin-B stage 1 from batch 2675 on) run as chunks of whole 256-row tiles
  // -- rows are independent.  TFIMM_MLP_LIMIT (bytes) lowers the threshold: a test hook, read once.
  static const int64_t limit = getenv("TFIMM_MLP_LIMIT") ? atoll(getenv("TFIMM_MLP_LIMIT")) : 0x7fffff00LL;
  const int64_t act_bytes = d.M * d.C * 2;
  if (act_bytes > limit) {
    const int64_t rows_max = limit / (d.C * 2) / 256 * 256;
    if (rows_max <= 0) TFIMM_FAIL(TFIMM_EINVAL, "mlp_fused: TFIMM_MLP_LIMIT below one tile");
    for (int64_t m = 0; m < d.M; m += rows_max) {
      tfimm_mlp_desc sub = d;
      const int64_t off = m * d.C * 2;
      sub.x = (const char*)d.x + off;
      sub.residual = (const char*)d.residual + off;
      sub.out = (char*)d.out + off;
      sub.M = std::min<int64_t>(rows_max, d.M - m);
      if (int rc = tfimm_hip_mlp_fused(&sub, stream)) return rc;
    }
    return 0;
  }
  MlpArgs a;
  a.x = (const bf16_t*)d.x; a.w1 = (const bf16_t*)d.w1; a.b1 = d.b1; a.w2 = (const bf16_t*)d.w2; a.b2 = d.b2;
  a.residual = (const bf16_t*)d.residual; a.out = (bf16_t*)d.out;
  a.M = (int)d.M; a.act = d.act; a.eps = d.eps;
  a.res_is_x = d.residual == d.x;
  a.x_bytes = (unsigned)act_bytes; a.out_bytes = (unsigned)act_bytes;
  a.w1_bytes = (unsigned)((size_t)d.hidden * d.C * 2); a.w2_bytes = (unsigned)((size_t)d.C * d.hidden * 2);
  a.b1_bytes = (unsigned)(d.hidden * 4); a.b2_bytes = (unsigned)(d.C * 4);
  a.n_tiles = (int)cdiv64(d.M, 256);
#ifdef MLP_STAMPS
  a.stamps = tfimm_mlp_stamps;
#endif
  constexpr int lds = MlpGeom<128>::LDS_BYTES;
  static tfimm_once_t ready;
  if (ready.need()) {
    TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)mlp_fused_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    ready.mark();
  }
  int64_t grid = ((int64_t)mlp_num_cu() + 7) / 8 * 8;
  const int64_t need = ((int64_t)a.n_tiles + 7) / 8 * 8;
  if (grid > need) grid = need;
  TFIMM_LAUNCH(mlp_fused_kernel<128>, dim3((unsigned)grid), dim3(512), (size_t)lds, (hipStream_t)stream, a);
  return 0;
}
