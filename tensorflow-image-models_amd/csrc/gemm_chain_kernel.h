// Two chained bf16 MFMA GEMMs in one persistent kernel (gfx950): the tail of a ResNet bottleneck block,
//
//     mid = act1( conv3x3(x) + b1 )                       GEMM 1: 3x3 / stride 1 / pad 1, 64 -> 64 channels
//     out = act2( mid . W2^T + b2 + residual )            GEMM 2: 1x1 convolution, K2 = 64, N2 = 64 * steps
//
// (reference Bottleneck.call, resnet.py:273-290: pad2 / conv2 / bn2 / act2, conv3 / bn3, += shortcut, act3 -- BatchNorm
// folded into W and b on the host).  As two launches the 64-channel intermediate is written to HBM by the first kernel and
// read back by the second, and the first one re-gathers every input pixel nine times (once per filter tap) from L2 into
// LDS -- what bounds such a low-intensity tile is the LDS fill rate (about 20 bytes per clock and CU: measured with the
// arithmetic removed), not the MFMA pipe.  Here
//
//   * INPUT STRIP: at stride 1 the pixels a 256-pixel tile needs are ONE contiguous run of the flattened NHWC tensor,
//     [m0 - W - 1, m0 + 128 + W + 1).  It is brought into LDS once per tile (31 KiB at W = 56) and the A fragment of tap
//     (ky, kx) is read from it by address arithmetic (row + ky W + kx); taps that fall outside the image are zeroed in the
//     register.  Per tile the LDS-DMA moves the strip + 72 KiB of W1 + 32 KiB of W2 instead of 9 activation k-tiles
//     (9 x 24 KiB);
//   * REGISTER CHAIN: a wave owns 32 output pixels and ALL 64 intermediate channels of them.  With the MFMA operands
//     swapped (D[n][m] = W . X^T) its GEMM-1 accumulators hold, per lane, one pixel (lane & 31) and the channels
//     32 j + 8 q + 4 (lane >> 5) + 0..3 -- the shape of a B operand of v_mfma_f32_32x32x16_bf16 (one column = one pixel,
//     8 consecutive k per lane) up to the ORDER of the k values.  A reduction does not care about that order as long as
//     both operands agree, so the host stores W2 with its K axis permuted to the order the accumulators come in
//     (pack.chain_k_order) and the packed bf16 accumulators feed GEMM 2 directly: no LDS round trip, no barrier between
//     the two GEMMs, the intermediate rounded to bf16 once exactly as the two-launch path does;
//   * persistent workgroups of FOUR waves (a 128-pixel tile, 80 KiB of LDS) so that TWO of them share a CU: within one
//     workgroup the phases of a tile -- tap steps, slice steps, the HBM-heavy epilogue -- run one after the other, and
//     measured with one 8-wave workgroup per CU their times simply added up; two independent workgroups drift apart and
//     fill each other's gaps.  XCD-contiguous tile ranges; ONE flattened step stream per workgroup -- 9 W1 taps then
//     N2 / 64 slices of W2 per tile, the next tile's right behind -- through a 2-stage LDS ring, one step ahead (deeper
//     rings and narrower epilogue passes were measured: 217 us at depth 2 / 64 columns, 223-227 us at depth 3-4 / 32), with
//     COUNTED vmcnt waits (VMEM operations retire in issue order: a wait names how many younger ones may stay in
//     flight); the next tile's strip is requested when the first GEMM-2 step begins (every wave is done reading the
//     current one);
//   * epilogue 2 is the stream kernel's vector epilogue (per-wave fp32 staging block, row-contiguous 16-byte residual
//     loads / stores); a step's residual rows and bias are requested before that step's LDS-DMA so that waiting for them
//     never waits for the DMA.
#pragma once
#include "gemm_stream_kernel.h"

namespace tfimm_gemm {

struct ChainArgs {
  const bf16_t* x;          // NHWC input [B][H][W][64]
  const bf16_t* w1;         // [64][576] bf16, K order (ky, kx, ci)
  const float* b1;          // [64]
  const bf16_t* w2;         // [N2][ldw2] bf16, K axis in pack.chain_k_order
  const float* b2;          // [N2]
  const bf16_t* residual;   // [M][ldr] or null
  const bf16_t* ds_x;       // DS flavour: the block input [M][64] whose 1x1 convolution IS the shortcut (resnet.py:315-330)
  const uint4* ds_w;        // DS flavour: its weights as MFMA fragments [N2 / 32][4][64 lanes][8] (pack.pack_chain_ds)
  bf16_t* out;              // [M][ldc]
  int M, N2;
  int B, H, W;
  int ldw1, ldw2, ldr, ldc;
  int act1, act2;           // act2 is applied after the residual add
  unsigned x_bytes, w1_bytes, w2_bytes, out_bytes, res_bytes, ds_bytes;
  int n_tiles;              // ceil(M / 128)
  int dbg;                  // TFIMM_CHAIN_DBG ablation switches (0 in production)
};

typedef void (*gemm_chain_fn)(const ChainArgs);

struct ChainGeom {
  static constexpr int BM = 128, NW = 4, C1 = 64, BN2 = 64;
  static constexpr int STRIP_ROWS = 256;                       // >= BM + 2 W + 2  ->  W <= 63
  static constexpr int STRIP_BYTES = STRIP_ROWS * 128;         // 32 KiB
  static constexpr int STAGE = 64 * 128;                       // 8 KiB: one W1 tap [64 x 64] or one W2 slice [64 x 64]
#ifndef TFIMM_CHAIN_NST
#define TFIMM_CHAIN_NST 2
#endif
#ifndef TFIMM_CHAIN_WTN
#define TFIMM_CHAIN_WTN 64
#endif
  static constexpr int NST = TFIMM_CHAIN_NST;                  // ring depth: weights are requested NST - 1 steps ahead
  static constexpr int WTN = TFIMM_CHAIN_WTN;                  // columns per epilogue pass
  static constexpr int EPI_WAVE = 32 * WTN * 4;                // fp32 staging block of one wave: 32 pixels x WTN channels
  static constexpr int LDS_BYTES = STRIP_BYTES + NST * STAGE + NW * EPI_WAVE;
  static_assert(LDS_BYTES <= 80 * 1024, "two workgroups must fit one CU's 160 KiB of LDS");
};

template <int N>
__device__ __forceinline__ void chain_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int I>
struct ChainIdx {
  static constexpr int value = I;
};

// NS2 = N2 / 128 GEMM-2 steps per tile.  Every VMEM operation sits in straight-line code (the step bodies are instantiated
// per step index), so hipcc's own vmcnt bookkeeping for the residual / bias loads stays exact next to the LDS-DMA.
// ACT >= 0: both activations are that TFIMM_ACT_* (ResNet: relu) with their parameters folded into the instructions -- the
// kernel sits at the scalar-register limit and run-time activation parameters are another 14 SGPRs; ACT < 0: p.act1 / p.act2
// DS: the shortcut of the block is a 1x1 convolution of the 64-channel block input (first block of a stage): instead of reading
// its 256-channel result as the residual, the four extra k-steps  W_ds . x0  accumulate into the SAME accumulators as GEMM 2
// (operands straight from global memory into registers: 16 bytes of x0 per lane and k-step once per tile, the weight fragments
// per slice), so that launch and its tensor disappear; its folded-BN shift is added to b2 on the host.
template <int NS2, int ACT = -1, bool DS = false>
__global__ void __launch_bounds__(256, 2) gemm_chain_kernel(const ChainArgs p) {
  using G = ChainGeom;
  constexpr int BM = G::BM, NW = G::NW, STAGE = G::STAGE, BN2 = G::BN2, NST = G::NST;
  constexpr int TN1 = 2;                           // GEMM-1 accumulator blocks per wave (32 pixels x 64)
  constexpr int STRIP_INSTR = G::STRIP_ROWS / 8 / NW;   // strip DMA pieces per wave (8)
  constexpr int B1_INSTR = 64 / 8 / NW;            // W1 pieces per wave and tap (2)
  constexpr int B2_INSTR = BN2 / 8 / NW;           // W2 pieces per wave and GEMM-2 step (2)
  constexpr int KS2 = 4;                           // MFMA k-steps of GEMM 2 (K2 = 64)
  constexpr int TN2 = BN2 / 32;                    // GEMM-2 accumulator blocks per wave and step (2)
  constexpr int NK1 = 9;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sRing = smem + G::STRIP_BYTES;
  float* const sEpi = reinterpret_cast<float*>(smem + G::STRIP_BYTES + NST * STAGE);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
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
  const __amdgpu_buffer_rsrc_t rsrc_r = DS ? make_rsrc(p.ds_x, p.ds_bytes) : make_rsrc(p.residual, p.res_bytes);

  static_assert(NS2 >= 2, "the counted waits below are derived for at least two GEMM-2 steps");
  constexpr int LOOK = NST - 1;
  const int lrow = lane >> 3, lpc = lane & 7;

  // ---- LDS-DMA source offsets of this lane (the DMA writes LDS linearly, so the XOR swizzle of the fragment reads
  //      goes on the SOURCE chunk)
  unsigned b1_off[B1_INSTR], b2_off[B2_INSTR];
  int strip_chunk[STRIP_INSTR];
#pragma unroll
  for (int j = 0; j < B1_INSTR; ++j) {
    const int r = (wave * B1_INSTR + j) * 8 + lrow;             // W1 row = intermediate channel
    b1_off[j] = (unsigned)(((size_t)r * p.ldw1 + (lpc ^ ((r >> 1) & 7)) * 8) * 2);
  }
#pragma unroll
  for (int j = 0; j < B2_INSTR; ++j) {
    const int r = (wave * B2_INSTR + j) * 8 + lrow;             // row within the 128-row slice
    b2_off[j] = (unsigned)(((size_t)r * p.ldw2 + (lpc ^ ((r >> 1) & 7)) * 8) * 2);
  }
#pragma unroll
  for (int j = 0; j < STRIP_INSTR; ++j) {
    const int r = (wave * STRIP_INSTR + j) * 8 + lrow;
    strip_chunk[j] = (lpc ^ ((r >> 1) & 7)) * 16;               // byte offset of the chunk this lane fetches
  }
  const int halo = p.W + 1;
  const int n_pix = p.B * p.H * p.W;
  auto issue_strip = [&](int tile, bool valid) __attribute__((always_inline)) {
    const int s0 = tile * BM - halo;                            // first pixel of the strip (may be negative)
#pragma unroll
    for (int j = 0; j < STRIP_INSTR; ++j) {
      const int r = (wave * STRIP_INSTR + j) * 8 + lrow;
      const int s = s0 + r;
      const bool ok = valid && s >= 0 && s < n_pix;
      const unsigned off = ok ? (unsigned)s * 128u + (unsigned)strip_chunk[j] : kOobOffset;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(smem + (wave * STRIP_INSTR + j) * 1024), 16, (int)off, 0, 0, 0);
    }
  };
  // Weight step S of the tile being multiplied (S >= NK1 + NS2: a tap of this workgroup's NEXT tile) -> ring stage
  // `iss_stage`.  S is a compile-time index: no VMEM operation inside a branch.
  int iss_stage = 0;
  auto issue_step = [&](auto sc, int tile) __attribute__((always_inline)) {
    constexpr int S = decltype(sc)::value;
    constexpr int SL = S >= NK1 + NS2 ? S - NK1 - NS2 : S;          // index within its own tile's program
    const bool valid = S >= NK1 + NS2 ? (tile + t_step < t_hi) : true;
    char* const sa = sRing + iss_stage * STAGE;
    if constexpr (SL < NK1) {
#pragma unroll
      for (int j = 0; j < B1_INSTR; ++j) {
        const unsigned off = valid ? b1_off[j] : kOobOffset;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w1, (lds_ptr_t)(sa + (wave * B1_INSTR + j) * 1024), 16, (int)off, SL * 128, 0, 0);
      }
    } else {
      const int soff = (int)((size_t)(SL - NK1) * BN2 * p.ldw2 * 2);
#pragma unroll
      for (int j = 0; j < B2_INSTR; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w2, (lds_ptr_t)(sa + (wave * B2_INSTR + j) * 1024), 16, (int)b2_off[j], soff, 0, 0);
    }
    iss_stage = iss_stage == NST - 1 ? 0 : iss_stage + 1;
  };

  // ---- epilogue-2 geometry (gemm_stream_kernel.h with a 32-row x 32-column wave block: one accumulator block per pass)
  constexpr int WTN = G::WTN, LPR = WTN / 8, RPI = 64 / LPR, ITS = 32 / RPI;  // lanes per row, rows per iteration, iterations
  constexpr int PASSES = BN2 / WTN;                                            // epilogue passes per GEMM-2 step
  constexpr int BPP = WTN / 32;                                                // accumulator blocks per pass
  constexpr int STORES = PASSES * ITS;                                         // store instructions per wave and GEMM-2 step (4)
  constexpr int DS_X = DS ? 4 : 0;                                             // x0 fragment loads per wave and tile (behind tap 8)
  constexpr int DS_W = DS ? 8 : 0;                                             // shortcut weight fragment loads per wave and slice
  constexpr int EPI_LOADS = (DS ? 0 : PASSES * ITS) + PASSES * 2;              // residual + bias loads per wave and step
  auto epi_slot = [](int row, int slot) -> int { return WTN == 64 ? (slot ^ (row & 15)) : (slot ^ ((row >> 1) & 7)); };
  const ActParams act1p = make_act(ACT >= 0 ? ACT : p.act1), act2p = make_act(ACT >= 0 ? ACT : p.act2);
  const int e_row = lane / LPR, e_c8 = lane % LPR;
  const bool has_res = !DS && p.residual != nullptr;
  float* const sEw = sEpi + wave * (G::EPI_WAVE / 4);

  // VMEM operations a step issues, in issue order (taps: their weight DMA; tap 8 first requests slice 0's residual / bias;
  // a slice: residual / bias of the NEXT slice, weight DMA, the next tile's strip behind slice 0, its own stores).
  // VMEM retires in issue order, so "step s's weights have landed" == "at most N(s) operations outstanding" with N(s) =
  // everything issued behind that DMA: the rest of step s - 3 and all of steps s - 2 and s - 1.
  struct Ops {
    static constexpr int all(int r) {           // r = position in the tile program
      if (DS && r == NK1 - 2) return B1_INSTR + TN1 * 4;      // DS: the GEMM-1 bias quads are fetched here, per tile
      if (r < NK1 - 1) return B1_INSTR;
      if (r == NK1 - 1) return EPI_LOADS + DS_X + B1_INSTR;
      const int u = r - NK1;
      return (u < NS2 - 1 ? EPI_LOADS : 0) + DS_W + B2_INSTR + (u == 0 ? STRIP_INSTR : 0) + STORES;
    }
    static constexpr int behind_dma(int r) {    // what step r issues after its weight DMA
      if (r < NK1) return 0;
      return (r - NK1 == 0 ? STRIP_INSTR : 0) + STORES;
    }
    static constexpr int wait_for(int r) {
      constexpr int T = NK1 + NS2;
      int n = behind_dma((r + T - LOOK) % T);
      for (int i = 1; i < LOOK; ++i) n += all((r + T - i) % T);
      return n;
    }
  };
  static_assert(B1_INSTR == B2_INSTR, "every weight step is two pieces per wave");
  static_assert(Ops::wait_for(NK1 + NS2 - 1) < 64, "vmcnt is a 6-bit counter");

  // bias of GEMM 1 in the accumulator layout: lane owns channels j*32 + q*8 + fhi*4 .. +3
  // (DS flavour: these 32 registers are what pushes it over the budget while GEMM 2 and the shortcut fragments are live, so it
  // fetches them per tile instead, behind tap 7 -- eight L2 hits that are back long before epilogue 1)
  f32x4 bias1[TN1][4];
  auto load_bias1 = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < TN1; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4*>(p.b1 + j * 32 + q * 8 + fhi * 4);
        bias1[j][q] = f32x4{b4.x, b4.y, b4.z, b4.w};
      }
  };
  if constexpr (!DS) {
    load_bias1();
    // their values must be in registers before the counted waits below start leaving operations in flight
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias1[0][0]), "+v"(bias1[0][3]), "+v"(bias1[TN1 - 1][0]), "+v"(bias1[TN1 - 1][3])::"memory");
  }

  if (TFIMM_PROBE(p.dbg) & 8) {
    const int n = ((blockIdx.x >> 3) & 7) * (TFIMM_PROBE(p.dbg) >> 8);
    for (int i = 0; i < n; ++i) asm volatile("s_sleep 8");
  }
  // ---- prime the pipeline: strip of the first tile, weight steps 0..2
  issue_strip(t_first, true);
  issue_step(ChainIdx<0>{}, t_first);
  if constexpr (LOOK > 1) issue_step(ChainIdx<1>{}, t_first);
  if constexpr (LOOK > 2) issue_step(ChainIdx<2>{}, t_first);
  static_assert(LOOK >= 1 && LOOK <= 3, "prologue written for up to three steps of lookahead");
  int cur = 0;
  bool first = true;        // no epilogue stores are in flight yet
  auto rotate = [&]() __attribute__((always_inline)) { cur = cur == NST - 1 ? 0 : cur + 1; };

  for (int tile = t_first; tile < t_hi; tile += t_step) {
    const int m0 = tile * BM;

    // validity of the 3 x 3 taps for this lane's pixel (zero padding; the strip holds whatever precedes / follows in memory)
    unsigned vmask;
    {
      const int m = m0 + wave * 32 + frow;
      const int hw = p.H * p.W;
      const int rem = m % hw;
      const int oy = rem / p.W, ox = rem - oy * p.W;
      const unsigned vy = (oy > 0 ? 1u : 0u) | 2u | (oy + 1 < p.H ? 4u : 0u);
      const unsigned vx = (ox > 0 ? 1u : 0u) | 2u | (ox + 1 < p.W ? 4u : 0u);
      vmask = 0;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
        if ((vy >> ky) & 1) vmask |= vx << (3 * ky);
      if (m >= p.M) vmask = 0;
    }

    // =========================== GEMM 1: 32 pixels x 64 channels per wave, one filter tap per step ===========================
    f32x16 acc1[TN1];
#pragma unroll
    for (int j = 0; j < TN1; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc1[j][e] = 0.f;
    // residual rows + bias of GEMM-2 slice u, requested one step before that slice runs (behind tap 8 for slice 0) and
    // ahead of that step's LDS-DMA: two register sets, slice u uses set u & 1
    uint4 rres[2][PASSES][ITS];
    u32x4 dsx[4];                                      // DS: x0 fragments of this tile
    float4 braw[2][PASSES][2];
    const int e_m = m0 + wave * 32 + e_row;            // output row at iteration 0
    const int em = e_m < p.M ? e_m : p.M;              // clamp: offsets stay inside 32 bits; rows >= M fall off the descriptor
    const unsigned ldc2 = (unsigned)p.ldc * 2u, ldr2 = (unsigned)p.ldr * 2u;
    auto load_epi = [&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      const unsigned res_off0 = (unsigned)(((size_t)em * p.ldr + u * BN2 + e_c8 * 8) * 2);
#pragma unroll
      for (int g = 0; g < PASSES; ++g) {
        if constexpr (!DS) {
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          const unsigned off = res_off0 + (unsigned)(it * RPI) * ldr2 + (unsigned)(g * WTN * 2);
          rres[u & 1][g][it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, (int)((TFIMM_PROBE(p.dbg) & 4) ? kOobOffset : off), 0, 0));
        }
        }
        braw[u & 1][g][0] = *reinterpret_cast<const float4*>(p.b2 + u * BN2 + g * WTN + e_c8 * 8);
        braw[u & 1][g][1] = *reinterpret_cast<const float4*>(p.b2 + u * BN2 + g * WTN + e_c8 * 8 + 4);
      }
    };
    auto tap_step = [&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      // (the first tile has only the prologue's two younger weight steps behind taps 0..2)
      if constexpr (k < LOOK) { if (first) chain_wait_vm<(LOOK - 1) * B1_INSTR>(); else chain_wait_vm<Ops::wait_for(k)>(); }
      else chain_wait_vm<Ops::wait_for(k)>();
      tfimm_lds_reuse_barrier();       // (this wave's reads of the stage about to be refilled are complete: common.h)
      if constexpr (DS && k == NK1 - 2) { load_bias1(); __builtin_amdgcn_sched_barrier(0); }
      if constexpr (k == NK1 - 1) {
        load_epi(ChainIdx<0>{});
        if constexpr (DS) {
          // this lane's pixel (accumulator layout: lane & 31), channels 16 t + 8 fhi .. + 7 of the block input: the B operand
          // of the shortcut's k-step t.  Rows >= M fall off the descriptor (zeros).
          const unsigned xo = (unsigned)(m0 + wave * 32 + frow) * 128u + (unsigned)fhi * 16u;
#pragma unroll
          for (int t = 0; t < 4; ++t) dsx[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, (int)(xo + t * 32), 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      issue_step(ChainIdx<k + LOOK>{}, tile);
      if constexpr (k == NK1 - 1) __builtin_amdgcn_sched_barrier(0);
      constexpr int ky = k / 3, kx = k % 3;
      const bool ok = (vmask >> k) & 1;
      const uint4* sB = reinterpret_cast<const uint4*>(sRing + cur * STAGE);
      const int arow = wave * 32 + frow + ky * p.W + kx;        // strip row of tap (ky, kx) of this lane's pixel
      // The four A fragments of this tap come out of the strip with explicit ds_reads: before an LDS read it can see,
      // hipcc waits for every LDS-DMA that may alias it -- the next tile's strip request would drain the whole queue here.
      u32x4 ua[4];
      {
        unsigned ra[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ra[ks] = (unsigned)(size_t)(lds_ptr_t)(smem + lds_slot(arow, ks * 2 + fhi) * 16);
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(ua[0]), "=&v"(ua[1]), "=&v"(ua[2]), "=&v"(ua[3])
                     : "v"(ra[0]), "v"(ra[1]), "v"(ra[2]), "v"(ra[3]) : "memory");
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint4 um = sel4(ok, __builtin_bit_cast(uint4, ua[ks]));
        const bf16x8 fa = __builtin_bit_cast(bf16x8, um);
        bf16x8 fb[TN1];
#pragma unroll
        for (int j = 0; j < TN1; ++j) fb[j] = __builtin_bit_cast(bf16x8, sB[lds_slot(j * 32 + frow, ks * 2 + fhi)]);
#pragma unroll
        for (int j = 0; j < TN1; ++j)
          if (!(TFIMM_PROBE(p.dbg) & 2)) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa, acc1[j], 0, 0, 0);
      }
      rotate();
    };
    tap_step(ChainIdx<0>{}); tap_step(ChainIdx<1>{}); tap_step(ChainIdx<2>{}); tap_step(ChainIdx<3>{}); tap_step(ChainIdx<4>{});
    tap_step(ChainIdx<5>{}); tap_step(ChainIdx<6>{}); tap_step(ChainIdx<7>{}); tap_step(ChainIdx<8>{});
    // ---- epilogue 1 (registers only): + b1, act1, bf16.  bmid[t] is the B operand of GEMM-2 k-step t: this lane's
    // quads (j = t / 2, q = 2 (t % 2)) and (j, q + 1), i.e. channels 32 j + 16 (t % 2) + {0, 8} + 4 fhi + 0..3 --
    // the order pack.chain_k_order gives W2's K axis.
    bf16x8 bmid[KS2];
#pragma unroll
    for (int j = 0; j < TN1; ++j)
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        tfimm_f32x2 v[4];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int q = q2 * 2 + h2;
          const f32x4 b4 = bias1[j][q];
          v[h2 * 2 + 0] = tfimm_f32x2{acc1[j][q * 4 + 0] + b4[0], acc1[j][q * 4 + 1] + b4[1]};
          v[h2 * 2 + 1] = tfimm_f32x2{acc1[j][q * 4 + 2] + b4[2], acc1[j][q * 4 + 3] + b4[3]};
        }
        act8p(v, act1p);
        bmid[j * 2 + q2] = __builtin_bit_cast(bf16x8, pack8p(v));
      }

    // =========================== GEMM 2: 32 pixels x 64 channels per wave and step ===========================
    auto slice_step = [&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      constexpr int n0 = u * BN2;
      const unsigned out_off0 = (unsigned)(((size_t)em * p.ldc + n0 + e_c8 * 8) * 2);
      chain_wait_vm<Ops::wait_for(NK1 + u)>();
      tfimm_lds_reuse_barrier();       // (this wave's reads of the stage about to be refilled are complete: common.h)
      if constexpr (u + 1 < NS2) load_epi(ChainIdx<u + 1>{});
      u32x4 wds[TN2][4];
      if constexpr (DS) {
#pragma unroll
        for (int j = 0; j < TN2; ++j)
#pragma unroll
          for (int t = 0; t < 4; ++t) wds[j][t] = __builtin_bit_cast(u32x4, p.ds_w[(size_t)((u * TN2 + j) * 4 + t) * 64 + lane]);
      }
      __builtin_amdgcn_sched_barrier(0);      // keep every one of those loads ahead of the LDS-DMA in issue order
      issue_step(ChainIdx<NK1 + u + LOOK>{}, tile);
      if constexpr (u == 0) issue_strip(tile + t_step, tile + t_step < t_hi);   // every wave has passed GEMM 1: the strip is free
      __builtin_amdgcn_sched_barrier(0);

      f32x16 acc[TN2];
#pragma unroll
      for (int j = 0; j < TN2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
      {
        const uint4* sW = reinterpret_cast<const uint4*>(sRing + cur * STAGE);
#pragma unroll
        for (int t = 0; t < KS2; ++t) {
          bf16x8 fw[TN2];
#pragma unroll
          for (int j = 0; j < TN2; ++j) fw[j] = __builtin_bit_cast(bf16x8, sW[lds_slot(j * 32 + frow, t * 2 + fhi)]);
#pragma unroll
          for (int j = 0; j < TN2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[j], bmid[t], acc[j], 0, 0, 0);
        }
      }
      if constexpr (DS) {     // + W_ds . x0: the shortcut convolution, four more k-steps into the same accumulators
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < TN2; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wds[j][t]), __builtin_bit_cast(bf16x8, dsx[t]),
                                                             acc[j], 0, 0, 0);
      }
      rotate();

      // ---- epilogue 2 (per wave; dedicated staging block, so it overlaps the other waves' work and the prefetch)
      const bool skip_epi = (TFIMM_PROBE(p.dbg) & 1) != 0;
#pragma unroll
      for (int g = 0; g < PASSES; ++g) {
        // hipcc pads no hazard in front of an asm statement: an MFMA result needs up to 18 wait states before a DS
        // instruction may read it -- naming the accumulators orders the pad behind the MFMAs that produce them
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[g * BPP]), "+v"(acc[g * BPP + BPP - 1]));
#pragma unroll
        for (int j = 0; j < BPP; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int slot = j * 8 + q * 2 + fhi;
            const f32x16& a16 = acc[g * BPP + j];
            const f32x4 v = {a16[q * 4 + 0], a16[q * 4 + 1], a16[q * 4 + 2], a16[q * 4 + 3]};
            const unsigned addr = (unsigned)(size_t)(lds_ptr_t)(&sEw[frow * WTN + epi_slot(frow, slot) * 4]);
            asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
          }
        const float4 bq0 = braw[u & 1][g][0], bq1 = braw[u & 1][g][1];
        const tfimm_f32x2 bias2[4] = {{bq0.x, bq0.y}, {bq0.z, bq0.w}, {bq1.x, bq1.y}, {bq1.z, bq1.w}};
#pragma unroll
        for (int ip = 0; ip < ITS; ip += 2) {
        f32x4 st[4];
        {
          unsigned ra[4];
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            const int pr = (ip + w) * RPI + e_row;
            ra[2 * w] = (unsigned)(size_t)(lds_ptr_t)(&sEw[pr * WTN + epi_slot(pr, 2 * e_c8) * 4]);
            ra[2 * w + 1] = (unsigned)(size_t)(lds_ptr_t)(&sEw[pr * WTN + epi_slot(pr, 2 * e_c8 + 1) * 4]);
          }
          asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(st[0]), "=&v"(st[1]), "=&v"(st[2]), "=&v"(st[3])
                       : "v"(ra[0]), "v"(ra[1]), "v"(ra[2]), "v"(ra[3]) : "memory");
        }
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const int it = ip + w;
          const f32x4 lo = st[2 * w], hi = st[2 * w + 1];
          tfimm_f32x2 v[4] = {{lo[0], lo[1]}, {lo[2], lo[3]}, {hi[0], hi[1]}, {hi[2], hi[3]}};
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bias2[e];
          if (has_res) {
            asm volatile("");
            tfimm_f32x2 r2[4];
            unpack8p(rres[u & 1][g][it], r2);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r2[e];
          }
          act8p(v, act2p);
          const unsigned off = out_off0 + (unsigned)(it * RPI) * ldc2 + (unsigned)(g * WTN * 2);
          // every wave issues all STORES stores of a step (rows >= M fall off the descriptor): the counted waits rely on it
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pack8p(v)), rsrc_o,
                                                 (int)(skip_epi ? kOobOffset : off), 0, 0);
        }
        }
      }
    };
    slice_step(ChainIdx<0>{});
    slice_step(ChainIdx<1>{});
    if constexpr (NS2 > 2) slice_step(ChainIdx<2>{});
    if constexpr (NS2 > 3) slice_step(ChainIdx<3>{});
    if constexpr (NS2 > 4) { slice_step(ChainIdx<4>{}); slice_step(ChainIdx<5>{}); slice_step(ChainIdx<6>{}); slice_step(ChainIdx<7>{}); }
    static_assert(NS2 == 2 || NS2 == 3 || NS2 == 4 || NS2 == 8, "instantiate the slice steps of this NS2");
    first = false;
  }
  // the trailing (all out-of-range) prefetches must have landed before this workgroup's LDS is released
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace tfimm_gemm
