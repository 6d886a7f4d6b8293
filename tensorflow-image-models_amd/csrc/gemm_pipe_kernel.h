// bf16 MFMA GEMM / implicit-GEMM convolution, 256x256 tile, DEEP-RING schedule (gfx950).
//
// Same operands, persistent tile walk, flattened (tile, k-tile) DMA cursor and vector epilogue as
// gemm_stream_kernel.h.  What changes is the K loop.  In the stream kernel (2 stages of 64-wide
// k-tiles) the barrier that publishes a stage sits between that stage's DMA and its first fragment
// read, so once per k-tile every wave waits out an LDS round trip with the matrix pipe empty, and both
// waves of a SIMD do so at the same moment (measured 3200 cycles per 64 k against 2048 of MFMA issue).
// Here the ring has FOUR stages of 32-wide k-tiles (same 128 KiB):
//   * the DMA of k-tile t+3 is issued in iteration t, and iteration t's barrier publishes k-tile t+1
//     -- one iteration AHEAD of its use.  Fragment reads therefore run one 16-wide k-step ahead of
//     the MFMAs straight through the barrier: a wave arrives at the barrier with the next k-step's
//     fragments already in registers and issues MFMAs the moment it leaves it;
//   * the barrier still orders the ring: the buffer refilled in iteration t held k-tile t-1, whose
//     last fragment reads every wave waited for before it arrived.
// 64-byte LDS rows (32 k): chunk c of row r lives at physical chunk c ^ ((r >> 2) & 3), which makes the
// 16-row ds_read_b128 groups conflict free; as in the other LDS-DMA kernels the swizzle is applied to
// the DMA's per-lane SOURCE address.  The epilogue stages through the k-tile buffer consumed last
// (32 KiB: 32 x 32 blocks per wave and pass instead of 32 x 64).
#pragma once
#include "gemm_stream_kernel.h"

namespace tfimm_gemm {

template <int KMODE>
__global__ void __launch_bounds__(512) gemm_pipe_kernel(const GemmStreamArgs pa) {
  const GemmArgs& p = pa.g;
  constexpr int BM = 256, BN = 256, WTM = 128, WTN = 64, TM = 4, TN = 2;
  constexpr int BKP = 32;                       // k per ring stage
  constexpr int NS = 4;                         // ring stages
  constexpr int A_BYTES = BM * 64, STAGE = (BM + BN) * 64;   // 32 KiB per stage
  constexpr int EPI_WAVE = 32 * 32 * 4;         // fp32 staging block of one wave and pass
  constexpr int NPIECE = 4;                     // DMA instructions per wave and stage (2 A + 2 B)
  constexpr int NSTORE = TM * TN * 2;           // output stores per wave and tile
  static_assert(KMODE == K_DENSE || KMODE == K_CONV, "LDS-DMA flavours only");
  static_assert(NS * STAGE <= 160 * 1024 && 8 * EPI_WAVE <= STAGE, "LDS budget");
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  int t_first, t_hi, t_step;
  {
    const int nb = gridDim.x;             // multiple of 8 (host)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int q = pa.n_tiles >> 3, r = pa.n_tiles & 7;
    const int t_lo = xcd * q + (xcd < r ? xcd : r);
    t_hi = t_lo + q + (xcd < r ? 1 : 0);
    t_step = nb >> 3;
    t_first = t_lo + j;
  }
  if (t_first >= t_hi) return;

  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(p.a, pa.a_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(p.wt, pa.w_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_o = make_rsrc(p.out, pa.out_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_r = make_rsrc(p.residual, pa.res_bytes);

  const int nk = (p.K + BKP - 1) / BKP;

  // ---- DMA source state of the tile being ISSUED.  A piece = 16 rows x 64 B; lane -> (row lane >> 2,
  //      physical chunk lane & 3); it fetches logical chunk (lane & 3) ^ ((row >> 2) & 3)
  const int drow = lane >> 2;
  const int dchunk = (lane & 3) ^ ((lane >> 4) & 3);   // rows of a piece start at a multiple of 16
  unsigned a_off[2], b_off[2];
  int a_iy0[2], a_ix0[2], a_pix[2];
  int s_ky = 0, s_kx = 0, s_ci0 = 0;

  auto setup_issue = [&](int tile, bool valid) __attribute__((always_inline)) {
    const int mt = tile / p.tiles_n, nt = tile - mt * p.tiles_n;
    const int m0 = mt * BM, n0 = nt * BN;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = (wave * 2 + j) * 16 + drow;
      const int m = m0 + r;
      const bool ok = valid && m < p.M;
      if (KMODE == K_DENSE) {
        a_off[j] = ok ? (unsigned)(((size_t)m * p.lda + dchunk * 8) * 2) : kOobOffset;
        a_iy0[j] = a_ix0[j] = a_pix[j] = 0;
      } else {
        const int mm = ok ? m : 0;
        const int ohw = p.OH * p.OW;
        const int b = mm / ohw;
        const int rem = mm - b * ohw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        a_iy0[j] = ok ? oy * p.stride - p.pad_t : -(1 << 28);
        a_ix0[j] = ox * p.stride_w - p.pad_l;
        a_pix[j] = b * p.H * p.W;
        a_off[j] = 0;
      }
      const int n = n0 + r;
      b_off[j] = (valid && n < p.N) ? (unsigned)(((size_t)n * p.ldw + dchunk * 8) * 2) : kOobOffset;
    }
    s_ky = s_kx = s_ci0 = 0;
  };
  auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
    char* sa = smem + stage * STAGE;
    char* sb = sa + A_BYTES;
    const int kbytes = kt * (BKP * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(sb + (wave * 2 + j) * 1024), 16, (int)b_off[j], kbytes, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (KMODE == K_DENSE) {
        const bool kok = (kt * BKP + dchunk * 8) < p.K;
        const unsigned off = kok ? a_off[j] : kOobOffset;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sa + (wave * 2 + j) * 1024), 16, (int)off, kbytes, 0, 0);
      } else {
        int ky, kx, ci;
        bool kok = true;
        if (pa.cin64) {
          ky = s_ky; kx = s_kx; ci = s_ci0 + dchunk * 8;
        } else {
          const int kg = kt * BKP + dchunk * 8;
          const int tap = pa.cin_magic ? (int)__umulhi((unsigned)kg, pa.cin_magic) : kg / p.Cin;
          ci = kg - tap * p.Cin;
          ky = pa.kw_magic ? (int)__umulhi((unsigned)tap, pa.kw_magic) : tap / p.KW;
          kx = tap - ky * p.KW;
          kok = kg < p.K;
        }
        const int iy = a_iy0[j] + ky, ix = a_ix0[j] + kx;
        const bool ok = kok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const unsigned off = ok ? (unsigned)((((size_t)(a_pix[j] + iy * p.W + ix)) * p.cpitch + ci) * 2) : kOobOffset;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sa + (wave * 2 + j) * 1024), 16, (int)off, 0, 0, 0);
      }
    }
    if (KMODE == K_CONV && pa.cin64) {
      s_ci0 += BKP;
      if (s_ci0 >= p.Cin) {
        s_ci0 = 0;
        if (++s_kx == p.KW) { s_kx = 0; ++s_ky; }
      }
    }
  };
  // flattened issue cursor: next (tile, k-tile) of this workgroup, past the end -> all out of range
  int iss_tile = t_first, iss_kt = 0;
  setup_issue(iss_tile, true);
  auto issue_next = [&](int stage) __attribute__((always_inline)) {
    if (iss_kt == nk) {
      iss_tile += t_step;
      iss_kt = 0;
      setup_issue(iss_tile, iss_tile < t_hi);
    }
    issue(iss_kt, stage);
    ++iss_kt;
  };

  // ---- fragment addressing: lane (frow, fhi) reads row (base + frow), logical chunk 2 ks + fhi
  const int frow = lane & 31;
  const int fhi = lane >> 5;
  const int fsw = (frow >> 2) & 3;
  const unsigned fa_base = (unsigned)((wm * WTM + frow) * 64 + ((fhi ^ fsw) * 16));            // ks = 0; ks = 1: ^ 32
  const unsigned fb_base = (unsigned)(A_BYTES + (wn * WTN + frow) * 64 + ((fhi ^ fsw) * 16));
  bf16x8 fa[2][TM], fb[2][TN];
  auto read_frags = [&](int stage, int ks, int set) __attribute__((always_inline)) {
    const char* sbase = smem + stage * STAGE;
    const unsigned x = ks ? 32u : 0u;
#pragma unroll
    for (int i = 0; i < TM; ++i)
      fa[set][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sbase + ((fa_base ^ x) + i * 32 * 64)));
#pragma unroll
    for (int j = 0; j < TN; ++j)
      fb[set][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sbase + ((fb_base ^ x) + j * 32 * 64)));
  };

  // epilogue geometry: 32 x 32 fp32 block per wave and pass, lane reads back 8 channels of one row
  constexpr int LPR = 4, RPI = 16, ITS = 2;
  auto epi_slot = [](int row, int slot) -> int { return slot ^ ((row >> 1) & 7); };
  const ActParams actp = make_act(p.act);
  const int e_row = lane / LPR;
  const int e_c8 = lane % LPR;
  const bool has_res = p.residual != nullptr;
  const int remap_eff = p.remap_in > 0 ? p.remap_in : 0x7fffffff;
  const int resmod_eff = p.res_mod > 0 ? p.res_mod : 0x7fffffff;
  const unsigned ldc2 = (unsigned)p.ldc * 2u, ldr2 = (unsigned)p.ldr * 2u;
  const unsigned out_wrap = p.remap_in > 0 ? (unsigned)(p.remap_out - p.remap_in) * ldc2 : 0u;
  const unsigned res_wrap = p.res_mod > 0 ? (unsigned)p.res_mod * ldr2 : 0u;

  // ---- prime: k-tiles 0, 1, 2 in flight; k-tile 0 landed and published; its first fragments read
  issue_next(0);
  issue_next(1);
  issue_next(2);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPIECE) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_frags(0, 0, 0);
  int cur = 0;            // ring stage of the k-tile being multiplied
  int stores_age = 0;     // > 0: the last tile's NSTORE output stores may still sit in the VMEM queue

  for (int tile = t_first; tile < t_hi; tile += t_step) {
    const int mt = tile / p.tiles_n, nt = tile - mt * p.tiles_n;
    const int m0 = mt * BM, n0 = nt * BN;
    const int e_m = m0 + wm * WTM + e_row;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    for (int kt = 0; kt < nk; ++kt) {
      // k-tile cur+1 must be complete before the barrier publishes it; VMEM retires in issue order, so
      // leaving the newest DMA group (k-tile cur+2) -- and, for two barriers after a tile end, the
      // NSTORE stores queued between the groups -- in flight is a counted wait
      if (stores_age > 0) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE + NSTORE) : "memory");
        --stores_age;
      } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
      }
      tfimm_lds_reuse_barrier();
      issue_next((cur + 3) & (NS - 1));        // refills the buffer of k-tile cur-1: dead behind this barrier
      // the scheduling barriers pin the software pipeline: hipcc otherwise sinks each fragment read
      // below the MFMAs of the other set (fewer live registers) and the LDS latency is exposed again
      read_frags(cur, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][j], fa[0][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      read_frags((cur + 1) & (NS - 1), 0, 0);  // next k-tile (possibly the next output tile's first)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1][j], fa[1][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      cur = (cur + 1) & (NS - 1);
    }

    // ---- epilogue through the buffer of the k-tile consumed last (stage cur-1): every wave must be done
    //      reading it; its refill is issued behind the next iteration's barrier, i.e. after all epilogues
    tfimm_lds_reuse_barrier();
    float* sEw = reinterpret_cast<float*>(smem + ((cur + NS - 1) & (NS - 1)) * STAGE + wave * EPI_WAVE);

    // per-tile epilogue state (see the stream kernel's VEC epilogue for the addressing scheme)
    tfimm_f32x2 bias2[TN][4];
    unsigned out_off0[TN], res_off0[TN];
    const int em = e_m < p.M ? e_m : p.M;
    const int rm0 = p.res_mod > 0 ? em % p.res_mod : em;
    const int oq0 = p.remap_in > 0 ? em / p.remap_in : 0;
    const int or0 = p.remap_in > 0 ? em - oq0 * p.remap_in : em;
    const int om0 = p.remap_in > 0 ? oq0 * p.remap_out + or0 + p.remap_off : em;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int e_n = n0 + wn * WTN + j * 32 + e_c8 * 8;
      const bool col_ok = e_n < p.N;
#pragma unroll
      for (int e = 0; e < 4; ++e) bias2[j][e] = tfimm_f32x2{0.f, 0.f};
      if (p.bias && col_ok) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + e_n);
        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + e_n + 4);
        bias2[j][0] = tfimm_f32x2{b0.x, b0.y}; bias2[j][1] = tfimm_f32x2{b0.z, b0.w};
        bias2[j][2] = tfimm_f32x2{b1.x, b1.y}; bias2[j][3] = tfimm_f32x2{b1.z, b1.w};
      }
      out_off0[j] = col_ok ? (unsigned)(((size_t)om0 * p.ldc + e_n) * 2) : kOobOffset;
      res_off0[j] = col_ok ? (unsigned)(((size_t)rm0 * p.ldr + e_n) * 2) : kOobOffset;
    }
    uint4 rres[ITS];
    auto load_res1 = [&](int i, int j, int it) __attribute__((always_inline)) {
      const int d = i * 32 + it * RPI;
      unsigned off = res_off0[j] + (unsigned)d * ldr2;
      off -= (rm0 + d >= resmod_eff) ? res_wrap : 0u;
      rres[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, (int)off, 0, 0));
    };
#pragma unroll
    for (int it = 0; it < ITS; ++it) load_res1(0, 0, it);

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        // MFMA result -> DS read hazard: hipcc pads nothing in front of inline asm (see the stream kernel)
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[i][j]));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int slot = q * 2 + fhi;
          const f32x4 v = {acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
          const unsigned addr = (unsigned)(size_t)(lds_ptr_t)(&sEw[frow * 32 + epi_slot(frow, slot) * 4]);
          asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
        }
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          const int pr = it * RPI + e_row;
          const float4 lo = *reinterpret_cast<const float4*>(&sEw[pr * 32 + epi_slot(pr, 2 * e_c8) * 4]);
          const float4 hi = *reinterpret_cast<const float4*>(&sEw[pr * 32 + epi_slot(pr, 2 * e_c8 + 1) * 4]);
          tfimm_f32x2 v[4] = {{lo.x, lo.y}, {lo.z, lo.w}, {hi.x, hi.y}, {hi.z, hi.w}};
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bias2[j][e];
          const uint4 rraw = rres[it];
          // next pass's residual segment: requested before this iteration's store, consumed a pass later
          if (j + 1 < TN) load_res1(i, j + 1, it);
          else if (i + 1 < TM) load_res1(i + 1, 0, it);
          tfimm_f32x2 r2[4];
          if (has_res) {
            asm volatile("");
            unpack8p(rraw, r2);
            if (p.act_after_res) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += r2[e];
            }
          }
          act8p(v, actp);
          if (has_res && !p.act_after_res) {
            asm volatile("");
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r2[e];
          }
          const int d = i * 32 + it * RPI;
          unsigned off = out_off0[j] + (unsigned)d * ldc2;
          off += (or0 + d >= remap_eff) ? out_wrap : 0u;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pack8p(v)), rsrc_o, (int)off, 0, 0);
        }
      }
    stores_age = 2;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace tfimm_gemm
