// bf16 MFMA GEMM / implicit-GEMM convolution, 256x128 tile, TWO CO-RESIDENT WORKGROUPS per CU (gfx950).
//
// Same operands, persistent XCD-contiguous tile walk, flattened (tile, k-tile) DMA cursor and fused epilogues as
// gemm_stream_kernel.h / gemm_pipe_kernel.h.  What changes is who overlaps with whom.  In the 8-wave kernels both
// waves of a SIMD belong to ONE workgroup: they meet at every barrier, and they run their epilogues (bias, GELU,
// LayerNorm correction, residual, conversion, stores -- a quarter of a K = 768 tile) at the same time, with the
// matrix pipe idle.  Here a workgroup has FOUR waves (one per SIMD, 128 x 64 accumulator block each, as before) and
// needs <= 80 KiB of LDS, so two workgroups share a CU.  They are independent instruction streams with their own
// barriers: while one sits at a barrier or works through its epilogue on the VALU / LDS / store path, the other has the
// SIMD's matrix pipe to itself.  The second workgroup of every CU starts `duo_delay` cycles late (about half a tile), so
// that the pair stays out of phase: an offset, once there, is kept (a workgroup in its epilogue does not slow the other's
// main loop down, two in their main loops share the pipe evenly).
//
// LDS (80 KiB per workgroup): three stages of 32-wide k-tiles (24 KiB each: 64-byte rows, chunk c of row r at physical
// chunk c ^ ((r >> 2) & 3), swizzle applied to the DMA's per-lane SOURCE address) + 8 KiB of per-tile tables (bias, and for
// a folded LayerNormalization the rows' (mean, rstd) pairs and the correction fragments) that arrive by LDS-DMA in the
// tile's first k-step.  The DMA runs TWO k-tiles ahead (a k-tile is 16 MFMAs per wave, half as long as in the 8-wave
// kernels), each k-tile is published by the barrier in front of its use; the stage refilled behind that barrier is the
// one every wave finished reading before it arrived.  The epilogue stages through the stage consumed last.
// Every LDS access outside the fragment reads is inline asm: hipcc orders an LDS access it can see behind ALL
// outstanding LDS-DMA (vmcnt(0)), which would drain the prefetch of the next tile at every epilogue.
#pragma once
#include "gemm_stream_kernel.h"

namespace tfimm_gemm {

struct DuoGeom {
  static constexpr int BM = 256, BN = 128, NW = 4, WTM = 128, WTN = 64, TM = 4, TN = 2;
  static constexpr int BKP = 32, NS = 3;
  static constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;   // 16 + 8 KiB
  static constexpr int TAB_BYTES = 8 * 1024;   // [bias 1 KiB][stats 2 KiB][c1 4 KiB][scratch 1 KiB]
  static constexpr int LDS_BYTES = NS * STAGE + TAB_BYTES;
  static_assert(LDS_BYTES <= 80 * 1024, "two workgroups must fit one CU's 160 KiB of LDS");
};

// EPI: 0 = vector epilogue with residual (fp32 staging, 32 x 32 passes), 1 = residual-free (arithmetic in the accumulator
// layout, bf16 staging), 2 = residual-free with a folded LayerNormalization (rank-1 MFMA correction, see gemm_stream_kernel.h)
template <int KMODE, int EPI>
__global__ void __launch_bounds__(256, 2) gemm_duo_kernel(const GemmStreamArgs pa) {
  using G = DuoGeom;
  const GemmArgs& p = pa.g;
  constexpr int BM = G::BM, BN = G::BN, WTM = G::WTM, WTN = G::WTN, TM = G::TM, TN = G::TN;
  constexpr int BKP = G::BKP, NS = G::NS, A_BYTES = G::A_BYTES, STAGE = G::STAGE;
  constexpr int NPA = 4, NPB = 2, NPIECE = NPA + NPB;   // 1-KiB DMA pieces (16 rows x 64 B) per wave and k-tile
  constexpr bool LNIN = EPI == 2, FAST = EPI >= 1;
  constexpr int NTAB = LNIN ? 2 : 1;                    // table DMA instructions per wave and tile
  constexpr int NSTORE = 16;                            // output stores per wave and tile (every epilogue flavour)
  static_assert(KMODE == K_DENSE || KMODE == K_CONV, "LDS-DMA flavours only");
  static_assert(!LNIN || KMODE == K_DENSE, "LayerNorm folding: dense rows");
  static_assert(NPIECE + NTAB + NSTORE < 64, "vmcnt is a 6-bit counter");
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

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
  // phase shift: the workgroups dispatched second onto the CUs of their XCD start late
  if (pa.duo_delay > 0 && (int)(blockIdx.x >> 3) >= pa.duo_first) {
    const long long t0 = (long long)__builtin_readcyclecounter();
    while ((long long)__builtin_readcyclecounter() - t0 < (long long)pa.duo_delay) __builtin_amdgcn_s_sleep(16);
  }

  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(p.a, pa.a_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(p.wt, pa.w_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_o = make_rsrc(p.out, pa.out_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_r = make_rsrc(p.residual, pa.res_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_b = make_rsrc(p.bias, p.bias ? (unsigned)p.N * 4u : 0u);
  const __amdgpu_buffer_rsrc_t rsrc_s = make_rsrc(pa.ln_stats, LNIN ? pa.ln_stats_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rsrc_c = make_rsrc(pa.ln_c1, LNIN ? pa.ln_c1_bytes : 0u);
  char* const tab = smem + NS * STAGE;
  char* const tab_bias = tab;
  char* const tab_stats = tab + 1024;
  char* const tab_c1 = tab + 3072;

  const int nk = (p.K + BKP - 1) / BKP;   // >= 2 (host)

  // ---- DMA source state of the tile being ISSUED.  A piece = 16 rows x 64 B; lane -> (row lane >> 2, physical chunk
  //      lane & 3), it fetches logical chunk (lane & 3) ^ ((row >> 2) & 3)
  const int drow = lane >> 2;
  const int dchunk = (lane & 3) ^ ((lane >> 4) & 3);   // rows of a piece start at a multiple of 16
  unsigned a_off[NPA], b_off[NPB];
  int a_iy0[NPA], a_ix0[NPA], a_pix[NPA];
  int s_ky = 0, s_kx = 0, s_ci0 = 0;

  auto setup_issue = [&](int tile, bool valid) __attribute__((always_inline)) {
    const int mt = tile / p.tiles_n, nt = tile - mt * p.tiles_n;
    const int m0 = mt * BM, n0 = nt * BN;
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
      const int r = (wave * NPA + j) * 16 + drow;
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
    }
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
      const int n = n0 + (wave * NPB + j) * 16 + drow;
      b_off[j] = (valid && n < p.N) ? (unsigned)(((size_t)n * p.ldw + dchunk * 8) * 2) : kOobOffset;
    }
    s_ky = s_kx = s_ci0 = 0;
  };
  auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
    char* sa = smem + stage * STAGE;
    char* sb = sa + A_BYTES;
    const int kbytes = kt * (BKP * 2);
#pragma unroll
    for (int j = 0; j < NPB; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(sb + (wave * NPB + j) * 1024), 16, (int)b_off[j], kbytes, 0, 0);
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
      if (KMODE == K_DENSE) {
        const bool kok = (kt * BKP + dchunk * 8) < p.K;
        const unsigned off = kok ? a_off[j] : kOobOffset;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sa + (wave * NPA + j) * 1024), 16, (int)off, kbytes, 0, 0);
      } else {
        int ky, kx, ci;
        bool kok = true;
        if (pa.cin64) {   // here: Cin % 32 == 0 -- the whole 32-wide k-tile lies inside tap (s_ky, s_kx)
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
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sa + (wave * NPA + j) * 1024), 16, (int)off, 0, 0, 0);
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
  // per-tile tables -> LDS.  Without LayerNorm folding every wave fetches the tile's 128 bias values into the same
  // 1 KiB (identical bytes: no wave depends on another's copy); with it the seven pieces (bias, 2 x stats of 128 rows,
  // 4 x correction fragments of 32 columns) are shared out two per wave and published by the barrier of the second k-step
  auto issue_tables = [&](int m0, int n0) __attribute__((always_inline)) {
    const unsigned boff = (lane < 32) ? (unsigned)((n0 + lane * 4) * 4) : kOobOffset;
    if (!LNIN) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)tab_bias, 16, (int)boff, 0, 0, 0);
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = wave * 2 + u;
        if (idx == 0 || idx == 7) {          // 7: the bias again, into the scratch KiB (keeps the per-wave count uniform)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(idx == 0 ? tab_bias : tab + 7168), 16, (int)boff, 0, 0, 0);
        } else if (idx <= 2) {               // rows m0 + 128 (idx - 1) + 2 lane, + 1
          const unsigned so = (unsigned)(m0 + (idx - 1) * 128) * 8u + (unsigned)lane * 16u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_s, (lds_ptr_t)(tab_stats + (idx - 1) * 1024), 16, (int)so, 0, 0, 0);
        } else {                             // columns n0 + 32 (idx - 3) + (lane & 31), fragment half lane >> 5
          const unsigned co = (unsigned)(((n0 + (idx - 3) * 32 + (lane & 31)) * 2 + (lane >> 5)) * 16);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_c, (lds_ptr_t)(tab_c1 + (idx - 3) * 1024), 16, (int)co, 0, 0, 0);
        }
      }
    }
  };

  // ---- fragment addressing: lane (frow, fhi) reads row (base + frow), logical chunk 2 ks + fhi
  const int frow = lane & 31;
  const int fhi = lane >> 5;
  const int fsw = (frow >> 2) & 3;
  const unsigned fa_base = (unsigned)((wm * WTM + frow) * 64 + ((fhi ^ fsw) * 16));            // ks = 0; ks = 1: ^ 32
  const unsigned fb_base = (unsigned)(A_BYTES + (wn * WTN + frow) * 64 + ((fhi ^ fsw) * 16));

  const ActParams actp = make_act(p.act);
  const bool has_res = p.residual != nullptr;
  const int remap_eff = p.remap_in > 0 ? p.remap_in : 0x7fffffff;
  const int resmod_eff = p.res_mod > 0 ? p.res_mod : 0x7fffffff;
  const unsigned ldc2 = (unsigned)p.ldc * 2u, ldr2 = (unsigned)p.ldr * 2u;
  const unsigned out_wrap = p.remap_in > 0 ? (unsigned)(p.remap_out - p.remap_in) * ldc2 : 0u;
  const unsigned res_wrap = p.res_mod > 0 ? (unsigned)p.res_mod * ldr2 : 0u;

  // ---- prime: k-tiles 0 and 1 of the first tile in flight
  issue_next(0);
  issue_next(1);
  int cur = 0;                   // ring stage of the k-tile being multiplied
  bool stores_pending = false;   // the previous tile's NSTORE output stores may still sit in the VMEM queue

  for (int tile = t_first; tile < t_hi; tile += t_step) {
    const int mt = tile / p.tiles_n, nt = tile - mt * p.tiles_n;
    const int m0 = mt * BM, n0 = nt * BN;

    // epilogue addressing of this tile (see the stream kernel's vector epilogue): lane -> 8 consecutive channels of one row
    constexpr int LPR = FAST ? 8 : 4;              // lanes per staged row at read-back
    constexpr int RPI = 64 / LPR;                  // rows per read-back instruction
    constexpr int ITS = 32 / RPI;                  // read-back instructions per 32-row pass
    const int e_row = lane / LPR, e_c8 = lane % LPR;
    const int e_m = m0 + wm * WTM + e_row;
    const int em = e_m < p.M ? e_m : p.M;          // clamp: offsets stay inside 32 bits
    const int rm0 = p.res_mod > 0 ? em % p.res_mod : em;
    const int oq0 = p.remap_in > 0 ? em / p.remap_in : 0;
    const int or0 = p.remap_in > 0 ? em - oq0 * p.remap_in : em;
    const int om0 = p.remap_in > 0 ? oq0 * p.remap_out + or0 + p.remap_off : em;
    unsigned out_off0[FAST ? 1 : TN], res_off0[FAST ? 1 : TN];
#pragma unroll
    for (int j = 0; j < (FAST ? 1 : TN); ++j) {
      const int e_n = n0 + wn * WTN + j * 32 + e_c8 * 8;
      const bool col_ok = e_n < p.N;               // N % 8 == 0: all 8 channels or none
      out_off0[j] = col_ok ? (unsigned)(((size_t)om0 * p.ldc + e_n) * 2) : kOobOffset;
      res_off0[j] = col_ok ? (unsigned)(((size_t)rm0 * p.ldr + e_n) * 2) : kOobOffset;
    }
    uint4 rres[ITS];
    auto load_res1 = [&](int i, int j, int it) __attribute__((always_inline)) {
      const int d = i * 32 + it * RPI;
      unsigned off = res_off0[FAST ? 0 : j] + (unsigned)d * ldr2;
      off -= (rm0 + d >= resmod_eff) ? res_wrap : 0u;
      rres[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, (int)off, 0, 0));
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    for (int kt = 0; kt < nk; ++kt) {
      // This wave's pieces of the k-tile in stage `cur` must have landed.  VMEM operations retire in issue order; younger
      // than those pieces are the next k-tile's (NPIECE) and, in a tile's first two steps, the previous tile's output
      // stores and this tile's table pieces: leave exactly those in flight.
      if (kt == 0) {
        if (stores_pending) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE + NSTORE) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
        }
        stores_pending = false;
      } else if (kt == 1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE + NTAB) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
      }
      tfimm_lds_reuse_barrier();                   // ... everyone's; the stage of k-tile cur-1 (and the epilogue block in it) is free: every wave's reads of it are complete
      // first residual rows of this tile: requested ahead of the DMA below, so waiting for them leaves that DMA in flight
      if (!FAST && kt == nk - 1) {
#pragma unroll
        for (int it = 0; it < ITS; ++it) load_res1(0, 0, it);
      }
      if (kt == 0) issue_tables(m0, n0);
      issue_next(cur >= 1 ? cur - 1 : NS - 1);     // (cur + 2) % NS: held k-tile cur-1

      const char* sbase = smem + cur * STAGE;
      bf16x8 fa[2][TM], fb[2][TN];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const unsigned x = ks ? 32u : 0u;
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[ks][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sbase + ((fb_base ^ x) + j * 32 * 64)));
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[ks][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sbase + ((fa_base ^ x) + i * 32 * 64)));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      cur = cur + 1 == NS ? 0 : cur + 1;
    }

    // ---- the tile's tables (bias, LayerNorm statistics / correction fragments) were requested in its FIRST k-step; a tile
    //      of three or more k-tiles has waited for them in its third step (the counted wait there leaves only the second
    //      step's pieces in flight).  With exactly two k-tiles no wait has covered them yet: without this one the epilogue
    //      read the PREVIOUS tile's bias whenever the 1-KiB table piece was slower than two k-steps (found by the
    //      eager-vs-replay bit-equality test on EfficientNet-B4's K = 56 expansions: one 256 x 128 tile in ~7000 wrong,
    //      in some runs).  Younger than the tables: the two k-steps' pieces -- the second step's (and this tile's first
    //      residual rows, requested in front of them) may stay in flight.
    if (nk == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE + (FAST ? 0 : ITS)) : "memory");
    // ---- epilogue through the stage of the k-tile consumed last (cur - 1): every wave must be done reading it; its
    //      refill is issued behind the next step's barrier, i.e. after every wave finished this epilogue
    tfimm_lds_reuse_barrier();
    char* const sE = smem + (cur >= 1 ? cur - 1 : NS - 1) * STAGE + wave * 4096;   // 4 KiB per wave

    if constexpr (FAST) {
      // in the accumulator layout a lane owns channels j*32 + q*8 + fhi*4 .. +3 of its rows: bias quads from the table
      u32x4 bq[TN * 4];
      {
        const unsigned ba = (unsigned)(size_t)(lds_ptr_t)(tab_bias + (wn * WTN + fhi * 4) * 4);
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:32\n\tds_read_b128 %2, %8 offset:64\n\t"
                     "ds_read_b128 %3, %8 offset:96\n\tds_read_b128 %4, %8 offset:128\n\tds_read_b128 %5, %8 offset:160\n\t"
                     "ds_read_b128 %6, %8 offset:192\n\tds_read_b128 %7, %8 offset:224\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(bq[0]), "=&v"(bq[1]), "=&v"(bq[2]), "=&v"(bq[3]), "=&v"(bq[4]), "=&v"(bq[5]), "=&v"(bq[6]), "=&v"(bq[7])
                     : "v"(ba) : "memory");
      }
      constexpr int CPR = WTN / 8;                 // 16-byte chunks per staged bf16 row
      const int wsw = (frow >> 1) & 7;             // chunk swizzle of this lane's row
      unsigned rb_addr[ITS];
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        const int pr = it * RPI + e_row;
        rb_addr[it] = (unsigned)(size_t)(lds_ptr_t)(sE + pr * (WTN * 2) + ((e_c8 ^ ((pr >> 1) & 7)) * 16));
      }
      static_assert(CPR == 8 && ITS == 4, "fast epilogue: 64 columns per wave");
      u32x4 cfr[TN];
      if (LNIN) {
        const unsigned ca = (unsigned)(size_t)(lds_ptr_t)(tab_c1 + (wn * TN) * 1024 + lane * 16);
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(cfr[0]), "=&v"(cfr[1]) : "v"(ca) : "memory");
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        tfimm_f32x2 rs2 = {1.f, 1.f};
        if (LNIN) {
          // (mean, rstd) of this lane's row of pass i; -mean as three bf16 terms in the k-slots that meet the column fragments
          tfimm_f32x2 st;
          const unsigned sa = (unsigned)(size_t)(lds_ptr_t)(tab_stats + (wm * WTM + i * 32 + frow) * 8);
          asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(st) : "v"(sa) : "memory");
          const float nm = -st[0];
          const uint32_t b1 = __float_as_uint(nm) & 0xffff0000u;
          const float r1 = nm - __uint_as_float(b1);
          const uint32_t b2 = __float_as_uint(r1) & 0xffff0000u;
          const float r2 = r1 - __uint_as_float(b2);
          const uint32_t b3 = __float_as_uint(r2) & 0xffff0000u;
          u32x4 fx;
          fx[0] = fhi ? (b3 >> 16) : (b1 | (b1 >> 16));
          fx[1] = fhi ? 0u : (b2 | (b1 >> 16));
          fx[2] = fhi ? 0u : (b2 | (b2 >> 16));
          fx[3] = fhi ? 0u : (b3 | (b3 >> 16));
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cfr[j]), __builtin_bit_cast(bf16x8, fx),
                                                                acc[i][j], 0, 0, 0);
          rs2 = tfimm_f32x2{st[1], st[1]};
          asm volatile("" : "+v"(rs2));   // a real register pair (see the stream kernel: op_sel folding misbehaved)
        }
        // MFMA result -> DS / VALU read through inline asm: hipcc pads no hazard in front of an asm statement
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[i][0]), "+v"(acc[i][1]));
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q2 = 0; q2 < 2; ++q2) {
            tfimm_f32x2 v[4];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
              const int q = q2 * 2 + h2;
              const f32x4 b4 = __builtin_bit_cast(f32x4, bq[j * 4 + q]);
              if (LNIN) {
                v[h2 * 2 + 0] = __builtin_elementwise_fma(tfimm_f32x2{acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1]}, rs2,
                                                          tfimm_f32x2{b4[0], b4[1]});
                v[h2 * 2 + 1] = __builtin_elementwise_fma(tfimm_f32x2{acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]}, rs2,
                                                          tfimm_f32x2{b4[2], b4[3]});
              } else {
                v[h2 * 2 + 0] = tfimm_f32x2{acc[i][j][q * 4 + 0] + b4[0], acc[i][j][q * 4 + 1] + b4[1]};
                v[h2 * 2 + 1] = tfimm_f32x2{acc[i][j][q * 4 + 2] + b4[2], acc[i][j][q * 4 + 3] + b4[3]};
              }
            }
            act8p(v, actp);
            const uint4 pk = pack8p(v);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
              const int chunk = j * 4 + q2 * 2 + h2;
              const unsigned addr = (unsigned)(size_t)(lds_ptr_t)(sE + frow * (WTN * 2) + ((chunk ^ wsw) * 16) + fhi * 8);
              const uint2 w2 = h2 ? make_uint2(pk.z, pk.w) : make_uint2(pk.x, pk.y);
              asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(w2) : "memory");
            }
          }
        u32x4 o16[ITS];
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(o16[0]), "=&v"(o16[1]), "=&v"(o16[2]), "=&v"(o16[3])
                     : "v"(rb_addr[0]), "v"(rb_addr[1]), "v"(rb_addr[2]), "v"(rb_addr[3]) : "memory");
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          const int d = i * 32 + it * RPI;
          unsigned off = out_off0[0] + (unsigned)d * ldc2;
          off += (or0 + d >= remap_eff) ? out_wrap : 0u;
          __builtin_amdgcn_raw_buffer_store_b128(o16[it], rsrc_o, (int)off, 0, 0);
        }
      }
    } else {
      // residual flavour: 32 x 32 fp32 block per pass, read back row-contiguous (lane: 8 channels of one row)
      auto epi_slot = [](int row, int slot) -> int { return slot ^ ((row >> 1) & 7); };
      tfimm_f32x2 bias2[TN][4];
      {
        u32x4 braw[TN * 2];
        const unsigned ba = (unsigned)(size_t)(lds_ptr_t)(tab_bias + (wn * WTN + e_c8 * 8) * 4);
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:128\n\t"
                     "ds_read_b128 %3, %4 offset:144\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(braw[0]), "=&v"(braw[1]), "=&v"(braw[2]), "=&v"(braw[3]) : "v"(ba) : "memory");
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const f32x4 lo = __builtin_bit_cast(f32x4, braw[2 * j]), hi = __builtin_bit_cast(f32x4, braw[2 * j + 1]);
          bias2[j][0] = tfimm_f32x2{lo[0], lo[1]}; bias2[j][1] = tfimm_f32x2{lo[2], lo[3]};
          bias2[j][2] = tfimm_f32x2{hi[0], hi[1]}; bias2[j][3] = tfimm_f32x2{hi[2], hi[3]};
        }
      }
      float* const sEw = reinterpret_cast<float*>(sE);
      unsigned ra[ITS][2];
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        const int pr = it * RPI + e_row;
        ra[it][0] = (unsigned)(size_t)(lds_ptr_t)(&sEw[pr * 32 + epi_slot(pr, 2 * e_c8) * 4]);
        ra[it][1] = (unsigned)(size_t)(lds_ptr_t)(&sEw[pr * 32 + epi_slot(pr, 2 * e_c8 + 1) * 4]);
      }
      static_assert(ITS == 2, "residual epilogue: 32-column passes");
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[i][j]));
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int slot = q * 2 + fhi;
            const f32x4 v = {acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
            const unsigned addr = (unsigned)(size_t)(lds_ptr_t)(&sEw[frow * 32 + epi_slot(frow, slot) * 4]);
            asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
          }
          f32x4 st[4];
          asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(st[0]), "=&v"(st[1]), "=&v"(st[2]), "=&v"(st[3])
                       : "v"(ra[0][0]), "v"(ra[0][1]), "v"(ra[1][0]), "v"(ra[1][1]) : "memory");
#pragma unroll
          for (int it = 0; it < ITS; ++it) {
            const f32x4 lo = st[2 * it], hi = st[2 * it + 1];
            tfimm_f32x2 v[4] = {{lo[0], lo[1]}, {lo[2], lo[3]}, {hi[0], hi[1]}, {hi[2], hi[3]}};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bias2[j][e];
            const uint4 rraw = rres[it];
            // the next pass's residual segment: requested before this iteration's store and consumed a pass later
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
    }
    stores_pending = true;
  }
  // the last steps' (all out-of-range) prefetches must have landed before this workgroup's LDS is released
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace tfimm_gemm
