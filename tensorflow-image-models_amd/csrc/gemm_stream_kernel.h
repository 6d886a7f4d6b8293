// bf16 MFMA GEMM / implicit-GEMM convolution, persistent LDS-DMA variant (gfx950).
//
// Same math, operand layouts and fused epilogue as gemm_dma_kernel.h; what changes is the
// schedule around the K loop:
//   * PERSISTENT workgroups: the grid is (CUs x resident blocks per CU); each workgroup walks a
//     strided list of output tiles.  The 8 XCDs own contiguous tile ranges and the workgroups of
//     one XCD work on neighbouring tiles at any moment, so an activation panel and the weight
//     panels it meets are pulled through ONE L2.
//   * the (tile, k-tile) loop is FLATTENED: the LDS-DMA for the next step -- which may be the
//     first k-tile of the NEXT output tile -- is issued right after the barrier that publishes the
//     current one.  HBM/L2 latency of a tile's first operands therefore hides under the previous
//     tile's last MFMAs and its epilogue instead of being exposed once per tile (which is what
//     made the one-tile-per-workgroup kernel latency-bound on the K <= 256 convolutions of
//     ResNet/EfficientNet and cost ~30 % on K = 768 ViT GEMMs).
//   * epilogue per WAVE: a wave stages its 32 x WTN fp32 accumulator block in a private,
//     XOR-swizzled LDS region and reads it back row-contiguous, so stores / residual loads are
//     16 B per lane over full 64..128-byte row segments and need no workgroup barrier.  The region
//     aliases the ring stage that was just consumed when a dedicated one does not fit in 160 KiB
//     (one extra barrier per tile), otherwise it is separate and a wave's epilogue overlaps the
//     other waves' MFMAs.
//   * VMEM operations retire in issue order (one vmcnt for loads AND stores on gfx9), so a load
//     whose data is needed right behind a store waits for that store's acknowledgement.  The
//     vector epilogue (VEC = true) is therefore straight-line code over buffer descriptors
//     (out-of-range lanes get an offset past num_records instead of a branch): bias is loaded at
//     tile start, residual rows are requested one 32-row pass ahead of their use and BEFORE the
//     stores of the current pass, and the wait in front of the next tile's first barrier leaves
//     this tile's stores in flight (counted vmcnt) instead of draining them.
//   * NHWC gather (K_CONV): when Cin % 64 == 0 every 64-wide k-tile lies inside one filter tap,
//     so (ky, kx, ci0) is wave-uniform scalar state stepped once per k-tile -- no per-lane
//     integer divisions in the loop.
// VEC = false is the catch-all (ragged N, unaligned rows): same main loop, element-wise epilogue
// loops over the staged block.
#pragma once
#include "gemm_dma_kernel.h"

namespace tfimm_gemm {

struct GemmStreamArgs {
  GemmArgs g;
  unsigned a_bytes, w_bytes;
  unsigned out_bytes, res_bytes;   // extents of the output / residual buffers (descriptor bounds)
  int n_tiles;       // tiles_m * tiles_n
  int cin64;         // K_CONV: Cin % 64 == 0 (scalar tap stepping)
  unsigned cin_magic, kw_magic;   // K_CONV otherwise: floor(2^32 / d) + 1 for d = Cin, KW (0: K >= 65536, divide)
  // SCALE flavour (SE gate on the A operand, a_scale[image][k]): the gate values of one k-tile -- 64 floats for each image a row
  // tile touches -- ride in LDS next to the operand stage of that k-tile, brought by the same DMA stream as the operands
  unsigned s_bytes;   // extent of the gate table
  int s_slots;        // image slots per tile: ceil(BM / rows_per_image) + 1
  int s_gp;           // 1-KiB DMA pieces per k-tile: ceil(s_slots / 4) (four slots of 256 bytes each), one per wave 0 .. s_gp-1
  // DUAL flavour (tfimm_gemm_desc::a2): a second dense A operand whose K2 channels are further k-tiles of the same tile -- the
  // shortcut convolution of a residual block inside the block's last 1x1 convolution.  Row m of the output reads row
  // a2_row(m) of a2: m itself at stride 1, pixel (b, oy s, ox s) of an [a2_H][a2_W] image at stride s.
  const bf16_t* a2;
  unsigned a2_bytes;
  int K2, lda2, a2_stride, a2_H, a2_W, a2_OH, a2_OW;
  int a2_window;     // taps per side of the second operand's window (1: a 1x1 view)
  // LNIN flavour (LayerNorm folded into this GEMM): per-row (mean, rstd) and the split column sums of the gamma-scaled weights
  const float* ln_stats;      // fp32 [M][2]
  const void* ln_c1;          // bf16 [N][2][8] correction fragments (pack.pack_ln_c1)
  unsigned ln_stats_bytes, ln_c1_bytes;
  // duo kernel (gemm_duo_kernel.h): workgroups with (blockIdx.x >> 3) >= duo_first start duo_delay cycles late
  // tile order inside the list: 0 = M-panel-major (a row panel of A meets every weight panel before the next row panel starts);
  // g > 0 = groups of g weight panels, each group swept over ALL row panels before the next group starts -- the group's weights
  // (g x BN x K x 2 bytes) stay in the XCD's L2 while the rows stream by, at the price of reading A once per group.  Pays when
  // the weight matrix does not fit in a 4-MiB L2 and is re-fetched per row panel (ViT-B fc1: 4.7 MB x 394 row panels).
  int ngroup;
  int duo_delay, duo_first;
  long long* dbg_ptr; // TFIMM_GEMM_DBG_PTR: s_memtime stamps of workgroup 0 (dbg & 64)
  int dbg;           // TFIMM_GEMM_DBG: bit 64 = record the stamps (tools/gemm_stamps.py)
};

typedef void (*gemm_stream_fn)(const GemmStreamArgs);

template <int BM, int BN, int WAVES_M, int WAVES_N>
struct StreamGeom {
  static constexpr int NW = WAVES_M * WAVES_N;
  static constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static constexpr int EPI_WAVE = 32 * WTN * 4;          // fp32 staging block of one wave
  static constexpr int EPI_BYTES = NW * EPI_WAVE;
  static constexpr bool EPI_ALIAS = EPI_BYTES <= STAGE;  // fits inside one ring stage
  static constexpr int LDS_BYTES = 2 * STAGE + (EPI_ALIAS ? 0 : EPI_BYTES);
  static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit in LDS");
};

// LNIN: the A operand is the RAW input of a LayerNormalization that the host folded into this layer -- the weights carry gamma,
// the bias carries beta . W, and  LN(x) . W = rstd_m * (x . W' - mean_m * c1[n]) + b'[n]  with c1[n] = sum_k W'[k][n].
// -mean_m * c1[n] is a rank-1 update: ONE extra MFMA k-step per accumulator block whose operands are the bf16 three-way splits
// of -mean_m (built in registers) and of c1[n] (host-packed), nine exact products that carry ~24 bits; rstd_m rides on the
// bias FMA.  Statistics and correction fragments of a tile arrive by LDS-DMA at tile start (no registers across the K loop).
template <int BM, int BN, int WAVES_M, int WAVES_N, int KMODE, bool VEC, bool SCALE = false, bool NORES = false, bool LNIN = false,
          bool DUAL = false>
__global__ void __launch_bounds__(WAVES_M* WAVES_N * 64) gemm_stream_kernel(const GemmStreamArgs pa) {
  using G = StreamGeom<BM, BN, WAVES_M, WAVES_N>;
  const GemmArgs& p = pa.g;
  constexpr int NW = G::NW;
  constexpr int A_INSTR = BM / 8 / NW;  // 1-KiB DMA instructions per wave per k-tile
  constexpr int B_INSTR = BN / 8 / NW;
  constexpr int WTM = G::WTM, WTN = G::WTN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(A_INSTR >= 1 && B_INSTR >= 1 && TM >= 1 && TN >= 1, "tile/wave mismatch");
  static_assert(KMODE == K_DENSE || KMODE == K_CONV, "LDS-DMA flavours: dense rows or Cin % 8 == 0 gather");
  static_assert(WTN == 32 || WTN == 64, "epilogue swizzle is written for 32/64-wide wave tiles");
  static_assert(!SCALE || KMODE == K_DENSE, "the SE-gate prologue exists for dense rows");
  static_assert(!LNIN || (VEC && NORES && !SCALE && KMODE == K_DENSE), "LayerNorm folding: dense rows, residual-free vector epilogue");
  static_assert(!DUAL || (VEC && NORES && !SCALE && !LNIN), "second A operand: residual-free vector epilogue");
  constexpr int A_BYTES = G::A_BYTES, STAGE = G::STAGE;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- this workgroup's tile list: XCD x owns tiles [t_lo, t_hi); its workgroups stride through it
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
  // tile index -> (row panel, column panel): see GemmStreamArgs::ngroup
  auto tile_mn = [&](int tile, int& mt, int& nt) __attribute__((always_inline)) {
    if (pa.ngroup <= 0) {
      mt = tile / p.tiles_n;
      nt = tile - mt * p.tiles_n;
    } else {
      const int per_group = p.tiles_m * pa.ngroup;
      const int g = tile / per_group, r = tile - g * per_group;
      const int gw = min(pa.ngroup, p.tiles_n - g * pa.ngroup);      // the last group may be narrower
      mt = r / gw;
      nt = g * pa.ngroup + (r - mt * gw);
    }
  };

  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(p.a, pa.a_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(p.wt, pa.w_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_o = make_rsrc(p.out, pa.out_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_r = make_rsrc(p.residual, pa.res_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_s = LNIN ? make_rsrc(pa.ln_stats, pa.ln_stats_bytes) : make_rsrc(p.a_scale, SCALE ? pa.s_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rsrc_c = make_rsrc(pa.ln_c1, LNIN ? pa.ln_c1_bytes : 0u);
  const __amdgpu_buffer_rsrc_t rsrc_a2 = make_rsrc(pa.a2, DUAL ? pa.a2_bytes : 0u);
  // LNIN: this wave's private LDS block behind the operand ring: 1 KiB of (mean, rstd) pairs for 128 rows, then TN x 1 KiB
  // of correction fragments (one 16-byte fragment per lane and 32-column block)
  char* const lnw = smem + G::LDS_BYTES + wave * (1 + TN) * 1024;
  // gate region behind the operand ring: for each of the two stages s_gp KiB, slot j of a k-tile at byte j * 256
  char* const sS = smem + G::LDS_BYTES;
  const int s_gbytes = SCALE ? pa.s_gp * 1024 : 0;

  const int nk1 = (p.K + BK - 1) / BK;                          // k-tiles of the (first) A operand
  const int nk2c = DUAL ? (pa.K2 + BK - 1) / BK : 0;            // DUAL: k-tiles of one tap of the second operand
  const int nk = nk1 + nk2c * (DUAL ? pa.a2_window * pa.a2_window : 0);      // ... they follow the first operand's (weights: column nk1 * 64 on)
  const int lrow = lane >> 3;   // row within an 8-row DMA piece
  const int lpc = lane & 7;     // physical 16-byte chunk this lane fills

  // ---- DMA source state of the tile being ISSUED (one step ahead of the tile being computed)
  unsigned a_off[A_INSTR];      // dense: byte offset of (row, chunk) at k = 0, or kOobOffset
  unsigned a_off2[DUAL ? A_INSTR : 1];   // DUAL: the same for the second operand
  int a_iy0[A_INSTR], a_ix0[A_INSTR], a_pix[A_INSTR];
  unsigned b_off[B_INSTR];
  int s_ky = 0, s_kx = 0, s_ci0 = 0;   // K_CONV + cin64: wave-uniform tap state of the next k-tile
  int s_b0 = 0;                        // SCALE: first image of the tile being issued (-1: no tile)
  int s_k2 = 0, s_dx = 0;              // DUAL: k-tile inside the current tap of the second operand, the tap's column
  unsigned s_tapoff = 0;               // ... and its byte offset (dy * a2_W + dx) * lda2 * 2

  auto a_chunk = [&](int j) -> int {    // logical 16-byte k-chunk this lane fetches for DMA piece j
    const int r = (wave * A_INSTR + j) * 8 + lrow;
    return lpc ^ ((r >> 1) & 7);
  };

  // `valid` false (no further tile for this workgroup): every offset out of range, the DMA that is
  // still issued unconditionally then only writes zeros -- keeps the K loop free of VMEM branches
  auto setup_issue = [&](int tile, bool valid) __attribute__((always_inline)) {
    int mt, nt;
    tile_mn(tile, mt, nt);
    const int m0 = mt * BM, n0 = nt * BN;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      const int r = (wave * A_INSTR + j) * 8 + lrow;
      const int m = m0 + r;
      const bool ok = valid && m < p.M;
      if (KMODE == K_DENSE) {
        // probe build (-DTFIMM_STREAM_DBG), TFIMM_GEMM_DBG & 128: every tile fetches the FIRST activation panel -- what the loop does when
        // the A operand always hits in L2
#ifdef TFIMM_STREAM_DBG
        const int msrc = (TFIMM_PROBE(pa.dbg) & 128) ? r : m;
#else
        const int msrc = m;
#endif
        a_off[j] = ok ? (unsigned)(((size_t)msrc * p.lda + a_chunk(j) * 8) * 2) : kOobOffset;
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
      if (DUAL) {        // the second operand is dense rows whatever the first one is (a gather for the 3 x 3 conv2 of a basic block)
        int r2 = m;
        if (pa.a2_stride > 1 || pa.a2_window > 1) {  // (b, oy, ox) of the output row -> pixel (b, oy s, ox s) of the input image
          const int mm = ok ? m : 0;
          const int ohw = pa.a2_OH * pa.a2_OW;
          const int b = mm / ohw, rem = mm - b * ohw;
          const int oy = rem / pa.a2_OW, ox = rem - oy * pa.a2_OW;
          r2 = (b * pa.a2_H + oy * pa.a2_stride) * pa.a2_W + ox * pa.a2_stride;
        }
        a_off2[j] = ok ? (unsigned)(((size_t)r2 * pa.lda2 + a_chunk(j) * 8) * 2) : kOobOffset;
      }
    }
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
      const int r = (wave * B_INSTR + j) * 8 + lrow;
      const int chunk = lpc ^ ((r >> 1) & 7);
      const int n = n0 + r;
#ifdef TFIMM_STREAM_DBG
      const int nsrc = (TFIMM_PROBE(pa.dbg) & 256) ? r : n;      // measurement only: every tile fetches the first weight panel
#else
      const int nsrc = n;
#endif
      b_off[j] = (valid && n < p.N) ? (unsigned)(((size_t)nsrc * p.ldw + chunk * 8) * 2) : kOobOffset;
    }
    s_ky = s_kx = s_ci0 = 0;
    if (DUAL) { s_k2 = 0; s_dx = 0; s_tapoff = 0u; }
    if (SCALE) s_b0 = valid ? m0 / p.rows_per_image : -1;
  };

  // One 1-KiB DMA piece of the step being issued: pieces [0, B_INSTR) are this wave's weight rows,
  // [B_INSTR, B_INSTR + A_INSTR) its activation rows.  (Measured: spreading the pieces over the
  // k-slices of the previous step instead of issuing them in one block behind the barrier, and
  // s_setprio around the MFMA groups, are both neutral in this one-barrier-per-k-tile structure.
  // What the DMA costs is LDS time: with the DMA removed the same loop runs 622k instead of 867k
  // cycles per wave on M=100864 K=3072 N=768, i.e. ~1100 cycles per k-step of fragment-read stall
  // while 64 KiB of DMA writes share the LDS with 192 KiB of ds_read_b128 traffic.)
  constexpr int N_PIECES = A_INSTR + B_INSTR;
  auto issue_piece = [&](int piece, int kt, int stage) __attribute__((always_inline)) {
    char* sa = smem + stage * STAGE;
    char* sb = sa + A_BYTES;
    const int kbytes = kt * 128;
    if (piece < B_INSTR) {
      // B (weights): rows are zero padded to a multiple of 64 -> always in range in k
      const int j = piece;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(sb + (wave * B_INSTR + j) * 1024), 16,
                                               (int)b_off[j], kbytes, 0, 0);
    } else if (DUAL && kt >= nk1) {      // (wave-uniform) a k-tile of the second operand: dense rows of tap (s_tapoff), k-tile s_k2
      const int j = piece - B_INSTR;
      const bool kok = (s_k2 * BK + a_chunk(j) * 8) < pa.K2;
      const unsigned off = (kok && a_off2[j] != kOobOffset) ? a_off2[j] + s_tapoff : kOobOffset;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a2, (lds_ptr_t)(sa + (wave * A_INSTR + j) * 1024), 16,
                                               (int)off, s_k2 * 128, 0, 0);
    } else if (KMODE == K_DENSE) {
      const int j = piece - B_INSTR;
      const bool kok = (kt * BK + a_chunk(j) * 8) < p.K;
      const unsigned off = kok ? a_off[j] : kOobOffset;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sa + (wave * A_INSTR + j) * 1024), 16,
                                               (int)off, kbytes, 0, 0);
    } else {
      const int j = piece - B_INSTR;
      int ky, kx, ci;
      bool kok = true;
      if (pa.cin64) {   // whole k-tile inside tap (s_ky, s_kx), channels s_ci0 .. s_ci0 + 63
        ky = s_ky; kx = s_kx; ci = s_ci0 + a_chunk(j) * 8;
      } else {
        // k -> (ky, kx, ci): exact multiply-high division (k < 65536), one per DMA piece per k-tile
        const int kg = kt * BK + a_chunk(j) * 8;
        const int tap = pa.cin_magic ? (int)__umulhi((unsigned)kg, pa.cin_magic) : kg / p.Cin;
        ci = kg - tap * p.Cin;
        ky = pa.kw_magic ? (int)__umulhi((unsigned)tap, pa.kw_magic) : tap / p.KW;
        kx = tap - ky * p.KW;
        kok = kg < p.K;
      }
      const int iy = a_iy0[j] + ky, ix = a_ix0[j] + kx;
      const bool ok = kok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const unsigned off = ok ? (unsigned)((((size_t)(a_pix[j] + iy * p.W + ix)) * p.cpitch + ci) * 2) : kOobOffset;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(sa + (wave * A_INSTR + j) * 1024), 16,
                                               (int)off, 0, 0, 0);
    }
  };
  // after the last piece of a step: advance the scalar tap state of the NHWC gather
  auto issue_done = [&](int kt) __attribute__((always_inline)) {
    if (DUAL && kt >= nk1) {                  // next k-tile of the second operand: same tap, or the next tap of its window
      if (++s_k2 == nk2c) {
        s_k2 = 0;
        if (++s_dx == pa.a2_window) { s_dx = 0; s_tapoff += (unsigned)((pa.a2_W - pa.a2_window + 1) * pa.lda2 * 2); }
        else s_tapoff += (unsigned)(pa.lda2 * 2);
      }
    }
    if (KMODE == K_CONV && pa.cin64) {        // (DUAL: the steps behind the last filter tap run past it harmlessly -- setup_issue resets the state per tile)
      s_ci0 += BK;
      if (s_ci0 >= p.Cin) {
        s_ci0 = 0;
        if (++s_kx == p.KW) { s_kx = 0; ++s_ky; }
      }
    }
  };
  // gate values of k-tile `kt` of the tile being issued -> gate slice of `stage`: wave w < s_gp brings slots 4 w .. 4 w + 3,
  // lane l the floats [4 (l % 16), +4) of slot 4 w + l / 16.  Anything outside the table (k >= K, image >= B, slot >= s_slots,
  // no tile) has an out-of-range offset and lands as zeros.
  auto issue_gate = [&](int kt, int stage) __attribute__((always_inline)) {
    if (wave < pa.s_gp) {
      const int slot = wave * 4 + (lane >> 4);
      const int k = kt * BK + (lane & 15) * 4;
      const int img = s_b0 + slot;
      const bool ok = s_b0 >= 0 && slot < pa.s_slots && k < p.K && (int64_t)img * p.rows_per_image < p.M;
      const unsigned off = ok ? (unsigned)(((size_t)img * p.K + k) * 4) : kOobOffset;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_s, (lds_ptr_t)(sS + stage * s_gbytes + wave * 1024), 16, (int)off, 0, 0, 0);
    }
  };
  auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < N_PIECES; ++q) issue_piece(q, kt, stage);
    if (SCALE) issue_gate(kt, stage);
    issue_done(kt);
  };

  const int frow = lane & 31;
  const int fhi = lane >> 5;

  // epilogue staging block of this wave
  constexpr int SLOTS = WTN / 4;                  // 16-byte slots per staged row
  constexpr int LPR = WTN / 8;                    // lanes per row at read-back (8 outputs each)
  constexpr int RPI = 64 / LPR;                   // rows per read-back iteration
  constexpr int ITS = 32 / RPI;                   // read-back iterations per 32-row pass
  auto epi_slot = [](int row, int slot) -> int {  // physical slot (conflict-free b128 writes and reads)
    return SLOTS == 16 ? (slot ^ (row & 15)) : (slot ^ ((row >> 1) & 7));
  };
  const ActParams actp = make_act(p.act);
  const int e_row = lane / LPR;
  const int e_c8 = lane % LPR;
  const bool has_res = p.residual != nullptr;
  // Without a residual the whole epilogue arithmetic can run on the accumulators BEFORE the transpose through
  // LDS: the staged block is bf16 (half the LDS write traffic, which is the slow direction at ~80 B/clk) and the
  // read-back feeds the stores directly -- no VALU work between LDS and the store.
  constexpr bool fast_epi = VEC && NORES;          // a kernel flavour of its own: both epilogues in one kernel spill

  // ---- prime the pipeline
  int iss_tile = t_first, iss_kt = 1;
  setup_issue(iss_tile, true);
  issue(0, 0);
  int cur = 0;
  bool stores_pending = false;   // the previous step ended an interior tile: its stores may still be in flight
  int stamp_i = 0;
  auto stamp = [&]() __attribute__((always_inline)) {
    if ((TFIMM_PROBE(pa.dbg) & 64) && blockIdx.x == 0 && lane == 0 && stamp_i < 60) pa.dbg_ptr[wave * 64 + stamp_i++] = __builtin_readcyclecounter();
  };

  for (int tile = t_first; tile < t_hi; tile += t_step) {
    int mt, nt;
    tile_mn(tile, mt, nt);
    const int m0 = mt * BM, n0 = nt * BN;
    const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N);
    const int e_n = n0 + wn * WTN + e_c8 * 8;        // first of this lane's 8 output channels
    const int e_m = m0 + wm * WTM + e_row;           // output row at (pass 0, iteration 0)

    // VEC epilogue state of this tile.  A lane owns 8 consecutive channels (e_n ..) of the rows
    // e_m + d, d = i * 32 + it * RPI (compile-time steps < 128).  Byte offsets of (row e_m, channel
    // e_n) in the output / residual are computed once; a step adds the uniform d * ld and, where the
    // row remap (patch rows -> token rows) or the residual period (pos_embed broadcast over images)
    // wraps inside the tile, one uniform correction (host guarantees remap_in, res_mod are 0 or >= 128,
    // so at most one wrap).  Rows >= M and channels >= N need no test: the descriptors end at the
    // last valid element, so those lanes' offsets are out of range (loads 0, stores dropped).
    tfimm_f32x2 bias2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bias2[e] = tfimm_f32x2{0.f, 0.f};
    unsigned out_off0 = kOobOffset, res_off0 = kOobOffset;
    int or0 = 0, rm0 = 0;
    const int remap_eff = p.remap_in > 0 ? p.remap_in : 0x7fffffff;
    const int resmod_eff = p.res_mod > 0 ? p.res_mod : 0x7fffffff;
    const unsigned ldc2 = (unsigned)p.ldc * 2u, ldr2 = (unsigned)p.ldr * 2u;
    const unsigned out_wrap = p.remap_in > 0 ? (unsigned)(p.remap_out - p.remap_in) * ldc2 : 0u;
    const unsigned res_wrap = p.res_mod > 0 ? (unsigned)p.res_mod * ldr2 : 0u;
    // fast epilogue: lane l < WTN/4 fetches bias quad l of the wave's columns here (a load issued in the epilogue would
    // have to wait behind the next step's DMA), parks it in LDS there
    f32x4 bfast = {0.f, 0.f, 0.f, 0.f};
    if (fast_epi && lane < WTN / 4) {
      const int n = n0 + wn * WTN + lane * 4;
      if (p.bias && n < p.N) {
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
        bfast = f32x4{b4.x, b4.y, b4.z, b4.w};
      }
    }
    if (LNIN) {
      // rows m0 + wm*WTM + 2*lane, +1: 16 bytes per lane; rows >= M and columns >= N are beyond the descriptors (zeros)
      const unsigned so = (unsigned)(m0 + wm * WTM) * 8u + (unsigned)lane * 16u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_s, (lds_ptr_t)lnw, 16, (int)so, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const unsigned co = (unsigned)(((n0 + wn * WTN + j * 32 + (lane & 31)) * 2 + (lane >> 5)) * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_c, (lds_ptr_t)(lnw + (1 + j) * 1024), 16, (int)co, 0, 0, 0);
      }
    }
    if (VEC) {
      const bool col_ok = e_n < p.N;   // N % 8 == 0: all 8 channels or none
      if (!fast_epi && p.bias && col_ok) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + e_n);
        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + e_n + 4);
        bias2[0] = tfimm_f32x2{b0.x, b0.y}; bias2[1] = tfimm_f32x2{b0.z, b0.w};
        bias2[2] = tfimm_f32x2{b1.x, b1.y}; bias2[3] = tfimm_f32x2{b1.z, b1.w};
      }
      const int em = e_m < p.M ? e_m : p.M;          // clamp: offsets stay inside 32 bits
      rm0 = p.res_mod > 0 ? em % p.res_mod : em;
      const int oq0 = p.remap_in > 0 ? em / p.remap_in : 0;
      or0 = p.remap_in > 0 ? em - oq0 * p.remap_in : em;
      const int om0 = p.remap_in > 0 ? oq0 * p.remap_out + or0 + p.remap_off : em;
      if (col_ok) {
        out_off0 = (unsigned)(((size_t)om0 * p.ldc + e_n) * 2);
        res_off0 = (unsigned)(((size_t)rm0 * p.ldr + e_n) * 2);
      }
    }
    // residual segment of (pass i, iteration it).  Issued unconditionally: without a residual the
    // descriptor has zero records and every lane reads 0 -- no VMEM inside a branch, so hipcc's vmcnt
    // bookkeeping through the epilogue stays exact.
    uint4 rres[ITS];
    auto load_res1 = [&](int i, int it) __attribute__((always_inline)) {
      const int d = i * 32 + it * RPI;
      unsigned off = res_off0 + (unsigned)d * ldr2;
      off -= (rm0 + d >= resmod_eff) ? res_wrap : 0u;
#ifdef TFIMM_STREAM_DBG   // probe build: TFIMM_GEMM_DBG & 2 = every residual load out of range (returns zeros, no memory access)
      if (TFIMM_PROBE(pa.dbg) & 2) off = kOobOffset;
#endif
      rres[it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, (int)off, 0, 0));
    };

    // SCALE: the gate is applied to the A tile IN LDS, once per element and k-tile, by all waves together between the barrier that
    // publishes the stage and a second one in front of the fragment reads (kstep).  (Rounds 1-5 scaled the FRAGMENTS in
    // registers: every wave column repeated the work on the same rows -- four times in the 256 x 256 tile -- 20 VALU operations
    // and 32 bytes of gate reads per 16-byte fragment inside the MFMA loop, which made the SE projections of EfficientNet
    // VALU-bound at ~320 TFLOP/s; and hipcc, seeing plain LDS reads of the gate region next to LDS-DMA writes it could not tell
    // apart, drained the prefetch of the next k-tile with vmcnt(0) right behind its issue.)
    // A thread owns SC_IT 16-byte chunks of the tile: physical chunk index it * NT + tid; its row's gate slot and the chunk's
    // first k inside a k-tile are fixed for the whole tile.  The gate values themselves arrive per k-tile (issue_gate): 1 KiB per
    // four image slots and stage instead of whole gate rows per tile, so every tile shape fits in LDS whatever K is (rounds 1-5:
    // two buffers of slots x K floats -- 43 KB for K = 1632 at 12 x 12 pixels -- sent EfficientNet-B4's last two stages to the
    // register-staged kernel).
    constexpr int NT = NW * 64;
    constexpr int SC_IT = BM * 8 / NT;
    static_assert(!SCALE || (SC_IT >= 1 && SC_IT * NT == BM * 8), "scale pass: whole chunks per thread");
    unsigned sc_goff[SCALE ? SC_IT : 1];       // byte offset into a k-tile's gate slice: (image slot) * 256 + 4 * first k of the chunk
    if (SCALE) {
      const int b0 = m0 / p.rows_per_image;
#pragma unroll
      for (int it = 0; it < SC_IT; ++it) {
        const int idx = it * NT + tid, row = idx >> 3;
        const int chunk = (idx & 7) ^ ((row >> 1) & 7);             // logical k-chunk behind this physical slot (lds_slot)
        const int m = m0 + row;
        sc_goff[it] = (unsigned)(((m < p.M ? m : p.M - 1) / p.rows_per_image - b0) * 256 + chunk * 32);
      }
    }
    const unsigned sc_gbase = (unsigned)(size_t)(lds_ptr_t)sS;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // one step of the flattened (tile, k-tile) pipeline; `last` = final k-tile of this tile
    auto kstep = [&](bool last, int kt_cur) __attribute__((always_inline)) {
      // This wave's DMA pieces of the current step must have landed.  VMEM operations retire in
      // issue order, and the only ones younger than that DMA are the previous tile's epilogue
      // (>= TM*ITS store instructions for an interior tile): leave exactly those in flight.
#ifdef TFIMM_STREAM_DBG   // probe build: TFIMM_GEMM_DBG & 1 = no counted wait (every step drains all VMEM)
      if (VEC && stores_pending && !(TFIMM_PROBE(pa.dbg) & 1)) {
#else
      if (VEC && stores_pending) {
#endif
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TM * ITS) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      stores_pending = false;
      tfimm_lds_reuse_barrier();                         // ... everyone's; stage cur^1 (and an aliased epilogue block in it) is
                                                         //     free again: every wave's reads of it are COMPLETE (common.h)
      // first residual rows of THIS tile: requested ahead of the next step's DMA, so the wait for
      // them in the epilogue leaves that DMA in flight
      if (VEC && last && !fast_epi) {
#pragma unroll
        for (int it = 0; it < ITS; ++it) load_res1(0, it);
      }
      // issue the next step: next k-tile of this tile, or k-tile 0 of this workgroup's next tile
      if (iss_kt == nk) {
        iss_tile += t_step;
        iss_kt = 0;
        setup_issue(iss_tile, iss_tile < t_hi);
      }
      issue(iss_kt, cur ^ 1);
      ++iss_kt;

      if (SCALE) {
        // a[m][k] * gate[image(m)][k], rounded to bf16 once, written back in place.  Explicit ds_* instructions: the DMA
        // just issued stays in flight (see above).  Two chunks at a time keep the pass inside the register budget.
        const unsigned sa_addr = (unsigned)(size_t)(lds_ptr_t)(smem + cur * STAGE) + (unsigned)tid * 16u;
        const unsigned gk = sc_gbase + (unsigned)(cur * s_gbytes);
        constexpr int PAIR = SC_IT >= 2 ? 2 : 1;
#pragma unroll
        for (int it0 = 0; it0 < SC_IT; it0 += PAIR) {
          u32x4 av[PAIR];
          f32x4 g0[PAIR], g1[PAIR];
          // reads and their wait in ONE asm statement: the outputs of an asm are "ready" for hipcc the moment the statement
          // ends, a separate s_waitcnt statement does not hold back the arithmetic on them
          const unsigned aa0 = sa_addr + (unsigned)(it0 * NT * 16), ga0 = gk + sc_goff[it0];
          if (PAIR == 2) {
            const unsigned aa1 = sa_addr + (unsigned)((it0 + PAIR - 1) * NT * 16), ga1 = gk + sc_goff[it0 + PAIR - 1];
            asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %7 offset:16\n\t"
                         "ds_read_b128 %3, %8\n\tds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:16\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(av[0]), "=&v"(g0[0]), "=&v"(g1[0]), "=&v"(av[PAIR - 1]), "=&v"(g0[PAIR - 1]), "=&v"(g1[PAIR - 1])
                         : "v"(aa0), "v"(ga0), "v"(aa1), "v"(ga1) : "memory");
          } else {
            asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %4 offset:16\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(av[0]), "=&v"(g0[0]), "=&v"(g1[0]) : "v"(aa0), "v"(ga0) : "memory");
          }
#pragma unroll
          for (int u = 0; u < PAIR; ++u) {
            tfimm_f32x2 v[4];
            unpack8p(make_uint4(av[u][0], av[u][1], av[u][2], av[u][3]), v);
            v[0] *= tfimm_f32x2{g0[u][0], g0[u][1]}; v[1] *= tfimm_f32x2{g0[u][2], g0[u][3]};
            v[2] *= tfimm_f32x2{g1[u][0], g1[u][1]}; v[3] *= tfimm_f32x2{g1[u][2], g1[u][3]};
            const uint4 o = pack8p(v);
            const u32x4 ov = {o.x, o.y, o.z, o.w};
            const unsigned aa = sa_addr + (unsigned)((it0 + u) * NT * 16);
            asm volatile("ds_write_b128 %0, %1" ::"v"(aa), "v"(ov) : "memory");
          }
        }
        tfimm_lds_reuse_barrier();        // every wave's scaled chunks are in LDS (lgkmcnt(0) + s_barrier)
      }

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
      cur ^= 1;
    };

    stamp();
    for (int kt = 0; kt + 1 < nk; ++kt) kstep(false, kt);
    kstep(true, nk - 1);
    stamp();

    // ---- epilogue (per wave).  Aliased staging lives in the stage consumed last (index cur^1 now):
    //      wait until every wave is done reading it; the next refill of that stage is issued only
    //      after the next step's barrier, i.e. after every wave finished this epilogue.
    float* sEw;
    if (G::EPI_ALIAS) {
      tfimm_lds_reuse_barrier();     // (with every fragment read of this wave COMPLETE: see common.h)
      sEw = reinterpret_cast<float*>(smem + (cur ^ 1) * STAGE + wave * G::EPI_WAVE);
    } else {
      sEw = reinterpret_cast<float*>(smem + 2 * STAGE + wave * G::EPI_WAVE);
    }
    // a wave whose 32/64 columns all lie beyond N has nothing to store (ragged N: the last column tile); such a
    // tile is never `interior`, so the counted wait of the next step does not expect its stores
    if (fast_epi) {
      if (n0 + wn * WTN < p.N) {
        constexpr int CPR = WTN / 8;                 // 16-byte chunks per staged bf16 row
        char* const sE16 = reinterpret_cast<char*>(sEw);
        // in the accumulator layout a lane owns channels j*32 + q*8 + fhi*4 .. +3 of its rows.  The wave's WTN bias
        // values go through the unused half of its staging block (the staged rows are bf16 here), read back as
        // quads where they are needed: holding all 8 quads in registers spills the main loop
        float* const bs = reinterpret_cast<float*>(sE16 + 32 * WTN * 2);
        if (lane < WTN / 4) {
          const unsigned baddr = (unsigned)(size_t)(lds_ptr_t)(bs + lane * 4);
          asm volatile("ds_write_b128 %0, %1" ::"v"(baddr), "v"(bfast) : "memory");
        }
        // every LDS read of this epilogue is an explicit ds_read: before an LDS access it can see, hipcc waits for ALL
        // outstanding LDS-DMA (vmcnt(0)), i.e. for the next step's prefetch
        static_assert(TN == 1 || TN == 2, "fast epilogue: 32 or 64 columns per wave");
        u32x4 bq[TN * 4];
        {
          const unsigned ba = (unsigned)(size_t)(lds_ptr_t)(bs + fhi * 4);
          if (TN == 2) {
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:32\n\tds_read_b128 %2, %8 offset:64\n\t"
                         "ds_read_b128 %3, %8 offset:96\n\tds_read_b128 %4, %8 offset:128\n\tds_read_b128 %5, %8 offset:160\n\t"
                         "ds_read_b128 %6, %8 offset:192\n\tds_read_b128 %7, %8 offset:224\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(bq[0]), "=&v"(bq[1]), "=&v"(bq[2]), "=&v"(bq[3]), "=&v"(bq[TN * 4 - 4]), "=&v"(bq[TN * 4 - 3]),
                           "=&v"(bq[TN * 4 - 2]), "=&v"(bq[TN * 4 - 1])
                         : "v"(ba) : "memory");
          } else {
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\t"
                         "ds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(bq[0]), "=&v"(bq[1]), "=&v"(bq[2]), "=&v"(bq[3]) : "v"(ba) : "memory");
          }
        }
        const int wsw = CPR == 8 ? ((frow >> 1) & 7) : ((frow >> 2) & 3);    // chunk swizzle of this lane's row
        unsigned rb_addr[ITS];
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          const int pr = it * RPI + e_row;
          const int rsw = CPR == 8 ? ((pr >> 1) & 7) : ((pr >> 2) & 3);
          rb_addr[it] = (unsigned)(size_t)(lds_ptr_t)(sE16 + pr * (WTN * 2) + ((e_c8 ^ rsw) * 16));
        }
        u32x4 cfr[TN];
        if (LNIN) {
          // with a single k-tile no later step has waited for this tile's table DMA yet
          if (nk == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          const unsigned ca = (unsigned)(size_t)(lds_ptr_t)(lnw + 1024 + lane * 16);
          if (TN == 2) {
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(cfr[0]), "=&v"(cfr[TN - 1]) : "v"(ca) : "memory");
          } else {
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(cfr[0]) : "v"(ca) : "memory");
          }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          tfimm_f32x2 rs2 = {1.f, 1.f};
          if (LNIN) {
            // (mean, rstd) of this lane's row of pass i; -mean as three bf16 terms (exact residuals) in the k-slots that meet
            // the (ca, cb, cc) pattern of the column fragments: k0..8 = m1 ca, m1 cb, m1 cc, m2 ca, m2 cb, m2 cc, m3 ca, m3 cb, m3 cc
            tfimm_f32x2 st;
            const unsigned sa = (unsigned)(size_t)(lds_ptr_t)(lnw + (i * 32 + frow) * 8);
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
            // a real register pair: folded into the FMAs as an operand select (v_pk_fma_f32 ... op_sel:[0,1,0] on the
            // (mean, rstd) pair the ds_read_b64 returned) the LOW products came out as 0 * rstd for runs of 8 lanes, racily, on
            // the 32-column wave tiles (tools/ln_tile_probe.py: 8..20 bad launches of 20; none with the pair materialised)
            asm volatile("" : "+v"(rs2));
          }
          if (TN == 2) {
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[i][0]), "+v"(acc[i][TN - 1]));
          } else {
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[i][0]));
          }
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
                const unsigned addr = (unsigned)(size_t)(lds_ptr_t)(sE16 + frow * (WTN * 2) + ((chunk ^ wsw) * 16) + fhi * 8);
                const uint2 w2 = h2 ? make_uint2(pk.z, pk.w) : make_uint2(pk.x, pk.y);
                asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(w2) : "memory");
              }
            }
          u32x4 o16[ITS];
          if (ITS == 4) {
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(o16[0]), "=&v"(o16[1]), "=&v"(o16[ITS - 2]), "=&v"(o16[ITS - 1])
                         : "v"(rb_addr[0]), "v"(rb_addr[1]), "v"(rb_addr[ITS - 2]), "v"(rb_addr[ITS - 1]) : "memory");
          } else {
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(o16[0]), "=&v"(o16[1]) : "v"(rb_addr[0]), "v"(rb_addr[1]) : "memory");
          }
#pragma unroll
          for (int it = 0; it < ITS; ++it) {
            const int d = i * 32 + it * RPI;
            unsigned off = out_off0 + (unsigned)d * ldc2;
            off += (or0 + d >= remap_eff) ? out_wrap : 0u;
            __builtin_amdgcn_raw_buffer_store_b128(o16[it], rsrc_o, (int)off, 0, 0);
          }
        }
      }
    } else
    if (!(VEC && n0 + wn * WTN >= p.N))
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      // The staging stores below are inline asm, and hipcc pads no hazards inside or in front of an
      // asm statement: an MFMA result needs up to 18 wait states before a DS instruction may read it
      // (XDL write VGPR -> LDS read, 16-pass case).  Naming this pass's accumulators as operands
      // orders the pad behind the MFMAs that produce them.
      if (TN == 2) {
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[i][0]), "+v"(acc[i][TN - 1]));
      } else {
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[i][0]));
      }
      // lane holds output row (frow) x 4 consecutive channels per accumulator quad
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int slot = j * 8 + q * 2 + fhi;
          const f32x4 v = {acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
          // Written with an explicit ds_write: hipcc orders every LDS store it can see behind ALL
          // outstanding LDS-DMA (s_waitcnt vmcnt(0)), which would drain the next step's prefetch at
          // every epilogue.  The staging block never overlaps the stage that DMA is filling, and
          // DS operations of one wave execute in order, so the read-back below needs no wait.
          const unsigned addr = (unsigned)(size_t)(lds_ptr_t)(&sEw[frow * WTN + epi_slot(frow, slot) * 4]);
          asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
        }
      if (i == 0) stamp();
      if (VEC) {
        // read back row-contiguous: lane handles 8 consecutive channels of one row
        // (explicit ds_reads, two iterations per wait: an LDS read hipcc can see waits for the next step's DMA)
        static_assert(ITS % 2 == 0, "read-back iterations come in pairs");
#pragma unroll
        for (int ip = 0; ip < ITS; ip += 2) {
        f32x4 st[4];
        {
          unsigned ra[4];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int pr = (ip + u) * RPI + e_row;
            ra[2 * u] = (unsigned)(size_t)(lds_ptr_t)(&sEw[pr * WTN + epi_slot(pr, 2 * e_c8) * 4]);
            ra[2 * u + 1] = (unsigned)(size_t)(lds_ptr_t)(&sEw[pr * WTN + epi_slot(pr, 2 * e_c8 + 1) * 4]);
          }
          asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(st[0]), "=&v"(st[1]), "=&v"(st[2]), "=&v"(st[3])
                       : "v"(ra[0]), "v"(ra[1]), "v"(ra[2]), "v"(ra[3]) : "memory");
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int it = ip + u;
          const f32x4 lo = st[2 * u], hi = st[2 * u + 1];
          tfimm_f32x2 v[4] = {{lo[0], lo[1]}, {lo[2], lo[3]}, {hi[0], hi[1]}, {hi[2], hi[3]}};
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bias2[e];
          const uint4 rraw = rres[it];
          // the next pass's segment for this slot: requested before this iteration's store and consumed
          // a pass later, so its return never sits behind a store acknowledgement
          if (i + 1 < TM) load_res1(i + 1, it);
          // wave-uniform branches around pure VALU work (the empty asm keeps them branches)
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
          unsigned off = out_off0 + (unsigned)d * ldc2;
          off += (or0 + d >= remap_eff) ? out_wrap : 0u;
          // (default cache policy: non-temporal stores are 3-8 % faster for this launch alone but the
          // next layer then misses L2/MALL on what it reads first -- ResNet-50 end to end -1.5 %)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pack8p(v)), rsrc_o, (int)off, 0, 0);
        }
        }
      } else {
        // catch-all: element loop over the staged block (the wave's own LDS, in order -> no barrier)
        for (int idx = lane; idx < 32 * WTN; idx += 64) {
          const int pr = idx / WTN, c = idx - pr * WTN;
          const int m = m0 + wm * WTM + i * 32 + pr, n = n0 + wn * WTN + c;
          if (m < p.M && n < p.N) {
            float v = sEw[pr * WTN + epi_slot(pr, c >> 2) * 4 + (c & 3)];
            if (p.bias) v += p.bias[n];
            float r = 0.f;
            if (has_res) {
              const int rm = p.res_mod > 0 ? (m % p.res_mod) : m;
              r = bf2f(p.residual[(size_t)rm * p.ldr + n]);
            }
            if (p.act_after_res) v += r;
            v = act1(v, actp);
            if (!p.act_after_res) v += r;
            const int om = p.remap_in > 0 ? (m / p.remap_in) * p.remap_out + (m % p.remap_in) + p.remap_off : m;
            if (p.out_f32) reinterpret_cast<float*>(p.out)[(size_t)om * p.ldc + n] = v;
            else reinterpret_cast<bf16_t*>(p.out)[(size_t)om * p.ldc + n] = (bf16_t)f2bf(v);
          }
        }
      }
      stamp();
    }
    stores_pending = VEC && interior;
  }
  // the last step's (all out-of-range) prefetch must have landed before this workgroup's LDS is released
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef TFIMM_STREAM_DBG   // probe build: TFIMM_GEMM_DBG & 4 = all waves leave together, after everything of every wave has landed
  if (TFIMM_PROBE(pa.dbg) & 4) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#endif
}

struct StreamTileCfg {
  int bm, bn, threads, lds_bytes;
  gemm_stream_fn fn[2][3];     // [K_DENSE, K_CONV][catch-all, VEC, VEC without residual]
  gemm_stream_fn fn_scale[3];  // K_DENSE + SE gate on A, same three epilogues (null: not built for this tile)
  gemm_stream_fn fn_dual[2];   // [K_DENSE, K_CONV] + a second dense A operand, residual-free vector epilogue (null: not built for this tile)
  gemm_stream_fn fn_ln;        // K_DENSE, VEC without residual, LayerNorm folded in (LNIN); needs ln_lds extra bytes of LDS
  int ln_lds;
};

}  // namespace tfimm_gemm

// persistent tile shapes: id, BM, BN, WAVES_M, WAVES_N
#define TFIMM_GEMM_STREAM_TILES(X) \
  X(0, 256, 256, 2, 4)             \
  X(1, 256, 128, 4, 2)             \
  X(2, 128, 128, 2, 2)             \
  X(3, 256, 64, 4, 2)              \
  X(4, 128, 64, 2, 2)              \
  X(5, 128, 256, 2, 4)             \
  X(6, 256, 64, 4, 1)             \
  X(8, 256, 32, 4, 1)
// id 7 = the 256x256 deep-ring schedule (gemm_pipe_kernel.h), instantiated on its own; id 8 = narrow outputs
// (N <= 32 per tile: the 24..48-channel layers of EfficientNet / MobileNet); id 9 = 256x128 with two co-resident
// four-wave workgroups per CU (gemm_duo_kernel.h), instantiated on its own
#define TFIMM_GEMM_STREAM_NUM_TILES 10
