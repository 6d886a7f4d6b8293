// tfimm_hip_gemm host side: descriptor validation, tile selection, launch.
// Kernel: gemm_kernel.h; per-tile instantiations: gemm_inst.hip.
#include "gemm_stream_kernel.h"

#include <algorithm>
#include <cstdlib>

using namespace tfimm_gemm;

#define TFIMM_DECL(ID, BM_, BN_, WM_, WN_) extern "C" const TileCfg tfimm_gemm_tile_##ID;
TFIMM_GEMM_TILES(TFIMM_DECL)
#undef TFIMM_DECL
#define TFIMM_DECL(ID, BM_, BN_, WM_, WN_) extern "C" const DmaTileCfg tfimm_gemm_dma_tile_##ID;
TFIMM_GEMM_DMA_TILES(TFIMM_DECL)
#undef TFIMM_DECL
#define TFIMM_DECL(ID, BM_, BN_, WM_, WN_) extern "C" const StreamTileCfg tfimm_gemm_stream_tile_##ID;
TFIMM_GEMM_STREAM_TILES(TFIMM_DECL)
#undef TFIMM_DECL
extern "C" const StreamTileCfg tfimm_gemm_stream_tile_7;
extern "C" const StreamTileCfg tfimm_gemm_stream_tile_9;

namespace {

const TileCfg* tile_table(int i) {
#define TFIMM_CASE(ID, BM_, BN_, WM_, WN_) \
  case ID: return &tfimm_gemm_tile_##ID;
  switch (i) {
    TFIMM_GEMM_TILES(TFIMM_CASE)
    default: return nullptr;
  }
#undef TFIMM_CASE
}

const DmaTileCfg* dma_tile_table(int i) {
#define TFIMM_CASE(ID, BM_, BN_, WM_, WN_) \
  case ID: return &tfimm_gemm_dma_tile_##ID;
  switch (i) {
    TFIMM_GEMM_DMA_TILES(TFIMM_CASE)
    default: return nullptr;
  }
#undef TFIMM_CASE
}

const StreamTileCfg* stream_tile_table(int i) {
#define TFIMM_CASE(ID, BM_, BN_, WM_, WN_) \
  case ID: return &tfimm_gemm_stream_tile_##ID;
  switch (i) {
    TFIMM_GEMM_STREAM_TILES(TFIMM_CASE)
    case 7: return &tfimm_gemm_stream_tile_7;
    case 9: return &tfimm_gemm_stream_tile_9;
    default: return nullptr;
  }
#undef TFIMM_CASE
}

int g_num_cu = 0;

int num_cu() {
  if (g_num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      g_num_cu = prop.multiProcessorCount;
    if (g_num_cu <= 0) g_num_cu = 256;
  }
  return g_num_cu;
}

// tile ids: 0 128x128, 1 128x64, 2 64x64, 3 256x128, 4 128x256, 5 64x128
int pick_tile(const tfimm_gemm_desc& d, int kmode) {
  if (kmode == K_DENSE_SCALAR || kmode == K_CONV_SCALAR) return 2;
  if (d.tile_hint > 0 && d.tile_hint <= TFIMM_GEMM_NUM_TILES && tile_table(d.tile_hint - 1)->fn[kmode])
    return d.tile_hint - 1;
  const int cus = num_cu();
  const int64_t M = d.M, N = d.N;
  auto blocks = [&](int bm, int bn) { return cdiv64(M, bm) * cdiv64(N, bn); };
  // prefer the biggest tile that still gives every CU >= 2 blocks
  if (N > 64 && blocks(128, 128) >= 2 * cus) return 0;
  if (N <= 64 && blocks(128, 64) >= 2 * cus) return 1;
  if (N > 64 && blocks(64, 128) >= cus) return 5;
  if (blocks(128, 64) >= 2 * cus) return 1;
  return 2;
}

// LDS-DMA tile ids: 0 256x256, 1 256x128, 2 128x128, 3 256x64, 4 128x64, 5 128x256.
// Score = relative kernel efficiency x useful fraction of the padded tile area x fill of the last
// wave of blocks.  Efficiency weights come from the measured table in DESIGN.md.
int pick_dma_tile(const tfimm_gemm_desc& d) {
  if (d.tile_hint > 10 && d.tile_hint <= 10 + TFIMM_GEMM_DMA_NUM_TILES) return d.tile_hint - 11;
  static const double eff[TFIMM_GEMM_DMA_NUM_TILES] = {1.00, 0.90, 0.70, 0.70, 0.50, 0.90};
  static const int occ[TFIMM_GEMM_DMA_NUM_TILES] = {1, 1, 2, 1, 3, 1};
  const int cus = num_cu();
  int best = 2;
  double best_score = -1.0;
  for (int i = 0; i < TFIMM_GEMM_DMA_NUM_TILES; ++i) {
    const DmaTileCfg* t = dma_tile_table(i);
    const double tm = (double)cdiv64(d.M, t->bm), tn = (double)cdiv64(d.N, t->bn);
    const double useful = ((double)d.M * d.N) / (tm * t->bm * tn * t->bn);
    const double blocks = tm * tn, slots = (double)cus * occ[i];
    const double waves = (double)cdiv64((int64_t)blocks, (int64_t)slots);
    const double fill = blocks / (waves * slots);
    const double score = eff[i] * useful * fill;
    if (score > best_score) {
      best_score = score;
      best = i;
    }
  }
  return best;
}

// Persistent (stream) family: same tile ids as the DMA family.  A persistent grid runs
// ceil(tiles / slots) rounds; score = efficiency x useful area x fill of those rounds.
int pick_stream_tile(const tfimm_gemm_desc& d, const int* occ) {
  if (d.tile_hint > 20 && d.tile_hint <= 20 + TFIMM_GEMM_STREAM_NUM_TILES) return d.tile_hint - 21;
#ifdef TFIMM_PROBE_HOOKS
  {
    // probe builds: TFIMM_GEMM_AUTO_TILE=k sends every launch WITHOUT a hint to stream tile k (0-based)
    static const int forced = getenv("TFIMM_GEMM_AUTO_TILE") ? atoi(getenv("TFIMM_GEMM_AUTO_TILE")) : -1;
    if (forced >= 0 && forced < TFIMM_GEMM_STREAM_NUM_TILES && forced != 7) return forced;
  }
#endif
  static const double eff[TFIMM_GEMM_STREAM_NUM_TILES] = {1.00, 0.90, 0.75, 0.70, 0.55, 0.90, 0.75, 0.0, 0.40, 0.0};
  const int cus = num_cu();
  int best = 2;
  double best_score = -1.0;
  for (int i = 0; i < TFIMM_GEMM_STREAM_NUM_TILES; ++i) {
    const StreamTileCfg* t = stream_tile_table(i);
    const double tm = (double)cdiv64(d.M, t->bm), tn = (double)cdiv64(d.N, t->bn);
    const double useful = ((double)d.M * d.N) / (tm * t->bm * tn * t->bn);
    const double blocks = tm * tn, slots = (double)cus * occ[i];
    const double rounds = (double)cdiv64((int64_t)blocks, (int64_t)slots);
    const double fill = blocks / (rounds * slots);
    const double score = eff[i] * useful * fill;
    if (score > best_score) {
      best_score = score;
      best = i;
    }
  }
  return best;
}

// bytes of weight panels a column-panel group may hold: 5/8 of the L2 of one XCD (hipDeviceProp_t::l2CacheSize; 4 MiB -> 2.5 MB)
int64_t l2_budget() {
  static int64_t v = 0;
  if (v == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    int64_t l2 = 4 << 20;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.l2CacheSize > 0) l2 = prop.l2CacheSize;
    if (l2 > (16 << 20)) l2 = 4 << 20;          // a runtime that reports the sum over the XCDs (or the MALL): the rule is per XCD
    v = l2 / 8 * 5;
  }
  return v;
}

bool env_flag(const char* name) {
  const char* e = getenv(name);
  return e && e[0] == '1';
}

bool stream_disabled() {
  static int v = -1;
  if (v < 0) v = env_flag("TFIMM_GEMM_NO_STREAM") ? 1 : 0;
  return v == 1;
}

bool strip_conv_enabled() {
  static int v = -1;
  if (v < 0) v = (getenv("TFIMM_STRIP_CONV") && atoi(getenv("TFIMM_STRIP_CONV")) == 0) ? 0 : 1;
  return v == 1;
}

bool dma_disabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("TFIMM_GEMM_NO_DMA");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

}  // namespace

// csrc/conv_strip.hip
int tfimm_launch_conv_strip(const tfimm_gemm::GemmArgs& g, int64_t a_bytes, int64_t w_bytes, int64_t out_bytes, int num_cu, hipStream_t stream);

extern "C" int tfimm_hip_gemm(const tfimm_gemm_desc* dp, void* stream) {
  if (!dp) TFIMM_FAIL(TFIMM_EINVAL, "gemm: null descriptor");
  const tfimm_gemm_desc& d = *dp;
  if (!d.a || !d.wt || !d.out) TFIMM_FAIL(TFIMM_EINVAL, "gemm: null a/wt/out pointer");
  if (d.M <= 0 || d.N <= 0 || d.K <= 0) TFIMM_FAIL(TFIMM_EINVAL, "gemm: M=%d N=%d K=%d", d.M, d.N, d.K);
  if (d.ldw < d.K || (d.ldw & 7)) TFIMM_FAIL(TFIMM_EINVAL, "gemm: ldw=%d must be >= K=%d and a multiple of 8", d.ldw, d.K);
  if (((uintptr_t)d.wt & 15)) TFIMM_FAIL(TFIMM_EINVAL, "gemm: wt must be 16-byte aligned");
  if (d.ldc < d.N) TFIMM_FAIL(TFIMM_EINVAL, "gemm: ldc=%d < N=%d", d.ldc, d.N);
  if (d.mode < 0 || d.mode > 2) TFIMM_FAIL(TFIMM_EINVAL, "gemm: mode=%d", d.mode);
  if (d.remap_in < 0 || d.res_mod < 0) TFIMM_FAIL(TFIMM_EINVAL, "gemm: negative remap/res_mod");
  if (d.bias && ((uintptr_t)d.bias & 15)) TFIMM_FAIL(TFIMM_EINVAL, "gemm: bias must be 16-byte aligned");
  if ((d.ln_stats != nullptr) != (d.ln_c1 != nullptr)) TFIMM_FAIL(TFIMM_EINVAL, "gemm: ln_stats and ln_c1 go together");
  if (d.ln_stats && (d.mode != TFIMM_A_DENSE || d.residual || d.a_scale || d.out_f32 || (((uintptr_t)d.ln_stats | (uintptr_t)d.ln_c1) & 15)))
    TFIMM_FAIL(TFIMM_EINVAL, "gemm: LayerNorm folding needs a dense bf16 layer without residual / gate and 16-byte aligned tables");

  if (d.a2) {
    // second A operand (ABI v4): the shortcut convolution of a residual block as further k-tiles of this GEMM
    if ((d.mode != TFIMM_A_DENSE && d.mode != TFIMM_A_CONV) || d.residual || d.a_scale || d.ln_stats || d.out_f32 || d.remap_in || d.res_mod)
      TFIMM_FAIL(TFIMM_EINVAL, "gemm: a second A operand needs a bf16 layer (dense or TFIMM_A_CONV) without residual / gate / LayerNorm / row remap");
    if (d.K2 <= 0 || (d.K2 & 7) || d.lda2 < d.K2 || (d.lda2 & 7) || ((uintptr_t)d.a2 & 15))
      TFIMM_FAIL(TFIMM_EINVAL, "gemm: a2 needs K2=%d %% 8 == 0, lda2=%d >= K2 and %% 8 == 0, a 16-byte aligned pointer", d.K2, d.lda2);
    if (d.a2_stride < 1 || d.a2_window < 0 || d.a2_window > 4) TFIMM_FAIL(TFIMM_EINVAL, "gemm: a2_stride=%d a2_window=%d", d.a2_stride, d.a2_window);
    const int a2w = d.a2_window > 1 ? d.a2_window : 1;
    if (d.a2_stride > 1 || a2w > 1) {
      if (d.a2_H <= 0 || d.a2_W <= 0 || d.a2_OH <= 0 || d.a2_OW <= 0 || d.M % ((int64_t)d.a2_OH * d.a2_OW) ||
          (d.a2_OH - 1) * d.a2_stride + a2w > d.a2_H || (d.a2_OW - 1) * d.a2_stride + a2w > d.a2_W)
        TFIMM_FAIL(TFIMM_EINVAL, "gemm: a2 geometry %dx%d -> %dx%d at stride %d, window %d (M=%d)", d.a2_H, d.a2_W, d.a2_OH, d.a2_OW, d.a2_stride, a2w, d.M);
    }
    const int64_t kp = cdiv64(d.K, 64) * 64;
    if (d.ldw < kp + (int64_t)(a2w * a2w - 1) * cdiv64(d.K2, 64) * 64 + d.K2)
      TFIMM_FAIL(TFIMM_EINVAL, "gemm: ldw=%d too small for K=%d + %d taps of K2=%d (each part padded to 64)", d.ldw, d.K, a2w * a2w, d.K2);
  }

  // The LDS-DMA kernels address every tensor through a buffer descriptor with a 32-bit byte offset.  A plain
  // dense GEMM whose activation, output or residual exceeds 2 GiB (EfficientNet-B4's first expand layer at batch 256:
  // 9.2 M rows x 144 channels) is therefore run as row chunks that each fit, instead of leaving those families.
  if (d.mode == TFIMM_A_DENSE && !d.a_scale && !d.a2 && d.remap_in == 0 && d.res_mod == 0) {
    const int64_t row_bytes = std::max<int64_t>(std::max<int64_t>((int64_t)d.lda * 2, (int64_t)d.ldc * (d.out_f32 ? 4 : 2)),
                                                 d.residual ? (int64_t)d.ldr * 2 : 0);
    const int64_t limit = 0x7fffff00LL;
    if (row_bytes > 0 && (int64_t)d.M * row_bytes > limit && row_bytes * 512 <= limit) {
      const int64_t chunk = (limit / row_bytes) / 256 * 256;
      for (int64_t m0 = 0; m0 < d.M; m0 += chunk) {
        tfimm_gemm_desc c = d;
        c.M = (int32_t)std::min<int64_t>(chunk, d.M - m0);
        c.a = (const char*)d.a + m0 * d.lda * 2;
        c.out = (char*)d.out + m0 * d.ldc * (d.out_f32 ? 4 : 2);
        if (d.residual) c.residual = (const char*)d.residual + m0 * d.ldr * 2;
        if (d.ln_stats) c.ln_stats = d.ln_stats + m0 * 2;
        const int rc = tfimm_hip_gemm(&c, stream);
        if (rc != 0) return rc;
      }
      return 0;
    }
  }

  GemmArgs g;
  g.a = (const bf16_t*)d.a; g.wt = (const bf16_t*)d.wt; g.bias = d.bias;
  g.residual = (const bf16_t*)d.residual; g.out = d.out; g.a_scale = d.a_scale;
  g.M = d.M; g.N = d.N; g.K = d.K;
  g.lda = d.lda; g.ldw = d.ldw; g.ldr = d.ldr; g.ldc = d.ldc;
  g.out_f32 = d.out_f32; g.act = d.act; g.act_after_res = d.act_after_res; g.res_mod = d.res_mod;
  g.remap_in = d.remap_in; g.remap_out = d.remap_out; g.remap_off = d.remap_off;
  g.B = d.B; g.H = d.H; g.W = d.W; g.Cin = d.Cin; g.KH = d.KH; g.KW = d.KW;
  g.KWp = (d.KW + 1) & ~1;
  g.stride = d.stride; g.pad_t = d.pad_t; g.pad_l = d.pad_l; g.OH = d.OH; g.OW = d.OW;
  g.stride_w = d.stride_w > 0 ? d.stride_w : d.stride;
  g.cpitch = d.pix_pitch > 0 ? d.pix_pitch : d.Cin;
  g.rows_per_image = d.rows_per_image;

  int kmode;
  if (d.mode == TFIMM_A_DENSE) {
    if (d.lda < d.K) TFIMM_FAIL(TFIMM_EINVAL, "gemm: lda=%d < K=%d", d.lda, d.K);
    const bool vec = ((d.lda & 7) == 0) && ((d.K & 7) == 0) && (((uintptr_t)d.a & 15) == 0);
    if (d.a_scale) {
      if (!vec || d.rows_per_image <= 0 || ((uintptr_t)d.a_scale & 15))
        TFIMM_FAIL(TFIMM_EINVAL, "gemm: a_scale needs aligned K %% 8 == 0 rows and rows_per_image > 0");
      kmode = K_DENSE_SCALE;
    } else {
      kmode = vec ? K_DENSE : K_DENSE_SCALAR;
    }
  } else {
    if (d.a_scale) TFIMM_FAIL(TFIMM_EINVAL, "gemm: a_scale only in dense mode");
    if (d.B <= 0 || d.H <= 0 || d.W <= 0 || d.KH <= 0 || d.KW <= 0 || d.stride <= 0 || d.OH <= 0 || d.OW <= 0)
      TFIMM_FAIL(TFIMM_EINVAL, "gemm: bad conv geometry");
    if ((int64_t)d.B * d.OH * d.OW != d.M) TFIMM_FAIL(TFIMM_EINVAL, "gemm: M != B*OH*OW");
    if (((uintptr_t)d.a & 15)) TFIMM_FAIL(TFIMM_EINVAL, "gemm: conv input must be 16-byte aligned");
    if (d.mode == TFIMM_A_CONV) {
      if (d.K != d.KH * d.KW * d.Cin) TFIMM_FAIL(TFIMM_EINVAL, "gemm: K != KH*KW*Cin");
      if (d.pix_pitch < 0 || (d.pix_pitch > 0 && d.pix_pitch < d.Cin))
        TFIMM_FAIL(TFIMM_EINVAL, "gemm: pix_pitch=%d < Cin=%d", d.pix_pitch, d.Cin);
      kmode = ((d.Cin | g.cpitch) & 7) ? K_CONV_SCALAR : K_CONV;  // odd channel counts: element loads
    } else {
      if (d.Cin != 4 || d.pix_pitch > 4) TFIMM_FAIL(TFIMM_EINVAL, "gemm: C4 mode needs Cin == 4 (and no pixel pitch)");
      if (d.K != d.KH * g.KWp * 4) TFIMM_FAIL(TFIMM_EINVAL, "gemm: K != KH*KWp*4 (K=%d)", d.K);
      kmode = K_CONV_C4;
    }
  }
  g.res_vec = d.residual ? (((d.ldr & 3) == 0) && (((uintptr_t)d.residual & 7) == 0)) : 0;
  g.res_vec16 = d.residual ? (((d.ldr & 7) == 0) && (((uintptr_t)d.residual & 15) == 0)) : 0;
  g.out_vec16 = ((d.ldc & 7) == 0) && (((uintptr_t)d.out & 15) == 0);
  if (d.out_f32)
    g.out_vec = ((d.ldc & 3) == 0) && (((uintptr_t)d.out & 15) == 0);
  else
    g.out_vec = ((d.ldc & 3) == 0) && (((uintptr_t)d.out & 7) == 0);

  // ---- 3x3 / stride 1 / pad 1, 128 -> 128 channels, rows of at most 31 pixels (ResNet-50 stage 2): the input-strip kernel
  //      (csrc/conv_strip.hip).  Tile hint 31 asks for it; hint 0 takes it unless TFIMM_STRIP_CONV=0 -- and only when its fixed
  //      128-pixel x 128-channel tiles give every CU at least one (M >= 128 CUs: it was measured at 28 x 28 and batch >= 6;
  //      below that the 128 x 64 / 256 x 32 implicit-GEMM tiles make more workgroups and the cost model decides); any other
  //      hint keeps the implicit-GEMM tiles (the tuner's candidates).
  if (kmode == K_CONV && d.KH == 3 && d.KW == 3 && d.stride == 1 && g.stride_w == 1 && d.pad_t == 1 && d.pad_l == 1 && d.OH == d.H &&
      d.OW == d.W && d.Cin == 128 && g.cpitch == 128 && d.N == 128 && !d.residual && !d.out_f32 && g.out_vec16 && d.remap_in == 0 &&
      !d.ln_stats && !d.a2 && d.W <= 31 && d.ldw >= d.K && (d.tile_hint == 31 || (d.tile_hint == 0 && strip_conv_enabled() && cdiv64(d.M, 128) >= num_cu()))) {
    const int64_t a_bytes = ((int64_t)d.B * d.H * d.W) * 128 * 2, w_bytes = (int64_t)d.N * d.ldw * 2;
    const int64_t out_bytes = ((int64_t)(d.M - 1) * d.ldc + d.N) * 2;
    if (a_bytes <= 0x7fffff00LL && w_bytes <= 0x7fffff00LL && out_bytes <= 0x7fffff00LL)
      return tfimm_launch_conv_strip(g, a_bytes, w_bytes, out_bytes, num_cu(), (hipStream_t)stream);
  }

  // ---- persistent LDS-DMA family (default): same operand requirements as the DMA family below
  {
    const int64_t a_bytes = (d.mode == TFIMM_A_DENSE) ? ((int64_t)(d.M - 1) * d.lda + d.K) * 2
                                                       : (((int64_t)d.B * d.H * d.W - 1) * g.cpitch + d.Cin) * 2;
    const int64_t w_bytes = (int64_t)d.N * d.ldw * 2;
    const bool hinted_other = d.tile_hint > 0 && d.tile_hint <= 20 && g.stride_w == g.stride && !d.ln_stats;
    // extents of the output / residual buffers as the epilogue addresses them (row remap included)
    const int64_t out_rows = d.remap_in > 0 ? ((int64_t)(d.M - 1) / d.remap_in) * d.remap_out + d.remap_in + d.remap_off : d.M;
    const int64_t out_bytes = ((out_rows - 1) * d.ldc + d.N) * (d.out_f32 ? 4 : 2);
    const int64_t res_rows = d.res_mod > 0 ? (d.res_mod < d.M ? d.res_mod : d.M) : d.M;
    const int64_t res_bytes = d.residual ? ((res_rows - 1) * d.ldr + d.N) * 2 : 0;
    const bool scale = kmode == K_DENSE_SCALE;
    const bool dual = d.a2 != nullptr;
    const int a2win = d.a2_window > 1 ? d.a2_window : 1;
    const int64_t a2_rows = !dual ? 0 : (d.a2_stride > 1 || a2win > 1) ? (int64_t)(d.M / ((int64_t)d.a2_OH * d.a2_OW)) * d.a2_H * d.a2_W : d.M;
    const int64_t a2_bytes = dual ? ((a2_rows - 1) * d.lda2 + d.K2) * 2 : 0;
    const bool ok = (kmode == K_DENSE || kmode == K_CONV || scale) && (!hinted_other || dual) && !stream_disabled() && !dma_disabled() &&
                    d.ldw >= (int)(cdiv64(d.K, 64) * 64) && a_bytes <= 0x7fffff00LL && w_bytes <= 0x7fffff00LL &&
                    out_bytes <= 0x7fffff00LL && res_bytes <= 0x7fffff00LL && a2_bytes <= 0x7fffff00LL &&
                    (!dual || ((kmode == K_DENSE || kmode == K_CONV) &&
                               d.ldw >= (int)(cdiv64(d.K, 64) * 64 + (int64_t)a2win * a2win * cdiv64(d.K2, 64) * 64)));
    if (ok) {
      const int fi = kmode == K_CONV ? 1 : 0;
      // vector epilogue: whole 16-byte groups per lane on aligned rows
      const int vi = ((d.N % 8) == 0 && !d.out_f32 && g.out_vec16 && (!d.residual || g.res_vec16) &&
                      (d.res_mod == 0 || d.res_mod >= 128) && (d.remap_in == 0 || d.remap_in >= 128)) ? 1 : 0;
      // epilogue flavour: 0 catch-all, 1 vector, 2 vector without residual (arithmetic in the accumulator layout)
      const int ei = vi ? (d.residual ? 1 : 2) : 0;
      // Workgroups per CU of every tile: the minimum over the epilogue flavours of an operand kind, queried for ALL
      // flavours on the first call.  (It used to be filled in flavour by flavour as launches came: the minimum -- and with
      // it the tile the heuristic picks for a shape without a table entry, and the grid -- depended on which OTHER layers
      // had been launched before, so the first forward of a model could differ in the last bits from its later ones:
      // tests/test_gpu_scored_batches.py under TFIMM_BRANCHES=2, cait_xxs24_224.)
      static int occ[TFIMM_GEMM_STREAM_NUM_TILES][2] = {};
      static tfimm_once_t ready;          // (attributes are per device; the occupancies are the same on every MI355X)
      if (ready.need()) {
        for (int f = 0; f < 2; ++f)
          for (int e = 0; e < 3; ++e)
            for (int i = 0; i < TFIMM_GEMM_STREAM_NUM_TILES; ++i) {
              const StreamTileCfg* t = stream_tile_table(i);
              if (!t->fn[f][e]) {         // the duo tile has no catch-all flavour (redirected below): two workgroups per CU
                if (occ[i][f] == 0) occ[i][f] = 2;
                continue;
              }
              TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)t->fn[f][e], hipFuncAttributeMaxDynamicSharedMemorySize, t->lds_bytes));
              int nb = 0;
              TFIMM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)t->fn[f][e], t->threads, (size_t)t->lds_bytes));
              nb = nb < 1 ? 1 : (nb > 4 ? 4 : nb);
              if (occ[i][f] == 0 || nb < occ[i][f]) occ[i][f] = nb;
            }
        ready.mark();
      }
      int occ_f[TFIMM_GEMM_STREAM_NUM_TILES];
      for (int i = 0; i < TFIMM_GEMM_STREAM_NUM_TILES; ++i) occ_f[i] = occ[i][fi];
      int ti = pick_stream_tile(d, occ_f);
      // the two-workgroups-per-CU tile: vector epilogues only, at least two 32-wide k-tiles; otherwise the 256x128 stream tile
      if (ti == 9 && (ei == 0 || d.K <= 32 || scale || dual)) ti = 1;
      const StreamTileCfg* t = stream_tile_table(ti);
      if (scale && !t->fn_scale[ei]) {   // the deep-ring tile has no SE-gate flavour
        ti = 0;
        t = stream_tile_table(ti);
      }
      if (dual) {
        if (ei != 2) TFIMM_FAIL(TFIMM_EUNSUP, "gemm: a second A operand needs N %% 8 == 0 and 16-byte aligned bf16 output rows");
        if (!t->fn_dual[fi]) {
          ti = 0;
          t = stream_tile_table(ti);
        }
      }
      const bool ln_in = d.ln_stats != nullptr;
      if (ln_in) {
        if (ei != 2 || fi != 0 || scale)
          TFIMM_FAIL(TFIMM_EUNSUP, "gemm: LayerNorm folding needs dense bf16 rows, N %% 8 == 0, 16-byte aligned output, no residual");
        if (!t->fn_ln || (size_t)t->lds_bytes + t->ln_lds > 160 * 1024) {
          ti = 0;
          t = stream_tile_table(ti);
        }
      }
      GemmStreamArgs ga;
      ga.g = g;
      ga.g.tiles_m = (int)cdiv64(d.M, t->bm);
      ga.g.tiles_n = (int)cdiv64(d.N, t->bn);
      ga.a_bytes = (unsigned)a_bytes;
      ga.w_bytes = (unsigned)w_bytes;
      ga.out_bytes = (unsigned)out_bytes;
      ga.res_bytes = (unsigned)res_bytes;
      const int64_t ntiles = (int64_t)ga.g.tiles_m * ga.g.tiles_n;
      if (ntiles > 0x7fffffffLL) TFIMM_FAIL(TFIMM_EINVAL, "gemm: grid too large");
      ga.n_tiles = (int)ntiles;
      {
        // column-panel groups (GemmStreamArgs::ngroup).  TFIMM_GEMM_NGROUP: -1 (default) = the rule below, 0 = never, n = groups of n
        static const int ng_env = getenv("TFIMM_GEMM_NGROUP") ? atoi(getenv("TFIMM_GEMM_NGROUP")) : -1;
        int ng = 0;
        if (ng_env > 0) ng = ng_env;
        else if (ng_env < 0 && kmode == K_DENSE && !scale && w_bytes > (int64_t)(3 << 20) && ga.g.tiles_m >= 64) {
          // as many weight panels as stay in one XCD's L2 next to the streaming rows: 5/8 of it (4 MiB on MI355X -> 2.5 MB;
          // ViT-B, K = 768, 256-column panels of 393 KB: groups of 6 -- measured -0.9 .. -1.1 % of a ViT-B step on three boxes;
          // 5, 7, 8 and the half split of 9 panels gain 0.2 .. 0.6 %, groups of 2 LOSE 2 %: A is re-read once per group).
          // Dense rows only (plain and LayerNorm-folded -- both ViT-B flavours were in that measurement): an implicit-GEMM
          // convolution re-gathers A per group and the SE-gate flavour re-scales it; their table entries were timed in
          // M-panel-major order.
          const int64_t panel = (int64_t)t->bn * d.ldw * 2;
          ng = (int)(l2_budget() / (panel > 0 ? panel : 1));
          if (ng < 3) ng = 0;
        }
        ga.ngroup = (ng > 0 && ng < ga.g.tiles_n) ? ng : 0;
      }
      ga.cin64 = (kmode == K_CONV && (d.Cin % (ti == 9 ? 32 : 64)) == 0) ? 1 : 0;   // whole k-tiles inside one filter tap
      ga.duo_delay = 0;
      ga.duo_first = num_cu() / 8;
      if (ti == 9) {
        // phase shift of the second workgroup of a CU: about half a tile (a k-tile is 16 MFMAs = 512 cycles of one wave)
        static const int c1 = getenv("TFIMM_DUO_DELAY_K") ? atoi(getenv("TFIMM_DUO_DELAY_K")) : 256;
        static const int c0 = getenv("TFIMM_DUO_DELAY_0") ? atoi(getenv("TFIMM_DUO_DELAY_0")) : 2000;
        const int64_t nk32 = cdiv64(d.K, 32);
        ga.duo_delay = (int)std::min<int64_t>(nk32 * c1 + c0, 200000);
        if (c1 == 0 && c0 == 0) ga.duo_delay = 0;
      }
      ga.cin_magic = ga.kw_magic = 0;
      {
        static const int dbg = getenv("TFIMM_GEMM_DBG") ? atoi(getenv("TFIMM_GEMM_DBG")) : 0;
        ga.dbg = dbg;
        static const char* dp = getenv("TFIMM_GEMM_DBG_PTR");
        ga.dbg_ptr = dp ? (long long*)strtoull(dp, nullptr, 0) : nullptr;
      }
      if (kmode == K_CONV && d.K < 65536) {
        if (d.Cin > 1) ga.cin_magic = (unsigned)(0x100000000ULL / (unsigned)d.Cin) + 1u;
        if (d.KW > 1) ga.kw_magic = (unsigned)(0x100000000ULL / (unsigned)d.KW) + 1u;
      }
      int64_t grid = (int64_t)num_cu() * occ_f[ti];
      grid = (grid + 7) / 8 * 8;
      const int64_t need = (ntiles + 7) / 8 * 8;
      if (grid > need) grid = need;
      ga.s_bytes = 0; ga.s_slots = ga.s_gp = 0;
      ga.a2 = (const bf16_t*)d.a2; ga.a2_bytes = (unsigned)a2_bytes;
      ga.a2_window = a2win;
      ga.K2 = d.K2; ga.lda2 = d.lda2; ga.a2_stride = d.a2_stride; ga.a2_H = d.a2_H; ga.a2_W = d.a2_W; ga.a2_OH = d.a2_OH; ga.a2_OW = d.a2_OW;
      ga.ln_stats = d.ln_stats; ga.ln_c1 = d.ln_c1;
      ga.ln_stats_bytes = ln_in ? (unsigned)((int64_t)d.M * 8) : 0u;
      ga.ln_c1_bytes = ln_in ? (unsigned)((int64_t)d.N * 32) : 0u;
      if (ln_in) {
        const size_t lds = (size_t)t->lds_bytes + t->ln_lds;
        static int ln_occ[TFIMM_GEMM_STREAM_NUM_TILES] = {};
        static tfimm_once_t ln_ready[TFIMM_GEMM_STREAM_NUM_TILES];
        if (ln_ready[ti].need()) {
          TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)t->fn_ln, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          int nb = 0;
          TFIMM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)t->fn_ln, t->threads, lds));
          ln_occ[ti] = nb < 1 ? 1 : (nb > 4 ? 4 : nb);
          ln_ready[ti].mark();
        }
        grid = ((int64_t)num_cu() * ln_occ[ti] + 7) / 8 * 8;
        if (grid > need) grid = need;
        TFIMM_LAUNCH(t->fn_ln, dim3((unsigned)grid), dim3(t->threads), lds, (hipStream_t)stream, ga);
        return 0;
      }
      if (dual) {
        static tfimm_once_t dual_attr[TFIMM_GEMM_STREAM_NUM_TILES][2];
        if (dual_attr[ti][fi].need()) {
          TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)t->fn_dual[fi], hipFuncAttributeMaxDynamicSharedMemorySize, t->lds_bytes));
          dual_attr[ti][fi].mark();
        }
        TFIMM_LAUNCH(t->fn_dual[fi], dim3((unsigned)grid), dim3(t->threads), (size_t)t->lds_bytes, (hipStream_t)stream, ga);
        return 0;
      }
      if (!scale) {
        TFIMM_LAUNCH(t->fn[fi][ei], dim3((unsigned)grid), dim3(t->threads), (size_t)t->lds_bytes, (hipStream_t)stream, ga);
        return 0;
      }
      // SE gate on A: per k-tile, 64 gate values of every image a row tile touches ride in LDS next to the operand stage
      // (1 KiB per four image slots and stage; wave w brings piece w)
      const int nw = t->threads / 64;
      const int64_t nimg = cdiv64(d.M, d.rows_per_image);
      ga.s_bytes = (unsigned)(nimg * d.K * 4);
      ga.s_slots = (int)cdiv64(t->bm, d.rows_per_image) + 1;
      ga.s_gp = (ga.s_slots + 3) / 4;
      const size_t lds = (size_t)t->lds_bytes + (size_t)2 * ga.s_gp * 1024;
      if (ga.s_gp <= nw && lds <= 160 * 1024 && nimg * d.K * 4 <= 0x7fffff00LL) {
        static tfimm_once_t scale_attr[TFIMM_GEMM_STREAM_NUM_TILES][3];
        if (scale_attr[ti][ei].need()) {
          TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)t->fn_scale[ei], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
          scale_attr[ti][ei].mark();
        }
        // resident workgroups per CU with the gate slices counted in
        int occ_s = (int)((160 * 1024) / lds);
        occ_s = occ_s < 1 ? 1 : (occ_s > occ_f[ti] ? occ_f[ti] : occ_s);
        grid = ((int64_t)num_cu() * occ_s + 7) / 8 * 8;
        if (grid > need) grid = need;
        TFIMM_LAUNCH(t->fn_scale[ei], dim3((unsigned)grid), dim3(t->threads), lds, (hipStream_t)stream, ga);
        return 0;
      }
      // more image slots per tile than waves to bring them (images of a few rows): the register-staged kernel below
    }
  }

  if (d.ln_stats) TFIMM_FAIL(TFIMM_EUNSUP, "gemm: LayerNorm folding needs the persistent LDS-DMA family (K-padded weights, 16-byte aligned rows)");
  if (d.a2) TFIMM_FAIL(TFIMM_EUNSUP, "gemm: a second A operand needs the persistent LDS-DMA family (channel counts %% 8 == 0, K-padded weights, 16-byte aligned rows)");
  if (d.mode != TFIMM_A_DENSE && g.stride_w != g.stride)
    TFIMM_FAIL(TFIMM_EUNSUP, "gemm: stride_w != stride needs the persistent LDS-DMA family (Cin %% 8 == 0, 16-byte aligned input)");

  // ---- LDS-DMA family: aligned dense rows or Cin % 8 == 0 gathers, weights padded to 64 in k,
  //      tensors addressable with a 31-bit byte offset
  {
    const int64_t a_bytes = (d.mode == TFIMM_A_DENSE) ? ((int64_t)(d.M - 1) * d.lda + d.K) * 2
                                                       : (((int64_t)d.B * d.H * d.W - 1) * g.cpitch + d.Cin) * 2;
    const int64_t w_bytes = (int64_t)d.N * d.ldw * 2;
    const bool hint_v1 = d.tile_hint > 0 && d.tile_hint <= TFIMM_GEMM_NUM_TILES;  // 11..16 = this family
    const bool ok = (kmode == K_DENSE || kmode == K_CONV) && !hint_v1 && !dma_disabled() &&
                    d.ldw >= (int)(cdiv64(d.K, 64) * 64) && a_bytes <= 0x7fffff00LL && w_bytes <= 0x7fffff00LL;
    if (ok) {
      const int ti = pick_dma_tile(d);
      const DmaTileCfg* t = dma_tile_table(ti);
      GemmDmaArgs ga;
      ga.g = g;
      ga.g.tiles_m = (int)cdiv64(d.M, t->bm);
      ga.g.tiles_n = (int)cdiv64(d.N, t->bn);
      ga.a_bytes = (unsigned)a_bytes;
      ga.w_bytes = (unsigned)w_bytes;
      const int64_t nblocks = (int64_t)ga.g.tiles_m * ga.g.tiles_n;
      if (nblocks > 0x7fffffffLL) TFIMM_FAIL(TFIMM_EINVAL, "gemm: grid too large");
      const size_t lds_bytes = (size_t)(t->bm + t->bn) * 128 * 2;
      gemm_dma_fn fn = t->fn[kmode == K_DENSE ? 0 : 1];
      static tfimm_once_t dma_attr_done[TFIMM_GEMM_DMA_NUM_TILES][2];
      if (dma_attr_done[ti][kmode == K_DENSE ? 0 : 1].need()) {
        TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        dma_attr_done[ti][kmode == K_DENSE ? 0 : 1].mark();
      }
      TFIMM_LAUNCH(fn, dim3((unsigned)nblocks), dim3(t->threads), lds_bytes, (hipStream_t)stream, ga);
      return 0;
    }
  }

  int ti = pick_tile(d, kmode);
  const TileCfg* t = tile_table(ti);
  if (!t->fn[kmode]) {  // flavour not built for the picked tile: fall back to 64x64 / 128x64
    ti = (kmode == K_DENSE_SCALE) ? 1 : 2;
    t = tile_table(ti);
    if (!t->fn[kmode]) TFIMM_FAIL(TFIMM_EUNSUP, "gemm: no kernel for flavour %d", kmode);
  }
  g.tiles_m = (int)cdiv64(d.M, t->bm);
  g.tiles_n = (int)cdiv64(d.N, t->bn);
  const int64_t nblocks = (int64_t)g.tiles_m * g.tiles_n;
  if (nblocks > 0x7fffffffLL) TFIMM_FAIL(TFIMM_EINVAL, "gemm: grid too large");
  const size_t lds_bytes = (size_t)(t->bm + t->bn) * BK * 2 * 2;
  gemm_fn fn = t->fn[kmode];
  static tfimm_once_t attr_done[TFIMM_GEMM_NUM_TILES][K_NUM];
  if (attr_done[ti][kmode].need()) {
    TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done[ti][kmode].mark();
  }
  TFIMM_LAUNCH(fn, dim3((unsigned)nblocks), dim3(t->threads), lds_bytes, (hipStream_t)stream, g);
  return 0;
}
