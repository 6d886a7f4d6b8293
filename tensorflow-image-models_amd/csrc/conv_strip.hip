// 3x3 / stride 1 / pad 1 convolution of a 128-channel NHWC tensor to 128 channels (gfx950): the middle convolution of a
// ResNet-50 stage-2 bottleneck (resnet.py:273-283: pad2 / conv2 / bn2 / act2, BatchNorm folded on the host), reached through
// tfimm_hip_gemm (csrc/gemm.hip dispatches here when the descriptor has exactly this shape).
//
// As an implicit GEMM on the persistent LDS-DMA tile (256 x 128 x 64) this layer ran at 460 TFLOP/s and 0.8 TB/s -- bound by
// neither roof but by LDS: every input pixel is gathered from L2 into LDS NINE times (once per filter tap), 48 KiB of DMA
// per 1024 cycles of MFMA work on top of 1 KiB of fragment reads per MFMA.  Here, as in the fused bottleneck tail of stage 1
// (gemm_chain_kernel.h):
//   * INPUT STRIP: at stride 1 the pixels a tile of 128 consecutive output pixels needs are ONE contiguous run of the
//     flattened NHWC tensor, [m0 - W - 1, m0 + 128 + W + 1): 186 rows of 256 bytes at W = 28.  It goes into LDS once per
//     tile (16-byte chunks XOR-swizzled by row) and the B fragment (pixels) of tap (ky, kx) is read from row
//     local + ky W + kx; taps outside the image are zeroed in the register.  Per tile the LDS-DMA moves 48 KiB of strip
//     + 288 KiB of weights instead of 288 + 288;
//   * four waves as 2 (pixel halves) x 2 (channel halves), 64 x 64 outputs each, operands swapped (D[n][m] = W . X^T) so
//     that a lane's accumulators are channel quads of ONE pixel; 80 KiB of LDS (strip 48 + two 16-KiB weight stages), TWO
//     workgroups per CU: one's strip load / epilogue runs under the other's MFMAs;
//   * weights as 18 steps of [128 out][64 in] (tap, channel half) = exactly the k-tiles of the [N][K] matrix the GEMM
//     path multiplies with (pack.pack_conv: K order (ky, kx, ci)), one step ahead in a two-stage ring, one barrier per step;
//   * epilogue: bias + activation on the accumulators, bf16 block of the wave through LDS (aliasing the strip, which is
//     dead by then), row-contiguous 16-byte stores.
#include "gemm_stream_kernel.h"

#include <cstdlib>

namespace tfimm_gemm {

struct StripArgs {
  GemmArgs g;
  unsigned a_bytes, w_bytes, out_bytes;
  int n_tiles;
  int hw;            // H * W
};

constexpr int STRIP_C = 128;                 // input channels = output channels
constexpr int STRIP_BM = 128;                // pixels per tile
constexpr int STRIP_ROWS = 192;              // strip rows held (>= BM + 2 W + 2  ->  W <= 31)
constexpr int STRIP_BYTES = STRIP_ROWS * 256;
constexpr int STRIP_STAGE = 128 * 128;       // [128 out][64 in] bf16
constexpr int STRIP_LDS = STRIP_BYTES + 2 * STRIP_STAGE;   // 80 KiB

__global__ void __launch_bounds__(256, 2) conv_strip_kernel(const StripArgs pa) {
  const GemmArgs& p = pa.g;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sStrip = smem;
  char* const sRing = smem + STRIP_BYTES;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave >> 1, wc = wave & 1;       // pixel half, channel half
  const int frow = lane & 31, fhi = lane >> 5;

  int t_first, t_hi, t_step;
  {
    const int nb = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int q = pa.n_tiles >> 3, r = pa.n_tiles & 7;
    const int t_lo = xcd * q + (xcd < r ? xcd : r);
    t_hi = t_lo + q + (xcd < r ? 1 : 0);
    t_step = nb >> 3;
    t_first = t_lo + j;
  }
  if (t_first >= t_hi) return;

  const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(p.a, pa.a_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(p.wt, pa.w_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_o = make_rsrc(p.out, pa.out_bytes);
  const ActParams actp = make_act(p.act);
  const int W = p.W, H = p.H;

  // weight DMA: 16 pieces of 1 KiB per step (8 rows of 128 bytes), four per wave; lane -> (row, physical chunk)
  unsigned w_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wave * 4 + j) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((r >> 1) & 7);
    w_off[j] = (unsigned)(((size_t)r * p.ldw + chunk * 8) * 2);
  }
  auto issue_w = [&](int step, int stage) __attribute__((always_inline)) {
    char* sb = sRing + stage * STRIP_STAGE;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(sb + (wave * 4 + j) * 1024), 16, (int)w_off[j], step * 128, 0, 0);
  };
  // strip DMA: 48 pieces of 1 KiB (4 rows of 256 bytes), twelve per wave; physical chunk pc of row r holds logical chunk pc ^ (r & 15)
  auto issue_strip = [&](int m0) __attribute__((always_inline)) {
    const long long g0 = (long long)m0 - W - 1;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int r = (wave * 12 + j) * 4 + (lane >> 4);
      const int chunk = (lane & 15) ^ (r & 15);
      const long long g = g0 + r;
      const unsigned off = (g >= 0 && g < (long long)p.M) ? (unsigned)(((size_t)g * STRIP_C + chunk * 8) * 2) : kOobOffset;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(sStrip + (wave * 12 + j) * 1024), 16, (int)off, 0, 0, 0);
    }
  };

  for (int tile = t_first; tile < t_hi; tile += t_step) {
    const int m0 = tile * STRIP_BM;
    // every wave is done with the previous tile's epilogue block (it aliases the strip) and with both weight stages
    tfimm_lds_reuse_barrier();
    issue_strip(m0);
    issue_w(0, 0);

    // this lane's two pixels (groups i = 0, 1 of the wave's 64): tap validity, strip row of tap (0, 0)
    unsigned vmask[2];
    int srow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int local = wp * 64 + i * 32 + frow;
      const int m = m0 + local;
      const int rem = m % pa.hw;
      const int y = rem / W, x = rem - y * W;
      unsigned mk = 0;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const bool ok = m < p.M && (unsigned)(y + ky - 1) < (unsigned)H && (unsigned)(x + kx - 1) < (unsigned)W;
          mk |= ok ? (1u << (ky * 3 + kx)) : 0u;
        }
      vmask[i] = mk;
      srow[i] = local;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Fragment reads are explicit ds_read_b128 with counted waits: in front of an LDS read it can see hipcc drains EVERY
    // outstanding LDS-DMA (s_waitcnt vmcnt(0)), i.e. the next step's weights.  Two register sets: the reads of k-slice
    // ks + 1 are in flight while the four MFMAs of slice ks issue.
    const unsigned strip_base = (unsigned)(size_t)(lds_ptr_t)sStrip;
    const unsigned ring_base = (unsigned)(size_t)(lds_ptr_t)sRing;
    unsigned wrow_off[2], wsw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wc * 64 + j * 32 + frow;
      wrow_off[j] = (unsigned)(row * 128);
      wsw[j] = (unsigned)((row >> 1) & 7);
    }
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const int roff = ky * W + kx;
      unsigned xrow_off[2], xsw[2], keep[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = srow[i] + roff;
        xrow_off[i] = strip_base + (unsigned)(r * 256);
        xsw[i] = (unsigned)(r & 15);
        keep[i] = ((vmask[i] >> tap) & 1u) ? 0xffffffffu : 0u;
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int step = tap * 2 + half;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this step's weights (and, first step, the strip) have landed
        tfimm_lds_reuse_barrier();                               // ... everyone's; the other stage is free again
        if (step + 1 < 18) issue_w(step + 1, (step + 1) & 1);
        const unsigned wst = ring_base + (unsigned)((step & 1) * STRIP_STAGE);
        u32x4 fx[2][2], fw[2][2];
        auto reads = [&](int ks, int buf) __attribute__((always_inline)) {
          const unsigned cx = (unsigned)(half * 8 + ks * 2 + fhi), cw = (unsigned)(ks * 2 + fhi);
          const unsigned ax0 = xrow_off[0] + ((cx ^ xsw[0]) << 4), ax1 = xrow_off[1] + ((cx ^ xsw[1]) << 4);
          const unsigned aw0 = wst + wrow_off[0] + ((cw ^ wsw[0]) << 4), aw1 = wst + wrow_off[1] + ((cw ^ wsw[1]) << 4);
          asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7"
                       : "=&v"(fx[buf][0]), "=&v"(fx[buf][1]), "=&v"(fw[buf][0]), "=&v"(fw[buf][1])
                       : "v"(ax0), "v"(ax1), "v"(aw0), "v"(aw1) : "memory");
        };
        reads(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int b = ks & 1;
          if (ks + 1 < 4) {
            reads(ks + 1, b ^ 1);
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fx[b][0]), "+v"(fx[b][1]), "+v"(fw[b][0]), "+v"(fw[b][1]));
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fx[b][0]), "+v"(fx[b][1]), "+v"(fw[b][0]), "+v"(fw[b][1]));
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) fx[b][i] &= keep[i];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[b][j]), __builtin_bit_cast(bf16x8, fx[b][i]),
                                                                  acc[i][j], 0, 0, 0);
        }
      }
    }

    // ---- epilogue: the strip is dead once every wave has left the last step
    tfimm_lds_reuse_barrier();
    char* const sE = sStrip + wave * 8192;            // this wave's 64 pixels x 64 channels, bf16: 128 bytes per pixel row
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // lane: pixel i*32 + frow, channels j*32 + q*8 + fhi*4 .. +3 (q = 0..3)
        const int n_base = wc * 64 + j * 32 + fhi * 4;
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          tfimm_f32x2 v[4];
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int q = q2 * 2 + h2;
            const float4 b4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + n_base + q * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[h2 * 2 + 0] = tfimm_f32x2{acc[i][j][q * 4 + 0] + b4.x, acc[i][j][q * 4 + 1] + b4.y};
            v[h2 * 2 + 1] = tfimm_f32x2{acc[i][j][q * 4 + 2] + b4.z, acc[i][j][q * 4 + 3] + b4.w};
          }
          act8p(v, actp);
          const uint4 pk = pack8p(v);
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int q = q2 * 2 + h2;
            const int row = i * 32 + frow;
            const int chunk = j * 4 + q;                       // 16-byte chunk of the wave's 128-byte row
            *reinterpret_cast<uint2*>(sE + row * 128 + ((chunk ^ (row & 7)) * 16) + fhi * 8) =
                h2 ? make_uint2(pk.z, pk.w) : make_uint2(pk.x, pk.y);
          }
        }
      }
    // read back row-contiguous (the wave's own block: DS operations of one wave execute in order) and store 16 bytes per lane
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
      const int row = ps * 8 + (lane >> 3), c = lane & 7;
      const uint4 u = *reinterpret_cast<const uint4*>(sE + row * 128 + ((c ^ (row & 7)) * 16));
      const int m = m0 + wp * 64 + row;
      const unsigned off = m < p.M ? (unsigned)(((size_t)m * p.ldc + wc * 64 + c * 8) * 2) : kOobOffset;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, u), rsrc_o, (int)off, 0, 0);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The same convolution CHAINED with the 1x1 convolution behind it: the tail of a ResNet stage-2 bottleneck in one launch,
//     mid = act1( conv3x3(x) + b1 )                        128 -> 128 channels, input strip as above
//     out = act2( mid . W2^T + b2 + residual )             K2 = 128, N2 = 64 * steps (512)
// (resnet.py:273-290; tfimm_hip_conv_chain with C1 = 128 -- the stage-1 case, C1 = 64, is gemm_chain_kernel.h).  The 128-channel
// intermediate never leaves the registers: a wave owns 32 pixels and ALL 128 intermediate channels of them; with the operands
// swapped its GEMM-1 accumulators are, per lane, one pixel and channel quads 32 j + 8 q + 4 (lane >> 5) -- packed to bf16 they
// are the B operand of GEMM 2 up to the order of K, and the host stores W2 with its K axis in that order
// (pack.chain_k_order).  One flattened step stream per tile through the same two-stage ring: 18 W1 steps [128 out][64 in],
// then N2 / 64 W2 slices [64 out][128 in] (16 KiB each).  GEMM-2 epilogue per slice: fp32 block of the wave through LDS (the
// strip is dead by then), row-contiguous residual loads and 16-byte stores; residual rows and bias of a slice are requested
// before its MFMAs.  VMEM operations retire in issue order: the wait in front of a GEMM-2 step leaves exactly the previous
// slice's four stores in flight.
// ---------------------------------------------------------------------------------------------------------------------------
struct StripChainArgs {
  const bf16_t* x; const bf16_t* w1; const float* b1; const bf16_t* w2; const float* b2; const bf16_t* residual; bf16_t* out;
  int M, N2, H, W, hw;
  int ldw1, ldw2, ldr, ldc;
  int act1, act2;
  unsigned x_bytes, w1_bytes, w2_bytes, out_bytes, res_bytes;
  int n_tiles;
};

// ACT >= 0: both activations are that TFIMM_ACT_* with its parameters folded into the instructions (ResNet: relu) -- run-time
// activation parameters are 16 scalar registers this kernel does not have; ACT < 0: p.act1 / p.act2
template <int NS2, int ACT>
__global__ void __launch_bounds__(256, 2) conv_strip_chain_kernel(const StripChainArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sStrip = smem;
  char* const sRing = smem + STRIP_BYTES;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  constexpr int NK1 = 18;

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
  const __amdgpu_buffer_rsrc_t rsrc_r = make_rsrc(p.residual, p.res_bytes);     // no residual: zero records, every load returns 0
  const ActParams act1 = make_act(ACT >= 0 ? ACT : p.act1), act2 = make_act(ACT >= 0 ? ACT : p.act2);
  const int W = p.W, H = p.H;
  // ring DMA.  W1 step: 16 pieces of 8 rows x 128 bytes; W2 slice: 16 pieces of 4 rows x 256 bytes (chunk swizzled by row & 15)
  unsigned w1_off[4], w2_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r1 = (wave * 4 + j) * 8 + (lane >> 3);
    w1_off[j] = (unsigned)(((size_t)r1 * p.ldw1 + ((lane & 7) ^ ((r1 >> 1) & 7)) * 8) * 2);
    const int r2 = (wave * 4 + j) * 4 + (lane >> 4);
    w2_off[j] = (unsigned)(((size_t)r2 * p.ldw2 + ((lane & 15) ^ (r2 & 15)) * 8) * 2);
  }
  auto issue_step = [&](int step) __attribute__((always_inline)) {       // step < 18: W1 k-tile; else W2 slice step - 18
    char* sb = sRing + (step & 1) * STRIP_STAGE;
    if (step < NK1) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w1, (lds_ptr_t)(sb + (wave * 4 + j) * 1024), 16, (int)w1_off[j], step * 128, 0, 0);
    } else {
      const int soff = (step - NK1) * 64 * p.ldw2 * 2;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w2, (lds_ptr_t)(sb + (wave * 4 + j) * 1024), 16, (int)w2_off[j], soff, 0, 0);
    }
  };
  auto issue_strip = [&](int m0) __attribute__((always_inline)) {
    const long long g0 = (long long)m0 - W - 1;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int r = (wave * 12 + j) * 4 + (lane >> 4);
      const int chunk = (lane & 15) ^ (r & 15);
      const long long g = g0 + r;
      const unsigned off = (g >= 0 && g < (long long)p.M) ? (unsigned)(((size_t)g * STRIP_C + chunk * 8) * 2) : kOobOffset;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(sStrip + (wave * 12 + j) * 1024), 16, (int)off, 0, 0, 0);
    }
  };

  const unsigned strip_base = (unsigned)(size_t)(lds_ptr_t)sStrip;
  const unsigned ring_base = (unsigned)(size_t)(lds_ptr_t)sRing;
  // epilogue block of this wave: 32 pixels x 64 channels fp32 (256 bytes per pixel row, 16 chunks swizzled by row & 15), in the dead strip
  char* const sE = sStrip + wave * 8192;
  const int e_row = lane >> 3, e_c = lane & 7;       // read-back: 8 lanes per row (8 channels = 32 bytes each), 8 rows per pass

  for (int tile = t_first; tile < t_hi; tile += t_step) {
    const int m0 = tile * STRIP_BM;
    // (the previous tile's last stores may still be in flight: their data left LDS for registers before they were issued)
    tfimm_lds_reuse_barrier();
    issue_strip(m0);
    issue_step(0);

    const int local = wave * 32 + frow;
    const int m = m0 + local;
    unsigned vmask = 0;
    {
      const int rem = m % p.hw;
      const int y = rem / W, x = rem - y * W;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const bool ok = m < p.M && (unsigned)(y + ky - 1) < (unsigned)H && (unsigned)(x + kx - 1) < (unsigned)W;
          vmask |= ok ? (1u << (ky * 3 + kx)) : 0u;
        }
    }

    // ---------------- GEMM 1: 32 pixels x 128 channels per wave
    f32x16 acc1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc1[j][e] = 0.f;
    unsigned wrow1[4], wsw1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = j * 32 + frow;
      wrow1[j] = (unsigned)(row * 128);
      wsw1[j] = (unsigned)((row >> 1) & 7);
    }
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const int r = local + ky * W + kx;
      const unsigned xrow = strip_base + (unsigned)(r * 256), xsw = (unsigned)(r & 15);
      const unsigned keep = ((vmask >> tap) & 1u) ? 0xffffffffu : 0u;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int step = tap * 2 + half;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tfimm_lds_reuse_barrier();
        issue_step(step + 1);                                   // (step 17 requests the first W2 slice)
        const unsigned wst = ring_base + (unsigned)((step & 1) * STRIP_STAGE);
        u32x4 fx[2], fw[2][4];
        auto reads = [&](int ks, int buf) __attribute__((always_inline)) {
          const unsigned cx = (unsigned)(half * 8 + ks * 2 + fhi), cw = (unsigned)(ks * 2 + fhi);
          const unsigned ax = xrow + ((cx ^ xsw) << 4);
          const unsigned a0 = wst + wrow1[0] + ((cw ^ wsw1[0]) << 4), a1 = wst + wrow1[1] + ((cw ^ wsw1[1]) << 4);
          const unsigned a2 = wst + wrow1[2] + ((cw ^ wsw1[2]) << 4), a3 = wst + wrow1[3] + ((cw ^ wsw1[3]) << 4);
          asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %6\n\tds_read_b128 %2, %7\n\tds_read_b128 %3, %8\n\tds_read_b128 %4, %9"
                       : "=&v"(fx[buf]), "=&v"(fw[buf][0]), "=&v"(fw[buf][1]), "=&v"(fw[buf][2]), "=&v"(fw[buf][3])
                       : "v"(ax), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
        };
        reads(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int b = ks & 1;
          if (ks + 1 < 4) {
            reads(ks + 1, b ^ 1);
            asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fx[b]), "+v"(fw[b][0]), "+v"(fw[b][1]), "+v"(fw[b][2]), "+v"(fw[b][3]));
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fx[b]), "+v"(fw[b][0]), "+v"(fw[b][1]), "+v"(fw[b][2]), "+v"(fw[b][3]));
          }
          fx[b] &= keep;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[b][j]), __builtin_bit_cast(bf16x8, fx[b]), acc1[j], 0, 0, 0);
        }
      }
    }
    // intermediate: + b1, act1, ONE rounding to bf16 -- as the two-launch path stores it.  P[t]: the B operand of GEMM-2 k-step t
    u32x4 P[8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        tfimm_f32x2 v[4];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int q = s2 * 2 + h2;
          const float4 b4 = *reinterpret_cast<const float4*>(p.b1 + j * 32 + q * 8 + fhi * 4);
          v[h2 * 2 + 0] = tfimm_f32x2{acc1[j][q * 4 + 0] + b4.x, acc1[j][q * 4 + 1] + b4.y};
          v[h2 * 2 + 1] = tfimm_f32x2{acc1[j][q * 4 + 2] + b4.z, acc1[j][q * 4 + 3] + b4.w};
        }
        act8p(v, act1);
        P[j * 2 + s2] = __builtin_bit_cast(u32x4, pack8p(v));
      }

    // ---------------- GEMM 2: 32 pixels x 64 channels per slice
    unsigned wrow2[2], wsw2[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int row = b * 32 + frow;
      wrow2[b] = (unsigned)(row * 256);
      wsw2[b] = (unsigned)(row & 15);
    }
#pragma unroll 1
    for (int u = 0; u < NS2; ++u) {
      const int step = NK1 + u;
      if (u == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        // the previous slice's stores may stay in flight
      tfimm_lds_reuse_barrier();                                   // (u = 0: every wave is done with the strip as well)
      if (u + 1 < NS2) issue_step(step + 1);
      // residual rows and bias of this slice: requested ahead of the MFMAs
      const int n0 = u * 64;
      uint4 rres[4];
      unsigned ooff[4];
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int mr = m0 + wave * 32 + ps * 8 + e_row;
        const bool ok = mr < p.M;
        ooff[ps] = ok ? (unsigned)(((size_t)mr * p.ldc + n0 + e_c * 8) * 2) : kOobOffset;
        const unsigned roff = ok ? (unsigned)(((size_t)mr * p.ldr + n0 + e_c * 8) * 2) : kOobOffset;
        rres[ps] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_r, (int)roff, 0, 0));
      }
      const float4 bb0 = *reinterpret_cast<const float4*>(p.b2 + n0 + e_c * 8);
      const float4 bb1 = *reinterpret_cast<const float4*>(p.b2 + n0 + e_c * 8 + 4);
      const unsigned wst = ring_base + (unsigned)((step & 1) * STRIP_STAGE);
      f32x16 acc2[2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[b][e] = 0.f;
      u32x4 fa[2][2];
      auto reads2 = [&](int t, int buf) __attribute__((always_inline)) {
        const unsigned c = (unsigned)(t * 2 + fhi);
        const unsigned a0 = wst + wrow2[0] + ((c ^ wsw2[0]) << 4), a1 = wst + wrow2[1] + ((c ^ wsw2[1]) << 4);
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(fa[buf][0]), "=&v"(fa[buf][1]) : "v"(a0), "v"(a1) : "memory");
      };
      reads2(0, 0);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int b = t & 1;
        if (t + 1 < 8) {
          reads2(t + 1, b ^ 1);
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[b][0]), "+v"(fa[b][1]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[b][0]), "+v"(fa[b][1]));
        }
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
          acc2[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[b][bb]), __builtin_bit_cast(bf16x8, P[t]), acc2[bb], 0, 0, 0);
      }
      // stage the fp32 block (lane: pixel frow, channel quads bb*32 + q*8 + fhi*4), read it back row-contiguous
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int chunk = bb * 8 + q * 2 + fhi;                  // 16-byte chunk of the 256-byte row
          const f32x4 v = {acc2[bb][q * 4 + 0], acc2[bb][q * 4 + 1], acc2[bb][q * 4 + 2], acc2[bb][q * 4 + 3]};
          *reinterpret_cast<f32x4*>(sE + frow * 256 + ((chunk ^ (frow & 15)) * 16)) = v;
        }
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int row = ps * 8 + e_row;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(sE + row * 256 + (((2 * e_c) ^ (row & 15)) * 16));
        const f32x4 hi = *reinterpret_cast<const f32x4*>(sE + row * 256 + (((2 * e_c + 1) ^ (row & 15)) * 16));
        tfimm_f32x2 v[4] = {{lo[0] + bb0.x, lo[1] + bb0.y}, {lo[2] + bb0.z, lo[3] + bb0.w}, {hi[0] + bb1.x, hi[1] + bb1.y}, {hi[2] + bb1.z, hi[3] + bb1.w}};
        tfimm_f32x2 r2[4];
        unpack8p(rres[ps], r2);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r2[e];
        act8p(v, act2);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pack8p(v)), rsrc_o, (int)ooff[ps], 0, 0);
      }
    }
  }
}

}  // namespace tfimm_gemm

// host side: called by tfimm_hip_gemm for descriptors of exactly this shape (see gemm.hip: conv_strip_applies)
int tfimm_launch_conv_strip(const tfimm_gemm::GemmArgs& g, int64_t a_bytes, int64_t w_bytes, int64_t out_bytes, int num_cu, hipStream_t stream) {
  using namespace tfimm_gemm;
  static tfimm_once_t attr_done;        // per device, shared by host threads (common.h)
  if (attr_done.need()) {
    TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)conv_strip_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, STRIP_LDS));
    attr_done.mark();
  }
  StripArgs a;
  a.g = g;
  a.a_bytes = (unsigned)a_bytes; a.w_bytes = (unsigned)w_bytes; a.out_bytes = (unsigned)out_bytes;
  a.n_tiles = (int)((g.M + STRIP_BM - 1) / STRIP_BM);
  a.hw = g.H * g.W;
  int64_t grid = ((int64_t)num_cu * 2 + 7) / 8 * 8;
  const int64_t need = ((int64_t)a.n_tiles + 7) / 8 * 8;
  if (grid > need) grid = need;
  TFIMM_LAUNCH(conv_strip_kernel, dim3((unsigned)grid), dim3(256), (size_t)STRIP_LDS, stream, a);
  return 0;
}

// tfimm_hip_conv_chain with C1 = 128 (csrc/gemm_chain.hip dispatches here): 3x3 / stride 1 / pad 1 over 128 -> 128 channels
// followed by a 1x1 convolution to N2 = 256 / 512 channels, rows of at most 31 pixels
int tfimm_launch_conv_strip_chain(const tfimm_chain_desc& d, int64_t M, int num_cu, hipStream_t stream) {
  using namespace tfimm_gemm;
  StripChainArgs a;
  a.x = (const bf16_t*)d.x; a.w1 = (const bf16_t*)d.w1; a.b1 = d.b1; a.w2 = (const bf16_t*)d.w2; a.b2 = d.b2;
  a.residual = (const bf16_t*)d.residual; a.out = (bf16_t*)d.out;
  a.M = (int)M; a.N2 = d.N2; a.H = d.H; a.W = d.W; a.hw = d.H * d.W;
  a.ldw1 = d.ldw1; a.ldw2 = d.ldw2; a.ldr = d.ldr; a.ldc = d.ldc;
  a.act1 = d.act1; a.act2 = d.act2;
  a.x_bytes = (unsigned)(M * 256);
  a.w1_bytes = (unsigned)((int64_t)128 * d.ldw1 * 2);
  a.w2_bytes = (unsigned)((int64_t)d.N2 * d.ldw2 * 2);
  a.out_bytes = (unsigned)(((M - 1) * d.ldc + d.N2) * 2);
  a.res_bytes = d.residual ? (unsigned)(((M - 1) * d.ldr + d.N2) * 2) : 0u;
  a.n_tiles = (int)((M + STRIP_BM - 1) / STRIP_BM);
  const bool relu = d.act1 == TFIMM_ACT_RELU && d.act2 == TFIMM_ACT_RELU;
  void (*fn)(const StripChainArgs) = relu ? (d.N2 == 512 ? conv_strip_chain_kernel<8, TFIMM_ACT_RELU> : conv_strip_chain_kernel<4, TFIMM_ACT_RELU>)
                                          : (d.N2 == 512 ? conv_strip_chain_kernel<8, -1> : conv_strip_chain_kernel<4, -1>);
  static tfimm_once_t ready[2][2];
  if (ready[relu][d.N2 == 512].need()) {
    TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, STRIP_LDS));
    ready[relu][d.N2 == 512].mark();
  }
  int64_t grid = ((int64_t)num_cu * 2 + 7) / 8 * 8;
  const int64_t need = ((int64_t)a.n_tiles + 7) / 8 * 8;
  if (grid > need) grid = need;
  TFIMM_LAUNCH(fn, dim3((unsigned)grid), dim3(256), (size_t)STRIP_LDS, stream, a);
  return 0;
}
