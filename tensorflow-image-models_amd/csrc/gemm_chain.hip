// tfimm_hip_conv_chain host side: descriptor validation and launch of the fused bottleneck-tail kernel
// (gemm_chain_kernel.h).
#include "gemm_chain_kernel.h"

#include <algorithm>
#include <cstdlib>

using namespace tfimm_gemm;

namespace {
int chain_num_cu() {
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

// csrc/conv_strip.hip
int tfimm_launch_conv_strip_chain(const tfimm_chain_desc& d, int64_t M, int num_cu, hipStream_t stream);

extern "C" int tfimm_hip_conv_chain(const tfimm_chain_desc* dp, void* stream) {
  if (!dp) TFIMM_FAIL(TFIMM_EINVAL, "conv_chain: null descriptor");
  const tfimm_chain_desc& d = *dp;
  if (!d.x || !d.w1 || !d.b1 || !d.w2 || !d.b2 || !d.out) TFIMM_FAIL(TFIMM_EINVAL, "conv_chain: null pointer");
  if (d.B <= 0 || d.H <= 0 || d.W <= 0 || d.Cin <= 0 || d.KH <= 0 || d.KW <= 0 || d.stride <= 0 || d.OH <= 0 || d.OW <= 0 ||
      d.C1 <= 0 || d.N2 <= 0)
    TFIMM_FAIL(TFIMM_EINVAL, "conv_chain: bad shape");
  // C1 = 128 (ResNet stage 2): the input-strip kernel with the 1x1 convolution chained behind it (csrc/conv_strip.hip)
  if (d.C1 == 128) {
    if (d.Cin != 128 || d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad_t != 1 || d.pad_l != 1 || d.OH != d.H || d.OW != d.W)
      TFIMM_FAIL(TFIMM_EUNSUP, "conv_chain: C1 = 128 is built for a 3x3 / stride 1 / pad 1 convolution over 128 -> 128 channels");
    if (d.W > 31) TFIMM_FAIL(TFIMM_EUNSUP, "conv_chain: C1 = 128 needs W <= 31 (W = %d): the strip of a 128-pixel tile must fit 192 LDS rows", d.W);
    if (d.N2 != 256 && d.N2 != 512) TFIMM_FAIL(TFIMM_EUNSUP, "conv_chain: N2 = %d (built: 256, 512)", d.N2);
    if (d.ds_x || d.ds_w) TFIMM_FAIL(TFIMM_EUNSUP, "conv_chain: the shortcut-convolution flavour exists for C1 = 64 only");
    if (d.ldw1 != 1152) TFIMM_FAIL(TFIMM_EINVAL, "conv_chain: ldw1 = %d, expected KH*KW*Cin = 1152", d.ldw1);
    if (d.ldw2 < 128 || (d.ldw2 & 7)) TFIMM_FAIL(TFIMM_EINVAL, "conv_chain: ldw2 = %d", d.ldw2);
    if (d.ldc < d.N2 || (d.ldc & 7) || (d.residual && (d.ldr < d.N2 || (d.ldr & 7))))
      TFIMM_FAIL(TFIMM_EINVAL, "conv_chain: output / residual rows must be 16-byte aligned (ldc = %d, ldr = %d)", d.ldc, d.ldr);
    if (((uintptr_t)d.x | (uintptr_t)d.w1 | (uintptr_t)d.w2 | (uintptr_t)d.out | (uintptr_t)d.residual | (uintptr_t)d.b1 | (uintptr_t)d.b2) & 15)
      TFIMM_FAIL(TFIMM_EINVAL, "conv_chain: pointers must be 16-byte aligned");
    const int64_t M = (int64_t)d.B * d.OH * d.OW;
    const int64_t big = std::max<int64_t>(M * 256, ((M - 1) * std::max(d.ldc, d.ldr) + d.N2) * 2);
    // (TFIMM_CHAIN_LIMIT lowers the threshold, as for C1 = 64 below: lets a test exercise the chunking on a small batch)
    static const int64_t lim128 = getenv("TFIMM_CHAIN_LIMIT") ? std::min<int64_t>(atoll(getenv("TFIMM_CHAIN_LIMIT")), 0x7fffff00LL) : 0x7fffff00LL;
    if (big > lim128) {             // image chunks, as below
      const int64_t per_img = (int64_t)d.OH * d.OW * std::max<int64_t>(256, (int64_t)std::max(d.ldc, d.ldr) * 2);
      const int64_t chunk = lim128 / per_img;
      if (chunk < 1 || d.B <= 1) TFIMM_FAIL(TFIMM_EUNSUP, "conv_chain: one image exceeds the 2 GiB a buffer descriptor addresses");
      for (int64_t b0 = 0; b0 < d.B; b0 += chunk) {
        tfimm_chain_desc c = d;
        c.B = (int32_t)std::min<int64_t>(chunk, d.B - b0);
        c.x = (const char*)d.x + b0 * d.H * d.W * 256;
        c.out = (char*)d.out + b0 * d.OH * d.OW * d.ldc * 2;
        if (d.residual) c.residual = (const char*)d.residual + b0 * d.OH * d.OW * d.ldr * 2;
        const int rc = tfimm_hip_conv_chain(&c, stream);
        if (rc != 0) return rc;
      }
      return 0;
    }
    return tfimm_launch_conv_strip_chain(d, M, chain_num_cu(), (hipStream_t)stream);
  }
  // built: 3x3 / stride 1 / pad 1 over 64 -> 64 channels (input strip: W <= 63), 256 or 512 output channels
  if (d.C1 != 64 || d.Cin != 64 || d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad_t != 1 || d.pad_l != 1 || d.OH != d.H ||
      d.OW != d.W)
    TFIMM_FAIL(TFIMM_EUNSUP, "conv_chain: built for a 3x3 / stride 1 / pad 1 convolution over 64 -> 64 channels");
  if (d.W > 63) TFIMM_FAIL(TFIMM_EUNSUP, "conv_chain: W = %d > 63 (the input strip of a 128-pixel tile must fit 256 LDS rows)", d.W);
  if (d.N2 != 256 && d.N2 != 512) TFIMM_FAIL(TFIMM_EUNSUP, "conv_chain: N2 = %d (built: 256, 512)", d.N2);
  const int64_t K1 = 576;
  if (d.ldw1 != K1) TFIMM_FAIL(TFIMM_EINVAL, "conv_chain: ldw1 = %d, expected KH*KW*Cin = %lld", d.ldw1, (long long)K1);
  if (d.ldw2 < d.C1 || (d.ldw2 & 7)) TFIMM_FAIL(TFIMM_EINVAL, "conv_chain: ldw2 = %d", d.ldw2);
  if (d.ldc < d.N2 || (d.ldc & 7) || (d.residual && (d.ldr < d.N2 || (d.ldr & 7))))
    TFIMM_FAIL(TFIMM_EINVAL, "conv_chain: output / residual rows must be 16-byte aligned (ldc = %d, ldr = %d)", d.ldc, d.ldr);
  if (((uintptr_t)d.x | (uintptr_t)d.w1 | (uintptr_t)d.w2 | (uintptr_t)d.out | (uintptr_t)d.residual | (uintptr_t)d.b1 |
       (uintptr_t)d.b2) & 15)
    TFIMM_FAIL(TFIMM_EINVAL, "conv_chain: pointers must be 16-byte aligned");
  const int64_t M = (int64_t)d.B * d.OH * d.OW;
  const int64_t x_bytes = (int64_t)d.B * d.H * d.W * d.Cin * 2;
  const int64_t w1_bytes = (int64_t)d.C1 * d.ldw1 * 2, w2_bytes = (int64_t)d.N2 * d.ldw2 * 2;
  const int64_t out_bytes = ((M - 1) * d.ldc + d.N2) * 2;
  const int64_t res_bytes = d.residual ? ((M - 1) * d.ldr + d.N2) * 2 : 0;
  const bool ds = d.ds_x != nullptr;
  if (ds != (d.ds_w != nullptr)) TFIMM_FAIL(TFIMM_EINVAL, "conv_chain: ds_x and ds_w go together");
  if (ds && (d.residual || d.ds_cin != 64 || (((uintptr_t)d.ds_x | (uintptr_t)d.ds_w) & 15)))
    TFIMM_FAIL(TFIMM_EUNSUP, "conv_chain: the shortcut convolution is built for a 64-channel block input and no other residual");
  if (ds && !(d.act1 == TFIMM_ACT_RELU && d.act2 == TFIMM_ACT_RELU))
    TFIMM_FAIL(TFIMM_EUNSUP, "conv_chain: the shortcut-convolution flavour is built with relu activations");
  if (w1_bytes > 0x7fffff00LL || w2_bytes > 0x7fffff00LL)
    TFIMM_FAIL(TFIMM_EUNSUP, "conv_chain: a weight tensor exceeds the 2 GiB a buffer descriptor addresses");
  // (TFIMM_CHAIN_LIMIT lowers the threshold: lets a test exercise the chunking on a small batch)
  static const int64_t limit = getenv("TFIMM_CHAIN_LIMIT") ? std::min<int64_t>(atoll(getenv("TFIMM_CHAIN_LIMIT")), 0x7fffff00LL) : 0x7fffff00LL;
  if (M > limit || x_bytes > limit || out_bytes > limit || res_bytes > limit) {
    // An activation tensor beyond the 2 GiB a buffer descriptor addresses (ResNet-50 stage 1 from batch 1338 on: 56 x 56 x 256
    // bf16 per image): images are independent (stride 1, zero border inside each image), so the batch is run as image
    // chunks that each fit -- as tfimm_hip_gemm does with the rows of an oversize dense layer.
    const int64_t per_img = std::max<int64_t>(std::max<int64_t>((int64_t)d.H * d.W * d.Cin * 2, (int64_t)d.OH * d.OW * d.ldc * 2),
                                              d.residual ? (int64_t)d.OH * d.OW * d.ldr * 2 : 0);
    const int64_t chunk = limit / per_img;
    if (chunk < 1 || d.B <= 1) TFIMM_FAIL(TFIMM_EUNSUP, "conv_chain: one image exceeds the 2 GiB a buffer descriptor addresses");
    for (int64_t b0 = 0; b0 < d.B; b0 += chunk) {
      tfimm_chain_desc c = d;
      c.B = (int32_t)std::min<int64_t>(chunk, d.B - b0);
      c.x = (const char*)d.x + b0 * d.H * d.W * d.Cin * 2;
      c.out = (char*)d.out + b0 * d.OH * d.OW * d.ldc * 2;
      if (d.residual) c.residual = (const char*)d.residual + b0 * d.OH * d.OW * d.ldr * 2;
      if (d.ds_x) c.ds_x = (const char*)d.ds_x + b0 * d.OH * d.OW * d.ds_cin * 2;
      const int rc = tfimm_hip_conv_chain(&c, stream);
      if (rc != 0) return rc;
    }
    return 0;
  }

  ChainArgs a;
  a.x = (const bf16_t*)d.x; a.w1 = (const bf16_t*)d.w1; a.b1 = d.b1;
  a.w2 = (const bf16_t*)d.w2; a.b2 = d.b2; a.residual = (const bf16_t*)d.residual; a.out = (bf16_t*)d.out;
  a.ds_x = (const bf16_t*)d.ds_x; a.ds_w = (const uint4*)d.ds_w; a.ds_bytes = ds ? (unsigned)(M * 128) : 0u;
  a.M = (int)M; a.N2 = d.N2;
  a.B = d.B; a.H = d.H; a.W = d.W;
  a.ldw1 = d.ldw1; a.ldw2 = d.ldw2; a.ldr = d.ldr; a.ldc = d.ldc;
  a.act1 = d.act1; a.act2 = d.act2;
  a.x_bytes = (unsigned)x_bytes; a.w1_bytes = (unsigned)w1_bytes; a.w2_bytes = (unsigned)w2_bytes;
  a.out_bytes = (unsigned)out_bytes; a.res_bytes = (unsigned)res_bytes;
  a.n_tiles = (int)cdiv64(M, 128);
  {
    static const int dbg = getenv("TFIMM_CHAIN_DBG") ? atoi(getenv("TFIMM_CHAIN_DBG")) : 0;
    a.dbg = dbg;
  }

  const int vi = d.N2 == 512;
  const int ri = (d.act1 == TFIMM_ACT_RELU && d.act2 == TFIMM_ACT_RELU) ? 1 : 0;      // the ResNet case: activations compiled in
  const gemm_chain_fn fn = ds ? (vi ? gemm_chain_kernel<8, TFIMM_ACT_RELU, true> : gemm_chain_kernel<4, TFIMM_ACT_RELU, true>)
                           : ri ? (vi ? gemm_chain_kernel<8, TFIMM_ACT_RELU> : gemm_chain_kernel<4, TFIMM_ACT_RELU>)
                                : (vi ? gemm_chain_kernel<8> : gemm_chain_kernel<4>);
  const int lds = ChainGeom::LDS_BYTES;
  static tfimm_once_t ready[2][3];
  const int fi = ds ? 2 : ri;
  if (ready[vi][fi].need()) {
    TFIMM_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    ready[vi][fi].mark();
  }
  int64_t grid = ((int64_t)chain_num_cu() * 2 + 7) / 8 * 8;      // two 4-wave workgroups per CU (80 KiB of LDS each)
  const int64_t need = ((int64_t)a.n_tiles + 7) / 8 * 8;
  if (grid > need) grid = need;
  TFIMM_LAUNCH(fn, dim3((unsigned)grid), dim3(256), (size_t)lds, (hipStream_t)stream, a);
  return 0;
}
