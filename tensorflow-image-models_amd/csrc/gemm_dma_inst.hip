// Instantiates the LDS-DMA GEMM flavours of ONE tile shape (-DTILE_ID=n).
#include "gemm_dma_kernel.h"

#ifndef TILE_ID
#error "compile with -DTILE_ID=<n>"
#endif

namespace tfimm_gemm {

#define TFIMM_SELECT(ID, BM_, BN_, WM_, WN_)                     \
  template <int I>                                               \
  struct DmaTileOf##ID {                                         \
    static constexpr int bm = BM_, bn = BN_, wm = WM_, wn = WN_; \
  };
TFIMM_GEMM_DMA_TILES(TFIMM_SELECT)
#undef TFIMM_SELECT

#define TFIMM_CAT_(a, b) a##b
#define TFIMM_CAT(a, b) TFIMM_CAT_(a, b)
using T = TFIMM_CAT(DmaTileOf, TILE_ID)<0>;

extern "C" __attribute__((visibility("hidden"))) const DmaTileCfg TFIMM_CAT(tfimm_gemm_dma_tile_, TILE_ID) = {
    T::bm, T::bn, T::wm* T::wn * 64,
    {gemm_dma_kernel<T::bm, T::bn, T::wm, T::wn, K_DENSE>, gemm_dma_kernel<T::bm, T::bn, T::wm, T::wn, K_CONV>}};

}  // namespace tfimm_gemm
