// Instantiates the persistent LDS-DMA GEMM flavours of ONE tile shape (-DTILE_ID=n).
#include "gemm_stream_kernel.h"

#ifndef TILE_ID
#error "compile with -DTILE_ID=<n>"
#endif

namespace tfimm_gemm {

#define TFIMM_SELECT(ID, BM_, BN_, WM_, WN_)                     \
  template <int I>                                               \
  struct StreamTileOf##ID {                                      \
    static constexpr int bm = BM_, bn = BN_, wm = WM_, wn = WN_; \
  };
TFIMM_GEMM_STREAM_TILES(TFIMM_SELECT)
#undef TFIMM_SELECT

#define TFIMM_CAT_(a, b) a##b
#define TFIMM_CAT(a, b) TFIMM_CAT_(a, b)
using T = TFIMM_CAT(StreamTileOf, TILE_ID)<0>;
using G = StreamGeom<T::bm, T::bn, T::wm, T::wn>;

extern "C" __attribute__((visibility("hidden"))) const StreamTileCfg TFIMM_CAT(tfimm_gemm_stream_tile_, TILE_ID) = {
    T::bm, T::bn, T::wm* T::wn * 64, G::LDS_BYTES,
    {{gemm_stream_kernel<T::bm, T::bn, T::wm, T::wn, K_DENSE, false>, gemm_stream_kernel<T::bm, T::bn, T::wm, T::wn, K_DENSE, true>,
      gemm_stream_kernel<T::bm, T::bn, T::wm, T::wn, K_DENSE, true, false, true>},
     {gemm_stream_kernel<T::bm, T::bn, T::wm, T::wn, K_CONV, false>, gemm_stream_kernel<T::bm, T::bn, T::wm, T::wn, K_CONV, true>,
      gemm_stream_kernel<T::bm, T::bn, T::wm, T::wn, K_CONV, true, false, true>}},
    {gemm_stream_kernel<T::bm, T::bn, T::wm, T::wn, K_DENSE, false, true>, gemm_stream_kernel<T::bm, T::bn, T::wm, T::wn, K_DENSE, true, true>,
     gemm_stream_kernel<T::bm, T::bn, T::wm, T::wn, K_DENSE, true, true, true>},
    {gemm_stream_kernel<T::bm, T::bn, T::wm, T::wn, K_DENSE, true, false, true, false, true>,
     gemm_stream_kernel<T::bm, T::bn, T::wm, T::wn, K_CONV, true, false, true, false, true>},
    gemm_stream_kernel<T::bm, T::bn, T::wm, T::wn, K_DENSE, true, false, true, true>,
    G::NW * (1 + G::WTN / 32) * 1024};

}  // namespace tfimm_gemm
