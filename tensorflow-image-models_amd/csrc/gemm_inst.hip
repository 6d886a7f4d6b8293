// Instantiates the GEMM kernel flavours of ONE tile shape (-DTILE_ID=n); compiled once per
// entry of TFIMM_GEMM_TILES so the heavy template expansions build in parallel.
#include "gemm_kernel.h"

#ifndef TILE_ID
#error "compile with -DTILE_ID=<n>"
#endif

namespace tfimm_gemm {

#define TFIMM_SELECT(ID, BM_, BN_, WM_, WN_)                                                   \
  template <int I>                                                                             \
  struct TileOf##ID {                                                                          \
    static constexpr int bm = BM_, bn = BN_, wm = WM_, wn = WN_;                               \
  };
TFIMM_GEMM_TILES(TFIMM_SELECT)
#undef TFIMM_SELECT

#define TFIMM_CAT_(a, b) a##b
#define TFIMM_CAT(a, b) TFIMM_CAT_(a, b)
using T = TFIMM_CAT(TileOf, TILE_ID)<0>;

// scalar-load and SE-scale flavours only exist for the small/narrow tiles that need them
constexpr bool kHasScalar = (T::bm == 64 && T::bn == 64);
constexpr bool kHasScale = (T::bn <= 128 && T::bm <= 128);

template <int KM, bool ENABLE>
struct Pick {
  static constexpr gemm_fn fn = gemm_kernel<T::bm, T::bn, T::wm, T::wn, KM>;
};
template <int KM>
struct Pick<KM, false> {
  static constexpr gemm_fn fn = nullptr;
};

extern "C" __attribute__((visibility("hidden"))) const TileCfg TFIMM_CAT(tfimm_gemm_tile_, TILE_ID) = {
    T::bm, T::bn, T::wm* T::wn * 64,
    {Pick<K_DENSE, true>::fn, Pick<K_CONV, true>::fn, Pick<K_CONV_C4, true>::fn,
     Pick<K_DENSE_SCALAR, kHasScalar>::fn, Pick<K_DENSE_SCALE, kHasScale>::fn,
     Pick<K_CONV_SCALAR, kHasScalar>::fn}};

}  // namespace tfimm_gemm
