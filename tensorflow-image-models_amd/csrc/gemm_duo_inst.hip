// Instantiates the 256x128 two-workgroups-per-CU GEMM (stream-family tile id 9, gemm_duo_kernel.h).  Its catch-all
// flavour (ragged N, unaligned rows, fp32 output) is served by the 256x128 stream tile (gemm.hip redirects).
#include "gemm_duo_kernel.h"

namespace tfimm_gemm {

extern "C" __attribute__((visibility("hidden"))) const StreamTileCfg tfimm_gemm_stream_tile_9 = {
    256, 128, 256, DuoGeom::LDS_BYTES,
    {{nullptr, gemm_duo_kernel<K_DENSE, 0>, gemm_duo_kernel<K_DENSE, 1>},
     {nullptr, gemm_duo_kernel<K_CONV, 0>, gemm_duo_kernel<K_CONV, 1>}},
    {nullptr, nullptr, nullptr},
    {nullptr, nullptr},                // (no second-operand flavour)
    gemm_duo_kernel<K_DENSE, 2>,
    0};

}  // namespace tfimm_gemm
