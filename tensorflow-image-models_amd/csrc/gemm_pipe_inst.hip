// Instantiates the 256x256 deep-ring GEMM (stream-family tile id 7).  Its catch-all flavour
// (ragged N, unaligned rows) is the stream kernel of the same tile shape.
#include "gemm_pipe_kernel.h"

namespace tfimm_gemm {

extern "C" __attribute__((visibility("hidden"))) const StreamTileCfg tfimm_gemm_stream_tile_7 = {
    256, 256, 512, 2 * (256 + 256) * 128,
    {{gemm_stream_kernel<256, 256, 2, 4, K_DENSE, false>, gemm_pipe_kernel<K_DENSE>, gemm_pipe_kernel<K_DENSE>},
     {gemm_stream_kernel<256, 256, 2, 4, K_CONV, false>, gemm_pipe_kernel<K_CONV>, gemm_pipe_kernel<K_CONV>}},
    {nullptr, nullptr, nullptr}};

}  // namespace tfimm_gemm
