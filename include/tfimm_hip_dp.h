/* tfimm_hip_dp.h -- data-parallel forward over the GPUs of one node behind a C ABI (libtfimm_hip_dp.so).
 *
 * The reference has no multi-GPU path at all (tfimm/train/trainer.py:72 is `SingleGPUTrainer`; no tf.distribute, no
 * collective anywhere: SURVEY.md 2.2 / 8e).  What a data-parallel tfimm forward needs is ONE exchange step: images are
 * independent at inference (every norm call passes training=training, e.g. tfimm/architectures/resnet.py:270,275,281), so each
 * rank runs `model(x[lo:hi], training=False)` (tfimm/models/factory.py:18-125 create_model + the model's call) on its
 * contiguous shard and the fp32 logits [B / G, nb_classes] of all ranks are all-gathered -- the `ncclAllGather` call site
 * SURVEY.md 8b sketches as tfimm_hip_dp_create / tfimm_hip_dp_forward.  RCCL over xGMI; one PROCESS per GPU (the launch
 * contract of bench.py), so "create" takes (world, rank) and the 128-byte RCCL id that rank 0 made and the host's own
 * rendezvous carried to the other ranks (torch.distributed in tfimm/engine/dp.py, a file in tools/capi/dp_host.cpp).
 *
 * Conventions as in tfimm_hip.h: extern "C", plain pointers and integers, the caller owns every device buffer, every call is
 * asynchronous on the hipStream_t passed as `stream` (and capturable into a hipGraph as far as RCCL's all-gather is),
 * int status (0 = ok) + tfimm_hip_dp_last_error().  This library is separate from libtfimm_hip.so so that the kernels carry
 * no RCCL dependency; it links librccl.so.1 and libtfimm_hip.so (for tfimm_hip_plan_forward / _output).
 */
#ifndef TFIMM_HIP_DP_H
#define TFIMM_HIP_DP_H
#include <stddef.h>
#include <stdint.h>

#include "tfimm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define TFIMM_HIP_DP_ABI_VERSION 1
#define TFIMM_HIP_DP_ID_BYTES 128 /* sizeof(ncclUniqueId) */

typedef struct tfimm_hip_dp* tfimm_dp_t;

TFIMM_API int tfimm_hip_dp_abi_version(void);
TFIMM_API const char* tfimm_hip_dp_last_error(void);

/* Contiguous shard [lo, hi) of `rank`: the first batch % world ranks get one extra image (tfimm/engine/dp.py shard_bounds). */
TFIMM_API int tfimm_hip_dp_shard_bounds(int64_t batch, int world, int rank, int64_t* lo, int64_t* hi);

/* Rank 0: a fresh RCCL id (ncclGetUniqueId) into id[TFIMM_HIP_DP_ID_BYTES]; the host hands the bytes to every rank. */
TFIMM_API int tfimm_hip_dp_unique_id(void* id, size_t bytes);

/* Every rank (collective): communicator of `world` ranks on HIP device `device` (ncclCommInitRank).  world = 1 is valid. */
TFIMM_API int tfimm_hip_dp_create(tfimm_dp_t* dp, const void* id, size_t id_bytes, int world, int rank, int device);
TFIMM_API int tfimm_hip_dp_world(tfimm_dp_t dp, int* world, int* rank);

/* THE exchange step: gathered[r * rows .. (r + 1) * rows) = rank r's `local` rows, fp32, `cols` values per row, on every rank
 * (ncclAllGather of rows * cols floats per rank, enqueued on `stream`).  Ragged shards: every rank passes the LARGEST shard's
 * row count and pads its own rows (tfimm_hip_dp_forward does that by itself). */
TFIMM_API int tfimm_hip_dp_all_gather_logits(tfimm_dp_t dp, const void* local, void* gathered, int64_t rows, int64_t cols,
                                             void* stream);

/* One data-parallel forward of this rank: tfimm_hip_plan_forward(plan, x_shard) -- the plan was exported for THIS rank's shard
 * size -- then its fp32 logits are padded to `max_rows` rows (the largest shard; pass 0 for "the plan's batch") and
 * all-gathered into gathered[world * max_rows][nb_classes]; rank r's valid rows start at r * max_rows.  `staging` is a device
 * buffer of max_rows * nb_classes floats (the send block; NULL when the shard is not ragged and the logits are sent straight
 * from the plan's workspace). */
TFIMM_API int tfimm_hip_dp_forward(tfimm_dp_t dp, tfimm_plan_t plan, const void* x_shard, int in_dtype, void* staging,
                                   int64_t max_rows, void* gathered, void* stream);

TFIMM_API int tfimm_hip_dp_destroy(tfimm_dp_t dp);

#ifdef __cplusplus
}
#endif
#endif /* TFIMM_HIP_DP_H */
