// tfimm_hip_dp_*: the data-parallel forward's ONE exchange step behind a C ABI (include/tfimm_hip_dp.h, libtfimm_hip_dp.so).
// One process per GPU; each rank runs its contiguous shard of the batch through a plan (csrc/plan.hip) and the fp32 logits
// of all ranks are all-gathered by RCCL over xGMI -- the single `ncclAllGather` call site of SURVEY.md 8b / 8e.  Nothing
// here touches a kernel: the library exists so that a host without Python (tools/capi/dp_host.cpp) has the same DP path
// as tfimm/engine/dp.py, and so that libtfimm_hip.so itself needs no RCCL.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/tfimm_hip_dp.h"

static_assert(sizeof(ncclUniqueId) == TFIMM_HIP_DP_ID_BYTES, "ncclUniqueId is 128 bytes");

struct tfimm_hip_dp {
  ncclComm_t comm;
  int world, rank, device;
};

namespace {

thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

#define DP_FAIL(code, ...)   \
  do {                       \
    set_error(__VA_ARGS__);  \
    return (code);           \
  } while (0)
#define DP_NCCL(expr)                                                                        \
  do {                                                                                       \
    const ncclResult_t r_ = (expr);                                                          \
    if (r_ != ncclSuccess) DP_FAIL(1000 + (int)r_, "%s failed: %s", #expr, ncclGetErrorString(r_)); \
  } while (0)
#define DP_HIP(expr)                                                                         \
  do {                                                                                       \
    const hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) DP_FAIL((int)e_, "%s failed: %s", #expr, hipGetErrorString(e_));   \
  } while (0)

}  // namespace

extern "C" {

int tfimm_hip_dp_abi_version(void) { return TFIMM_HIP_DP_ABI_VERSION; }
const char* tfimm_hip_dp_last_error(void) { return g_err; }

int tfimm_hip_dp_shard_bounds(int64_t batch, int world, int rank, int64_t* lo, int64_t* hi) {
  if (batch < 0 || world <= 0 || rank < 0 || rank >= world || !lo || !hi)
    DP_FAIL(TFIMM_EINVAL, "dp_shard_bounds: batch=%lld world=%d rank=%d", (long long)batch, world, rank);
  const int64_t q = batch / world, r = batch % world;
  *lo = rank * q + (rank < r ? rank : r);
  *hi = *lo + q + (rank < r ? 1 : 0);
  return 0;
}

int tfimm_hip_dp_unique_id(void* id, size_t bytes) {
  if (!id || bytes < sizeof(ncclUniqueId)) DP_FAIL(TFIMM_EINVAL, "dp_unique_id: need a buffer of %zu bytes", sizeof(ncclUniqueId));
  ncclUniqueId u;
  DP_NCCL(ncclGetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return 0;
}

int tfimm_hip_dp_create(tfimm_dp_t* dp, const void* id, size_t id_bytes, int world, int rank, int device) {
  if (!dp || !id || id_bytes < sizeof(ncclUniqueId)) DP_FAIL(TFIMM_EINVAL, "dp_create: null handle / id of fewer than %zu bytes", sizeof(ncclUniqueId));
  if (world <= 0 || rank < 0 || rank >= world) DP_FAIL(TFIMM_EINVAL, "dp_create: world=%d rank=%d", world, rank);
  DP_HIP(hipSetDevice(device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclComm_t comm = nullptr;
  DP_NCCL(ncclCommInitRank(&comm, world, u, rank));
  *dp = new tfimm_hip_dp{comm, world, rank, device};
  return 0;
}

int tfimm_hip_dp_world(tfimm_dp_t dp, int* world, int* rank) {
  if (!dp) DP_FAIL(TFIMM_EINVAL, "dp_world: null handle");
  if (world) *world = dp->world;
  if (rank) *rank = dp->rank;
  return 0;
}

int tfimm_hip_dp_all_gather_logits(tfimm_dp_t dp, const void* local, void* gathered, int64_t rows, int64_t cols, void* stream) {
  if (!dp || !local || !gathered) DP_FAIL(TFIMM_EINVAL, "dp_all_gather_logits: null handle / buffer");
  if (rows <= 0 || cols <= 0 || rows * cols > (int64_t)1 << 40) DP_FAIL(TFIMM_EINVAL, "dp_all_gather_logits: rows=%lld cols=%lld", (long long)rows, (long long)cols);
  // the one collective of the path: every rank's [rows][cols] fp32 block -> [world][rows][cols] on every rank
  DP_NCCL(ncclAllGather(local, gathered, (size_t)(rows * cols), ncclFloat32, dp->comm, (hipStream_t)stream));
  return 0;
}

int tfimm_hip_dp_forward(tfimm_dp_t dp, tfimm_plan_t plan, const void* x_shard, int in_dtype, void* staging, int64_t max_rows,
                         void* gathered, void* stream) {
  if (!dp || !plan || !x_shard || !gathered) DP_FAIL(TFIMM_EINVAL, "dp_forward: null handle / buffer");
  int rc = tfimm_hip_plan_forward(plan, x_shard, in_dtype, stream);
  if (rc != 0) DP_FAIL(rc, "dp_forward: plan_forward: %s", tfimm_hip_last_error());
  void* logits = nullptr;
  int64_t rows = 0, cols = 0;
  int dtype = 0;
  rc = tfimm_hip_plan_output(plan, "logits", &logits, &rows, &cols, &dtype);
  if (rc != 0) DP_FAIL(rc, "dp_forward: plan_output: %s", tfimm_hip_last_error());
  if (dtype != 1) DP_FAIL(TFIMM_EUNSUP, "dp_forward: the plan's logits are not fp32");
  if (max_rows <= 0) max_rows = rows;
  if (rows > max_rows) DP_FAIL(TFIMM_EINVAL, "dp_forward: this rank's shard has %lld rows, max_rows is %lld", (long long)rows, (long long)max_rows);
  const void* send = logits;
  if (rows < max_rows || staging) {
    // ragged shard (or the caller wants the workspace free for the next forward right away): rows into the send block, zero padding
    if (!staging) DP_FAIL(TFIMM_EINVAL, "dp_forward: a shard of %lld rows padded to %lld needs a staging buffer", (long long)rows, (long long)max_rows);
    DP_HIP(hipMemcpyAsync(staging, logits, (size_t)(rows * cols) * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (rows < max_rows)
      DP_HIP(hipMemsetAsync((char*)staging + (size_t)(rows * cols) * 4, 0, (size_t)((max_rows - rows) * cols) * 4, (hipStream_t)stream));
    send = staging;
  }
  return tfimm_hip_dp_all_gather_logits(dp, send, gathered, max_rows, cols, stream);
}

int tfimm_hip_dp_destroy(tfimm_dp_t dp) {
  if (!dp) return 0;
  const ncclResult_t r = ncclCommDestroy(dp->comm);
  delete dp;
  if (r != ncclSuccess) DP_FAIL(1000 + (int)r, "ncclCommDestroy failed: %s", ncclGetErrorString(r));
  return 0;
}

}  // extern "C"
