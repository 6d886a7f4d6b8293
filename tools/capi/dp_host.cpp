// A data-parallel host without Python: one process per GPU, every rank runs ITS shard of the batch through a plan blob
// (tfimm.engine.graph.Plan.export, made for the shard size) and the fp32 logits of all ranks are all-gathered by the one
// collective of the path (include/tfimm_hip_dp.h: tfimm_hip_dp_forward -> ncclAllGather).  Built by
// `make -C tensorflow-image-models_amd/csrc dp_host` (hipcc; links libtfimm_hip_dp.so + libtfimm_hip.so).
//   RANK=r WORLD_SIZE=n LOCAL_RANK=d dp_host <plan.blob> <input.f32 of the GLOBAL batch> <logits.out> <id file> [repeats]
// The 128-byte RCCL id travels through <id file> (rank 0 writes it, the others wait for it): the launcher's job in a real
// deployment.  Rank 0 writes the gathered logits [global batch][classes]; tests/test_gpu_dp_capi.py compares them with the
// Python engine bit for bit.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tfimm_hip_dp.h"

static std::vector<char> slurp(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<char> b((size_t)n);
  if (fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read on %s\n", path); exit(2); }
  fclose(f);
  return b;
}

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

#define CHECK(expr, errfn)                                                                 \
  do {                                                                                     \
    const int rc_ = (expr);                                                                \
    if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #expr, rc_, errfn()); return 1; } \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: dp_host plan.blob input.f32 logits.out id_file [repeats]\n"); return 2; }
  const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), device = env_int("LOCAL_RANK", rank);
  const int repeats = argc > 5 ? atoi(argv[5]) : 1;
  const std::vector<char> blob = slurp(argv[1]);
  tfimm_plan_info info;
  CHECK(tfimm_hip_plan_query(blob.data(), blob.size(), &info), tfimm_hip_last_error);

  // ---- rendezvous: the RCCL id from rank 0 to everyone
  char id[TFIMM_HIP_DP_ID_BYTES];
  const std::string id_path = argv[4];
  if (rank == 0) {
    CHECK(tfimm_hip_dp_unique_id(id, sizeof(id)), tfimm_hip_dp_last_error);
    const std::string tmp = id_path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) { fprintf(stderr, "cannot write %s\n", tmp.c_str()); return 2; }
    fclose(f);
    if (rename(tmp.c_str(), id_path.c_str()) != 0) { fprintf(stderr, "cannot publish %s\n", id_path.c_str()); return 2; }
  } else {
    for (int tries = 0;; ++tries) {
      FILE* f = fopen(id_path.c_str(), "rb");
      if (f) {
        const size_t n = fread(id, 1, sizeof(id), f);
        fclose(f);
        if (n == sizeof(id)) break;
      }
      if (tries > 600) { fprintf(stderr, "rank %d: no RCCL id in %s after 60 s\n", rank, id_path.c_str()); return 2; }
      usleep(100000);
    }
  }
  tfimm_dp_t dp = nullptr;
  CHECK(tfimm_hip_dp_create(&dp, id, sizeof(id), world, rank, device), tfimm_hip_dp_last_error);

  // ---- this rank's shard of the global batch (equal shards: the blob's batch is the shard size)
  const size_t img_bytes = (size_t)info.in_h * info.in_w * info.in_c * sizeof(float);
  const int64_t global_batch = (int64_t)info.batch * world;
  int64_t lo = 0, hi = 0;
  CHECK(tfimm_hip_dp_shard_bounds(global_batch, world, rank, &lo, &hi), tfimm_hip_dp_last_error);
  const std::vector<char> input = slurp(argv[2]);
  if (input.size() != img_bytes * (size_t)global_batch) {
    fprintf(stderr, "input has %zu bytes, %d ranks x batch %d want %zu\n", input.size(), world, info.batch, img_bytes * (size_t)global_batch);
    return 2;
  }
  void *ws = nullptr, *x = nullptr, *gathered = nullptr;
  hipStream_t st;
  if (hipStreamCreate(&st) != hipSuccess || hipMalloc(&ws, info.workspace_bytes) != hipSuccess ||
      hipMalloc(&x, img_bytes * (size_t)(hi - lo)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  (void)hipMemcpyAsync(x, input.data() + img_bytes * (size_t)lo, img_bytes * (size_t)(hi - lo), hipMemcpyHostToDevice, st);
  tfimm_plan_t plan = nullptr;
  CHECK(tfimm_hip_plan_create(blob.data(), blob.size(), ws, st, &plan), tfimm_hip_last_error);
  void* logits = nullptr;
  int64_t rows = 0, cols = 0;
  int dtype = 0;
  CHECK(tfimm_hip_plan_output(plan, "logits", &logits, &rows, &cols, &dtype), tfimm_hip_last_error);
  if (hipMalloc(&gathered, (size_t)(rows * world * cols) * 4) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }

  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  CHECK(tfimm_hip_dp_forward(dp, plan, x, /*float32*/ 0, nullptr, 0, gathered, st), tfimm_hip_dp_last_error);      // warm-up
  (void)hipEventRecord(e0, st);
  for (int r = 0; r < repeats; ++r) CHECK(tfimm_hip_dp_forward(dp, plan, x, 0, nullptr, 0, gathered, st), tfimm_hip_dp_last_error);
  (void)hipEventRecord(e1, st);
  std::vector<char> host((size_t)(rows * world * cols) * 4);
  (void)hipMemcpyAsync(host.data(), gathered, host.size(), hipMemcpyDeviceToHost, st);
  if (hipStreamSynchronize(st) != hipSuccess) { fprintf(stderr, "forward failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  if (rank == 0) {
    FILE* f = fopen(argv[3], "wb");
    fwrite(host.data(), 1, host.size(), f);
    fclose(f);
    printf("dp_host: %d ranks x batch %d = %lld images, logits %lld x %lld f32 gathered, %.3f ms per step\n", world, info.batch,
           (long long)global_batch, (long long)(rows * world), (long long)cols, ms / repeats);
  }
  CHECK(tfimm_hip_dp_destroy(dp), tfimm_hip_dp_last_error);
  CHECK(tfimm_hip_plan_destroy(plan), tfimm_hip_last_error);
  (void)hipFree(ws); (void)hipFree(x); (void)hipFree(gathered);
  return 0;
}
