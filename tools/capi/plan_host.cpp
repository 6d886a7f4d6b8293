// A host without Python: loads a plan blob (tfimm.engine.graph.Plan.export), runs the forward through the program-level
// C entry points of include/tfimm_hip.h and writes the logits.  Built by `make -C tensorflow-image-models_amd/csrc plan_host`
// (hipcc, links libtfimm_hip.so); tests/test_gpu_plan_capi.py compares its output with the Python engine bit for bit.
//   plan_host <plan.blob> <input.f32> <logits.out> [repeats]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/tfimm_hip.h"

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

#define CHECK(expr)                                                                                   \
  do {                                                                                                \
    const int rc_ = (expr);                                                                           \
    if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #expr, rc_, tfimm_hip_last_error()); return 1; } \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: plan_host plan.blob input.f32 logits.out [repeats]\n"); return 2; }
  const int repeats = argc > 4 ? atoi(argv[4]) : 1;
  const std::vector<char> blob = slurp(argv[1]);
  const std::vector<char> input = slurp(argv[2]);
  tfimm_plan_info info;
  CHECK(tfimm_hip_plan_query(blob.data(), blob.size(), &info));
  const size_t in_bytes = (size_t)info.batch * info.in_h * info.in_w * info.in_c * sizeof(float);
  if (input.size() != in_bytes) { fprintf(stderr, "input has %zu bytes, the plan wants %zu\n", input.size(), in_bytes); return 2; }
  void *ws = nullptr, *x = nullptr;
  if (hipMalloc(&ws, info.workspace_bytes) != hipSuccess || hipMalloc(&x, in_bytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  hipStream_t st;
  (void)hipStreamCreate(&st);
  (void)hipMemcpyAsync(x, input.data(), in_bytes, hipMemcpyHostToDevice, st);
  tfimm_plan_t plan = nullptr;
  CHECK(tfimm_hip_plan_create(blob.data(), blob.size(), ws, st, &plan));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  CHECK(tfimm_hip_plan_forward(plan, x, /*in_dtype float32*/ 0, st));       // warm-up (lazy function attributes)
  (void)hipEventRecord(e0, st);
  for (int r = 0; r < repeats; ++r) CHECK(tfimm_hip_plan_forward(plan, x, 0, st));
  (void)hipEventRecord(e1, st);
  void* out = nullptr;
  int64_t rows = 0, cols = 0;
  int dtype = 0;
  CHECK(tfimm_hip_plan_output(plan, "logits", &out, &rows, &cols, &dtype));
  std::vector<char> host((size_t)rows * cols * (dtype == 1 ? 4 : 2));
  (void)hipMemcpyAsync(host.data(), out, host.size(), hipMemcpyDeviceToHost, st);
  if (hipStreamSynchronize(st) != hipSuccess) { fprintf(stderr, "forward failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  FILE* f = fopen(argv[3], "wb");
  fwrite(host.data(), 1, host.size(), f);
  fclose(f);
  printf("plan_host: batch %d, %d calls, workspace %.1f MB, logits %lld x %lld (%s), %.3f ms per forward\n", info.batch, info.n_calls,
         info.workspace_bytes / 1e6, (long long)rows, (long long)cols, dtype == 1 ? "f32" : "bf16", ms / repeats);
  CHECK(tfimm_hip_plan_destroy(plan));
  (void)hipFree(ws); (void)hipFree(x);
  return 0;
}
