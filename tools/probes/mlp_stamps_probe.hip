// Where the cycles of tfimm_hip_mlp_fused go: compiles csrc/mlp.hip with MLP_STAMPS (s_memtime of workgroup 0, waves 0 and 4,
// at the top of every group, behind its barrier + DMA requests, behind GEMM 1 and behind the activation) and prints the
// cycles of every part for a steady-state tile.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMLP_STAMPS -Iinclude -Itensorflow-image-models_amd/csrc \
//         tools/probes/mlp_stamps_probe.hip -o /tmp/mlp_probe && /tmp/mlp_probe [rows]
#include "../../tensorflow-image-models_amd/csrc/mlp.hip"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

static char g_err[512];
void tfimm_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }

int main(int argc, char** argv) {
  const long long M = argc > 1 ? atoll(argv[1]) : 802816;
  const int C = 128, H = 512;
  std::vector<unsigned short> hx((size_t)M * C), hw1((size_t)H * C), hw2((size_t)C * H);
  std::vector<float> hb1(H, 0.1f), hb2(C, 0.2f);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (unsigned short)(0x3f00u + ((s >> 16) & 0xffu) + ((s >> 9) & 0x8000u)); };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw1) v = (unsigned short)(rnd() - 0x0300u);
  for (auto& v : hw2) v = (unsigned short)(rnd() - 0x0380u);
  void *x, *w1, *w2, *out; float *b1, *b2; long long* st;
  hipMalloc(&x, hx.size() * 2); hipMalloc(&out, hx.size() * 2); hipMalloc(&w1, hw1.size() * 2); hipMalloc(&w2, hw2.size() * 2);
  hipMalloc(&b1, H * 4); hipMalloc(&b2, C * 4); hipMalloc(&st, 4096 * 8);
  hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(w1, hw1.data(), hw1.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w2, hw2.data(), hw2.size() * 2, hipMemcpyHostToDevice); hipMemcpy(b1, hb1.data(), H * 4, hipMemcpyHostToDevice);
  hipMemcpy(b2, hb2.data(), C * 4, hipMemcpyHostToDevice); hipMemset(st, 0, 4096 * 8);
  tfimm_mlp_stamps = st;
  tfimm_mlp_desc d{};
  d.x = x; d.w1 = w1; d.b1 = b1; d.w2 = w2; d.b2 = b2; d.residual = x; d.out = out; d.M = M; d.C = C; d.hidden = H;
  d.act = TFIMM_ACT_GELU; d.eps = 1e-5f;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0, 0);
    if (tfimm_hip_mlp_fused(&d, nullptr) != 0) { printf("launch failed: %s\n", g_err); return 1; }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("run %d: %.1f us  (%.1f TFLOP/s)\n", it, ms * 1e3, 4.0 * M * C * H / (ms * 1e-3) / 1e12);
  }
  std::vector<long long> h(4096);
  hipMemcpy(h.data(), st, 4096 * 8, hipMemcpyDeviceToHost);
  // stamps per group: [top, behind barrier + requests, behind GEMM 1, behind the activation]; + one in front of the stores
  for (int half = 0; half < 2; ++half) {
    const long long* t = h.data() + half * 2048;
    const int tile = 1, per_tile = 8 * 4 + 1;       // second tile of the workgroup: steady state
    const int base = tile * per_tile;
    printf("wave %d, tile %d: group  wait+barrier+requests  [normalise+]GEMM1  activation  GEMM2   (cycles)\n", half * 4, tile);
    for (int g = 0; g < 8; ++g) {
      const long long* q = t + base + 4 * g;
      printf("  %d  %7lld  %7lld  %7lld  %7lld\n", g, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] - q[3]);
    }
    printf("  stores %lld;  tile total %lld cycles\n", t[base + per_tile] - t[base + per_tile - 1], t[base + per_tile] - t[base]);
  }
  return 0;
}
