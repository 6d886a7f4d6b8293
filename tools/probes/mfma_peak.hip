// MFMA-only rate probe: what the matrix pipes sustain on this box (clock included).
//   mfma_peak <mode>   mode 0: 8 independent accumulators; 1: 2 accumulators (dependency distance 2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ void __launch_bounds__(512) k(float* out, int iters, long long* clk, int rnd) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
  if (rnd) {   // random-looking operands: realistic switching activity (power -> clock)
    unsigned h = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
    for (int e = 0; e < 8; ++e) {
      h = h * 1664525u + 1013904223u; a[e] = (__bf16)(((int)(h >> 9) % 2001 - 1000) * 1e-3f);
      h = h * 1664525u + 1013904223u; b[e] = (__bf16)(((int)(h >> 9) % 2001 - 1000) * 1e-3f);
    }
  }
  long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

// the same for v_mfma_f32_16x16x32_bf16 (modes 10 / 11 / 12: 8 / 2 / 4 accumulators)
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NACC>
__global__ void __launch_bounds__(512) k16(float* out, int iters, long long* clk, int rnd) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
  long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

int main(int argc, char** argv) {
  int mode = argc > 1 ? atoi(argv[1]) : 0;
  int blocks = argc > 2 ? atoi(argv[2]) : 256;
  int rnd = argc > 3 ? atoi(argv[3]) : 0;
  int iters = 200000;
  int threads = argc > 4 ? atoi(argv[4]) : 512;
  float* out; long long* clk;
  hipMalloc(&out, blocks * 512 * 4); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    if (mode == 10) k16<8><<<blocks, threads>>>(out, iters, clk, rnd); else if (mode == 11) k16<2><<<blocks, threads>>>(out, iters, clk, rnd);
    else if (mode == 12) k16<4><<<blocks, threads>>>(out, iters, clk, rnd); else
    if (mode == 0) k<8><<<blocks, threads>>>(out, iters, clk, rnd); else if (mode == 1) k<2><<<blocks, threads>>>(out, iters, clk, rnd); else k<4><<<blocks, threads>>>(out, iters, clk, rnd);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double flops = (double)blocks * (threads / 64) * iters * 8 * (mode >= 10 ? 16384.0 : 32768.0);
    printf("mode %d blocks %d: %.2f ms %.1f TF/s; clock64 %lld wall %lld -> shader clock %.0f MHz (if wall=100MHz); cycles/MFMA/wave %.1f\n",
           mode, blocks, ms, flops / ms / 1e9, h[0], h[1], 100.0 * h[0] / h[1], (double)h[0] / (iters * 8.0));
  }
  // PROBE_SUSTAIN_MS: the same launch back to back for that long between CLOCK_MONOTONIC stamps (tools/power_probe.py)
  if (const char* sm = getenv("PROBE_SUSTAIN_MS")) {
    const double want = atof(sm) * 1e-3;
    auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
    hipDeviceSynchronize();
    const double t0 = now();
    long n = 0;
    while (now() - t0 < want) {
      if (mode == 10) k16<8><<<blocks, threads>>>(out, iters, clk, rnd); else if (mode == 11) k16<2><<<blocks, threads>>>(out, iters, clk, rnd);
      else if (mode == 12) k16<4><<<blocks, threads>>>(out, iters, clk, rnd); else
      if (mode == 0) k<8><<<blocks, threads>>>(out, iters, clk, rnd); else if (mode == 1) k<2><<<blocks, threads>>>(out, iters, clk, rnd); else k<4><<<blocks, threads>>>(out, iters, clk, rnd);
      hipDeviceSynchronize();
      ++n;
    }
    const double t1 = now();
    const double flops = (double)blocks * (threads / 64) * iters * 8 * (mode >= 10 ? 16384.0 : 32768.0);
    printf("SUSTAIN|mfma_peak mode %d %s operands|%.6f|%.6f|%.1f\n", mode, rnd ? "random" : "constant", t0, t1, flops * n / (t1 - t0) * 1e-12);
  }
  return 0;
}
