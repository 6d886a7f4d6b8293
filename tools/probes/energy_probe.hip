// Energy coefficients of one MI355X: each load below runs back to back for PROBE_SUSTAIN_MS between CLOCK_MONOTONIC stamps;
// tools/power_probe.py cuts its amdsmi samples (socket power, shader clock) to the window.  joules per unit =
// (power - idle power) / rate.  The table (profiles/r05_power.md) is what the "energy roofline" of DESIGN.md 3.0 is built from.
//   energy_probe <mode>
//     hbm_read / hbm_read_zero   dwordx4 loads of a 2-GiB buffer (random / zero contents), one add per dword
//     hbm_write                  dwordx4 stores of a per-lane pattern over a 2-GiB buffer
//     l2_read                    the same loads over a 2-MiB block per XCD-sized group (hits in L2)
//     lds_read                   ds_read_b128 of random LDS contents, no bank conflicts
//     valu_pk_fma / valu_fma     v_pk_fma_f32 / v_fma_f32, 8 independent chains per lane, random operands
//     valu_exp                   v_exp_f32
//     mfma_rand / mfma_const     v_mfma_f32_32x32x16_bf16, registers only, 8 accumulators
//   hipcc --offload-arch=gfx950 -O3 tools/probes/energy_probe.hip -o tools/probes/bin/energy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__global__ void __launch_bounds__(256) k_read(const u32x4* __restrict__ p, size_t n_vec, unsigned* out, size_t wrap) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  unsigned s = 0;
  for (; i < n_vec; i += stride) {
    const u32x4 v = p[wrap ? (i % wrap) : i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 0x12345u) out[0] = s;
}
__global__ void __launch_bounds__(256) k_write(u32x4* __restrict__ p, size_t n_vec, unsigned seed) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  unsigned h = (unsigned)i * 2654435761u + seed;
  for (; i < n_vec; i += stride) {
    h = h * 1664525u + 1013904223u;
    p[i] = u32x4{h, h ^ 0x9e3779b9u, h * 3u, ~h};
  }
}
__global__ void __launch_bounds__(256) k_lds(unsigned* out, int iters) {
  __shared__ u32x4 buf[4096];      // 64 KiB
  unsigned h = threadIdx.x * 2654435761u + 12345u;
  for (int i = threadIdx.x; i < 4096; i += 256) { h = h * 1664525u + 1013904223u; buf[i] = u32x4{h, ~h, h * 7u, h ^ 0x5bd1e995u}; }
  __syncthreads();
  unsigned s = 0;
  unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&buf[threadIdx.x];
  for (int it = 0; it < iters; ++it) {
    u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:4096\n\tds_read_b128 %2, %8 offset:8192\n\tds_read_b128 %3, %8 offset:12288\n\t"
                 "ds_read_b128 %4, %8 offset:16384\n\tds_read_b128 %5, %8 offset:20480\n\tds_read_b128 %6, %8 offset:24576\n\tds_read_b128 %7, %8 offset:28672\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(addr) : "memory");
    s += v0.x ^ v1.y ^ v2.z ^ v3.w ^ v4.x ^ v5.y ^ v6.z ^ v7.w;
  }
  if (s == 0x12345u) out[0] = s;
}
template <int OP>
__global__ void __launch_bounds__(256) k_valu(float* out, int iters) {
  float v[8];
  f32x2 p[8];
  unsigned h = (threadIdx.x + blockIdx.x * 256 + 1) * 2654435761u;
  float a, b;
  f32x2 pa, pb;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    h = h * 1664525u + 1013904223u; v[i] = ((int)(h >> 9) % 2001 - 1000) * 1e-3f;
    h = h * 1664525u + 1013904223u; p[i] = f32x2{v[i], ((int)(h >> 9) % 2001 - 1000) * 1e-3f};
  }
  h = h * 1664525u + 1013904223u; a = 0.999f + (h & 1023) * 1e-6f; b = ((int)(h >> 12) % 2001 - 1000) * 1e-4f;
  pa = f32x2{a, 0.9991f}; pb = f32x2{b, -b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
        if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
        if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(512) k_mfma(float* out, int iters, int rnd) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
  if (rnd) {
    unsigned h = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
    for (int e = 0; e < 8; ++e) {
      h = h * 1664525u + 1013904223u; a[e] = (__bf16)(((int)(h >> 9) % 2001 - 1000) * 1e-3f);
      h = h * 1664525u + 1013904223u; b[e] = (__bf16)(((int)(h >> 9) % 2001 - 1000) * 1e-3f);
    }
  }
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "hbm_read";
  const double want = getenv("PROBE_SUSTAIN_MS") ? atof(getenv("PROBE_SUSTAIN_MS")) * 1e-3 : 1.0;
  const size_t bytes = (size_t)2 << 30, n_vec = bytes / 16;
  unsigned* out; (void)hipMalloc(&out, 1 << 22);
  u32x4* big = nullptr;
  double units = 0; const char* unit = "";
  auto launch = [&]() {
    if (mode == "hbm_read" || mode == "hbm_read_zero") { hipLaunchKernelGGL(k_read, dim3(256 * 16), dim3(256), 0, 0, big, n_vec, out, (size_t)0); units = (double)bytes; unit = "GB/s"; }
    else if (mode == "l2_read") { hipLaunchKernelGGL(k_read, dim3(256 * 16), dim3(256), 0, 0, big, n_vec, out, (size_t)(2 << 20) / 16); units = (double)bytes; unit = "GB/s"; }
    else if (mode == "hbm_write") { hipLaunchKernelGGL(k_write, dim3(256 * 16), dim3(256), 0, 0, big, n_vec, 77u); units = (double)bytes; unit = "GB/s"; }
    else if (mode == "lds_read") { hipLaunchKernelGGL(k_lds, dim3(256 * 2), dim3(256), 0, 0, out, 20000); units = 512.0 * 256 * 20000.0 * 8 * 16; unit = "GB/s"; }
    else if (mode == "valu_fma") { hipLaunchKernelGGL(k_valu<0>, dim3(256 * 8), dim3(256), 0, 0, (float*)out, 20000); units = 2048.0 * 256 * 20000.0 * 32 * 2; unit = "GFLOP/s"; }
    else if (mode == "valu_pk_fma") { hipLaunchKernelGGL(k_valu<1>, dim3(256 * 8), dim3(256), 0, 0, (float*)out, 20000); units = 2048.0 * 256 * 20000.0 * 32 * 4; unit = "GFLOP/s"; }
    else if (mode == "valu_exp") { hipLaunchKernelGGL(k_valu<2>, dim3(256 * 8), dim3(256), 0, 0, (float*)out, 20000); units = 2048.0 * 256 * 20000.0 * 32; unit = "Gop/s"; }
    else if (mode == "mfma_rand" || mode == "mfma_const") { hipLaunchKernelGGL(k_mfma, dim3(256), dim3(512), 0, 0, (float*)out, 100000, mode == "mfma_rand" ? 1 : 0); units = 256.0 * 8 * 100000.0 * 8 * 32768.0; unit = "GFLOP/s"; }
    else { fprintf(stderr, "unknown mode %s\n", mode.c_str()); exit(2); }
  };
  if (mode.rfind("hbm", 0) == 0 || mode == "l2_read") {
    (void)hipMalloc(&big, bytes);
    if (mode == "hbm_read_zero") (void)hipMemset(big, 0, bytes);
    else hipLaunchKernelGGL(k_write, dim3(256 * 16), dim3(256), 0, 0, big, n_vec, 1u);
  }
  launch(); (void)hipDeviceSynchronize();
  const double t0 = now();
  long n = 0;
  while (now() - t0 < want) { launch(); (void)hipDeviceSynchronize(); ++n; }
  const double t1 = now();
  printf("SUSTAIN|%s|%.6f|%.6f|%.1f|%s\n", mode.c_str(), t0, t1, units * n / (t1 - t0) * 1e-9, unit);
  printf("err=%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
