// VALU issue-rate probe (gfx950): how many cycles does a SIMD spend per wave64 instruction of
//   v_fma_f32, v_pk_fma_f32, v_exp_f32, v_rcp_f32
// and per 8-element swish in the two forms discussed in DESIGN.md 3.8 (8 reciprocals / 2 shared reciprocals)?
// One workgroup of 256 threads per CU (one wave per SIMD) or 1024 (four per SIMD); 8 independent chains per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate_probe.hip -o /tmp/valu && /tmp/valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void probe(float* out, int iters, long long* clk) {
  float v[8];
  f32x2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = 0.5f + 0.001f * (threadIdx.x + i); p[i] = f32x2{v[i], v[i] + 0.25f}; }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
      if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
      if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      if (OP == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
      if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
    }
    if (OP == 5) {   // swish on 4 pairs, one reciprocal per element
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x2 z = p[e] * -1.4426950408889634f;
        const f32x2 en = {__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)};
        const f32x2 dn = 1.f + en;
        const f32x2 sg = {__builtin_amdgcn_rcpf(dn.x), __builtin_amdgcn_rcpf(dn.y)};
        p[e] = p[e] * sg + 1.f;
      }
      asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]));
    }
    if (OP == 6) {   // swish on 4 pairs, two shared reciprocals
      f32x2 t[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f32x2 z = p[e] * -1.4426950408889634f;
        z.x = fminf(z.x, 31.f); z.y = fminf(z.y, 31.f);
        t[e] = 1.f + f32x2{__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)};
      }
      const f32x2 m01 = t[0] * t[1], m23 = t[2] * t[3], pr = m01 * m23;
      const f32x2 rp = {__builtin_amdgcn_rcpf(pr.x), __builtin_amdgcn_rcpf(pr.y)};
      const f32x2 u01 = rp * m23, u23 = rp * m01;
      p[0] = p[0] * (u01 * t[1]) + 1.f; p[1] = p[1] * (u01 * t[0]) + 1.f;
      p[2] = p[2] * (u23 * t[3]) + 1.f; p[3] = p[3] * (u23 * t[2]) + 1.f;
      asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]));
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int OP>
static void run(const char* name, int threads, float* out, long long* clk, int per_iter) {
  const int iters = 20000;
  hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(threads), 0, 0, out, 100, clk);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(threads), 0, 0, out, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long long c = 0; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  const int wps = threads / 256;
  // clock64 ticks at a fixed 100 MHz on gfx9: use wall time x nominal 2.4 GHz as the cycle estimate as well
  printf("%-34s %d wave/SIMD: %7.3f ms  -> %6.2f ns per SIMD per %s (x2.4 GHz = %5.1f cycles)   s_memtime ticks %lld\n", name, wps, ms,
         ms * 1e6 / ((double)iters * per_iter * wps), per_iter == 8 ? "instruction" : "8-element swish",
         ms * 1e6 / ((double)iters * per_iter * wps) * 2.4, c);
}

int main() {
  float* out; long long* clk;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 8);
  for (int threads : {256, 1024}) {
    if (threads == 256) {
      run<0>("v_fma_f32", 256, out, clk, 8); run<1>("v_pk_fma_f32", 256, out, clk, 8); run<4>("v_pk_mul_f32", 256, out, clk, 8);
      run<2>("v_exp_f32", 256, out, clk, 8); run<3>("v_rcp_f32", 256, out, clk, 8);
      run<5>("swish x8, 8 reciprocals", 256, out, clk, 1); run<6>("swish x8, 2 shared reciprocals", 256, out, clk, 1);
    } else {
      run<0>("v_fma_f32", 1024, out, clk, 8); run<1>("v_pk_fma_f32", 1024, out, clk, 8); run<4>("v_pk_mul_f32", 1024, out, clk, 8);
      run<2>("v_exp_f32", 1024, out, clk, 8); run<3>("v_rcp_f32", 1024, out, clk, 8);
      run<5>("swish x8, 8 reciprocals", 1024, out, clk, 1); run<6>("swish x8, 2 shared reciprocals", 1024, out, clk, 1);
    }
  }
  return 0;
}
