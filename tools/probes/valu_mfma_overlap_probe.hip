// Do the VALU instructions of one wave overlap with the MFMAs of ANOTHER wave of the same SIMD (gfx950)?
// One workgroup of 512 threads per CU (two waves per SIMD).  The waves of `valu_mask` run a loop of 8 independent (or 4)
// v_pk_fma_f32 chains, the waves of `mfma_mask` a loop of independent v_mfma_f32_32x32x16_bf16 (4 accumulators) with the
// accumulators in ArchVGPRs ("v") or AccVGPRs ("a"); the others exit.  Prints cycles per wave instruction for each role,
// alone and together.   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_mfma_overlap_probe.hip -o /tmp/vmo && /tmp/vmo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int ACC_A, int CHAINS>
__global__ void __launch_bounds__(512) probe(float* out, int iters, unsigned valu_mask, unsigned mfma_mask, long long* clk) {
  const int wave = threadIdx.x >> 6;
  const bool do_valu = (valu_mask >> wave) & 1, do_mfma = (mfma_mask >> wave) & 1;
  if (!do_valu && !do_mfma) return;
  __syncthreads();   // (the exited waves do not count)
  if (do_valu) {
    f32x2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f32x2{0.5f + 0.001f * (threadIdx.x + i), 0.75f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8 / CHAINS; ++r)
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) clk[wave] = t1 - t0;
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * threadIdx.x); b[e] = (__bf16)0.5f; }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
      if (ACC_A)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\t"
                     "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %5, %3"
                     : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]) : "v"(a), "v"(b));
      else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\t"
                     "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %5, %3"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(a), "v"(b));
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int bq = 0; bq < 4; ++bq)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[bq][e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) clk[wave] = t1 - t0;
  }
}

template <int ACC_A, int CHAINS>
void run(const char* what, unsigned vm, unsigned mm, float* out, long long* clk) {
  const int iters = 4096;
  hipMemset(clk, 0, 8 * 8);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<ACC_A, CHAINS>), dim3(256), dim3(512), 0, 0, out, iters, vm, mm, clk);
  hipDeviceSynchronize();
  long long h[8];
  hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
  printf("%-64s", what);
  for (int w = 0; w < 8; ++w) {
    if ((vm >> w) & 1) printf("  w%d valu %5.2f", w, (double)h[w] / (iters * 8.0));
    else if ((mm >> w) & 1) printf("  w%d mfma %5.1f", w, (double)h[w] / (iters * 4.0));
  }
  printf("\n");
}

int main() {
  float* out; long long* clk;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 64);
  printf("cycles per wave instruction (v_pk_fma_f32 / v_mfma_f32_32x32x16_bf16)\n");
  run<0, 8>("valu alone: wave 0, 8 chains", 0x01, 0x00, out, clk);
  run<0, 4>("valu alone: wave 0, 4 chains", 0x01, 0x00, out, clk);
  run<0, 8>("valu: waves 0 and 4 (same SIMD?), 8 chains", 0x11, 0x00, out, clk);
  run<0, 8>("valu: waves 0 and 1 (different SIMDs?), 8 chains", 0x03, 0x00, out, clk);
  run<0, 8>("mfma alone: wave 4, acc in v", 0x00, 0x10, out, clk);
  run<1, 8>("mfma alone: wave 4, acc in a", 0x00, 0x10, out, clk);
  run<0, 8>("mfma: waves 0 and 4, acc in v", 0x00, 0x11, out, clk);
  run<0, 8>("valu wave 0 (8 chains) + mfma wave 4, acc in v", 0x01, 0x10, out, clk);
  run<1, 8>("valu wave 0 (8 chains) + mfma wave 4, acc in a", 0x01, 0x10, out, clk);
  run<0, 4>("valu wave 0 (4 chains) + mfma wave 4, acc in v", 0x01, 0x10, out, clk);
  run<1, 4>("valu wave 0 (4 chains) + mfma wave 4, acc in a", 0x01, 0x10, out, clk);
  run<0, 8>("valu wave 0 (8 chains) + mfma wave 1 (other SIMD?), acc in v", 0x01, 0x02, out, clk);
  run<0, 8>("valu waves 0-3 + mfma waves 4-7, acc in v", 0x0f, 0xf0, out, clk);
  run<1, 8>("valu waves 0-3 + mfma waves 4-7, acc in a", 0x0f, 0xf0, out, clk);
  return 0;
}
