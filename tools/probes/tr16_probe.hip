// Probe of ds_read_b64_tr_b16 lane semantics on gfx950: lds[i] = i, lane i of each 16-lane group
// supplies the address of (row i/4, col group i%4) of a 4x16 block with a given row stride.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(unsigned short* o, int row_stride) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  int l = threadIdx.x, i = l & 15, g = l >> 4;
  __attribute__((address_space(3))) s16x4* p =
      (__attribute__((address_space(3))) s16x4*)(lds + g * 4 * row_stride + (i >> 2) * row_stride + (i & 3) * 4);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int j = 0; j < 4; ++j) o[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  for (int rs : {16, 64}) {
    k<<<1, 64>>>(d, rs); unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("row_stride=%d\n", rs);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", h[l * 4 + j]); printf("%s", (l % 4 == 3) ? "\n" : "   "); }
  }
  return 0;
}
