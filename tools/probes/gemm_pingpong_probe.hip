// Main-loop probe: 256 x 256 bf16 GEMM tile, 8 waves (2 x 4, 128 x 64 per wave), v_mfma_f32_32x32x16_bf16.
//   baseline : the shipping loop of csrc/gemm_stream_kernel.h (2 stages of 64-wide k-tiles, one barrier per k-tile)
//   pingpong : two wave groups (wm = 0 / wm = 1: the two waves of every SIMD) alternate between an MFMA slot (8 MFMAs = one
//              16-wide k-step of the wave's 128 x 64 block) and a LOAD slot (the 6 fragment reads of its next k-step + its
//              share of the LDS-DMA), one s_barrier per slot; four ring stages of 32-wide k-tiles, DMA three stages ahead
// No real epilogue; one tile per workgroup, K = 4096 by default.  Prints us, TFLOP/s and cycles per 64 k (workgroup 0).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/gemm_pingpong_probe.hip -o /tmp/gpp && /tmp/gpp [rounds] [K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned kOob = 0x7fffff00u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ int lds_slot(int row, int chunk) { return row * 8 + (chunk ^ ((row >> 1) & 7)); }

// ------------------------------------------------------------------------------------------------ baseline
__global__ void __launch_bounds__(512) probe_base(const unsigned short* a, const unsigned short* b, float* out, int M, int N, int K,
                                                  long long* clk) {
  constexpr int BM = 256, BN = 256, BK = 64, NW = 8, WAVES_N = 4;
  constexpr int WTM = 128, WTN = 64, TM = 4, TN = 2;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int frow = lane & 31, fhi = lane >> 5;
  const int tiles_n = N / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(a, (unsigned)((size_t)M * K * 2));
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(b, (unsigned)((size_t)N * K * 2));
  const int lrow = lane >> 3, lpc = lane & 7;
  unsigned a_off[A_INSTR], b_off[B_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int r = (wave * A_INSTR + j) * 8 + lrow;
    a_off[j] = (unsigned)(((size_t)(m0 + r) * K + (lpc ^ ((r >> 1) & 7)) * 8) * 2);
  }
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    const int r = (wave * B_INSTR + j) * 8 + lrow;
    b_off[j] = (unsigned)(((size_t)(n0 + r) * K + (lpc ^ ((r >> 1) & 7)) * 8) * 2);
  }
  auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
    char* sa = smem + stage * STAGE;
    char* sb = sa + A_BYTES;
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(sb + (wave * B_INSTR + j) * 1024), 16, (int)b_off[j], kt * 128, 0, 0);
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(sa + (wave * A_INSTR + j) * 1024), 16, (int)a_off[j], kt * 128, 0, 0);
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int nk = K / BK;
  issue(0, 0);
  int cur = 0;
  const long long t0 = clock64();
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(kt + 1 < nk ? kt + 1 : kt, cur ^ 1);
    const uint4* sA = reinterpret_cast<const uint4*>(smem + cur * STAGE);
    const uint4* sB = reinterpret_cast<const uint4*>(smem + cur * STAGE + A_BYTES);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = __builtin_bit_cast(bf16x8, sA[lds_slot(wm * WTM + i * 32 + frow, ks * 2 + fhi)]);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = __builtin_bit_cast(bf16x8, sB[lds_slot(wn * WTN + j * 32 + frow, ks * 2 + fhi)]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    cur ^= 1;
  }
  const long long t1 = clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][j][e] * (float)(1 + ((i * 2 + j) * 16 + e) % 7);
  out[(size_t)blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------ ping-pong
// VAR bit 0: s_setprio 1 around the MFMA slot; bit 1: lgkmcnt(0) AFTER the barrier that ends a LOAD slot instead of before it
template <int VAR>
__global__ void __launch_bounds__(512) probe_pp(const unsigned short* a, const unsigned short* b, float* out, int M, int N, int K,
                                                long long* clk) {
  constexpr int BM = 256, BN = 256, WTM = 128, WTN = 64, TM = 4, TN = 2;
  constexpr int BKP = 32, NS = 4;
  constexpr int A_BYTES = BM * 64, STAGE = (BM + BN) * 64;   // 32 KiB per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int tiles_n = N / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(a, (unsigned)((size_t)M * K * 2));
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(b, (unsigned)((size_t)N * K * 2));
  // DMA piece = 16 rows x 64 B; lane -> (row lane >> 2, physical chunk lane & 3) fetches logical chunk (lane & 3) ^ ((row >> 2) & 3)
  const int drow = lane >> 2;
  const int dchunk = (lane & 3) ^ ((lane >> 4) & 3);
  unsigned a_off[2], b_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 16 + drow;
    a_off[j] = (unsigned)(((size_t)(m0 + r) * K + dchunk * 8) * 2);
    b_off[j] = (unsigned)(((size_t)(n0 + r) * K + dchunk * 8) * 2);
  }
  const int nst = K / BKP;
  auto issue_b = [&](int st) __attribute__((always_inline)) {      // this wave's two weight pieces of stage st
    char* sb = smem + (st & (NS - 1)) * STAGE + A_BYTES;
    const bool ok = st < nst;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(sb + (wave * 2 + j) * 1024), 16, (int)(ok ? b_off[j] : kOob), st * (BKP * 2), 0, 0);
  };
  auto issue_a = [&](int st) __attribute__((always_inline)) {      // ... two activation pieces
    char* sa = smem + (st & (NS - 1)) * STAGE;
    const bool ok = st < nst;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(sa + (wave * 2 + j) * 1024), 16, (int)(ok ? a_off[j] : kOob), st * (BKP * 2), 0, 0);
  };
  const int frow = lane & 31, fhi = lane >> 5, fsw = (frow >> 2) & 3;
  const unsigned fa_base = (unsigned)((wm * WTM + frow) * 64 + ((fhi ^ fsw) * 16));            // ks = 0; ks = 1: ^ 32
  const unsigned fb_base = (unsigned)(A_BYTES + (wn * WTN + frow) * 64 + ((fhi ^ fsw) * 16));
  bf16x8 fa[TM], fb[TN];
  auto load_frags = [&](int st, int ks) __attribute__((always_inline)) {
    const char* sbase = smem + (st & (NS - 1)) * STAGE;
    const unsigned x = ks ? 32u : 0u;
#pragma unroll
    for (int j = 0; j < TN; ++j)
      fb[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sbase + ((fb_base ^ x) + j * 32 * 64)));
#pragma unroll
    for (int i = 0; i < TM; ++i)
      fa[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sbase + ((fa_base ^ x) + i * 32 * 64)));
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  auto mfma_slot = [&]() __attribute__((always_inline)) {
    if (VAR & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (VAR & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    if (VAR & 1) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto end_load_slot = [&]() __attribute__((always_inline)) {
    if (!(VAR & 2)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // prologue: stages 0, 1, 2 in flight; stage 0 landed and published
  issue_b(0); issue_a(0); issue_b(1); issue_a(1); issue_b(2); issue_a(2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const long long t0 = clock64();
  if (wm == 1) {   // the second group runs one slot behind the first
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  for (int st = 0; st < nst; ++st) {
    // LOAD slot, k-step 0 of stage st
    load_frags(st, 0);
    issue_b(st + 3);
    __builtin_amdgcn_sched_barrier(0);
    end_load_slot();
    mfma_slot();
    // LOAD slot, k-step 1
    load_frags(st, 1);
    issue_a(st + 3);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // this wave's pieces of stage st + 1 have landed (st + 2, st + 3 in flight)
    end_load_slot();
    mfma_slot();
  }
  if (wm == 0) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  const long long t1 = clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][j][e] * (float)(1 + ((i * 2 + j) * 16 + e) % 7);
  out[(size_t)blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

typedef void (*kern_t)(const unsigned short*, const unsigned short*, float*, int, int, int, long long*);

static double run(kern_t k, const unsigned short* a, const unsigned short* b, float* out, long long* clk, int M, int N, int K,
                  const char* name, std::vector<float>* keep) {
  const size_t lds = 2 * (256 + 256) * 128;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = (M / 256) * (N / 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, a, b, out, M, N, K, clk);
  hipEventRecord(e0);
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, a, b, out, M, N, K, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  long long c = 0;
  hipMemcpy(&c, clk, sizeof(c), hipMemcpyDeviceToHost);
  std::vector<float> h((size_t)grid * 512);
  hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
  double diff = 0, mx = 0;
  if (keep->empty()) *keep = h;
  for (size_t i = 0; i < h.size(); ++i) { diff = fmax(diff, fabs((double)h[i] - (*keep)[i])); mx = fmax(mx, fabs((double)(*keep)[i])); }
  const double us = ms * 1e3 / reps;
  printf("%-34s grid %4d  %8.1f us  %7.1f TFLOP/s  %6.0f cycles per 64 k (workgroup 0)  max|diff vs first| %.3g (of %.3g)  %s\n", name,
         grid, us, 2.0 * M * N * K / us * 1e-6, (double)c / (K / 64), diff, mx, hipGetErrorString(hipGetLastError()));
  return us;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 1;            // tiles per CU
  const int K = argc > 2 ? atoi(argv[2]) : 4096;
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int M = 256 * 16 * rounds, N = 256 * (cus / 16);
  std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K);
  unsigned h = 12345u;
  auto rnd = [&]() { h = h * 1664525u + 1013904223u; float f = ((int)(h >> 9) % 2001 - 1000) * 1e-3f; unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
  for (auto& v : ha) v = rnd();
  for (auto& v : hb) v = rnd();
  unsigned short *a, *b; float* out; long long* clk;
  hipMalloc(&a, ha.size() * 2); hipMalloc(&b, hb.size() * 2); hipMalloc(&out, (size_t)(M / 256) * (N / 256) * 512 * 4); hipMalloc(&clk, 8);
  hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
  printf("M=%d N=%d K=%d  (%d CUs)\n", M, N, K, cus);
  std::vector<float> keep;
  run(probe_base, a, b, out, clk, M, N, K, "baseline (stream loop)", &keep);
  run(probe_pp<0>, a, b, out, clk, M, N, K, "pingpong", &keep);
  run(probe_pp<1>, a, b, out, clk, M, N, K, "pingpong + setprio", &keep);
  run(probe_pp<2>, a, b, out, clk, M, N, K, "pingpong, lgkm after barrier", &keep);
  run(probe_pp<3>, a, b, out, clk, M, N, K, "pingpong + setprio, lgkm after", &keep);
  run(probe_base, a, b, out, clk, M, N, K, "baseline again", &keep);
  return 0;
}
