// Main-loop probe for the 256 x 256 bf16 GEMM tile: the same LDS-DMA ring (1-KiB pieces, source-side XOR swizzle, one barrier
// per 64-wide k-tile) as csrc/gemm_stream_kernel.h, no real epilogue, with the tile split over
//   8 waves (2 x 4: 128 x 64 per wave, 128 accumulator registers, two waves per SIMD)   -- what ships
//   4 waves (2 x 2: 128 x 128 per wave, 256 accumulator registers, one wave per SIMD)   -- a third fewer LDS fragment bytes per MFMA
// One workgroup per CU-sized tile grid, K = 4096: prints microseconds, TFLOP/s and shader cycles per k-tile.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/gemm_wave_tile_probe.hip -o /tmp/gwt && /tmp/gwt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr unsigned kOob = 0x7fffff00u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ int lds_slot(int row, int chunk) { return row * 8 + (chunk ^ ((row >> 1) & 7)); }

// Hand-pipelined variant (round 4): every fragment read is an explicit ds_read_b128 into one of two register sets, requested one
// k-slice ahead of the MFMAs that consume it with counted lgkmcnt waits; the ONE barrier per k-tile sits in front of the last
// slice's MFMAs (its fragments have arrived, so every wave is done reading the stage), the refill of that stage and the first
// reads of the next stage are issued right behind it and run under those MFMAs -- no exposed LDS latency at the k-tile boundary.
template <int WAVES_M, int WAVES_N>
__global__ void __launch_bounds__(WAVES_M* WAVES_N * 64) probe_pipe(const unsigned short* a, const unsigned short* b, float* out, int M,
                                                                  int N, int K, long long* clk) {
  constexpr int BM = 256, BN = 256, BK = 64, NW = WAVES_M * WAVES_N;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N, TM = WTM / 32, TN = WTN / 32;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int NR = TM + TN;                       // fragment reads per k-slice
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
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
  // LDS byte offsets of this lane's fragment rows (k-slice 0) and the XOR term of the chunk swizzle
  const unsigned base = (unsigned)(size_t)(lds_ptr_t)smem;
  unsigned row_off[NR], sw[NR];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wm * WTM + i * 32 + frow;
    row_off[i] = (unsigned)(row * 128); sw[i] = (unsigned)((row >> 1) & 7);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * WTN + j * 32 + frow;
    row_off[TM + j] = (unsigned)(A_BYTES + row * 128); sw[TM + j] = (unsigned)((row >> 1) & 7);
  }
  u32x4 fr[2][NR];
  auto reads = [&](int stage, int ks, int buf) __attribute__((always_inline)) {
    const unsigned st = base + (unsigned)(stage * STAGE);
    const unsigned c = (unsigned)(ks * 2 + fhi);
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const unsigned ad = st + row_off[q] + ((c ^ sw[q]) << 4);
      asm volatile("ds_read_b128 %0, %1" : "=v"(fr[buf][q]) : "v"(ad) : "memory");
    }
  };
  auto wait_set = [&](int buf, bool more) __attribute__((always_inline)) {
    // the older NR reads have landed; `more`: the NR reads of the next slice stay in flight
    if (NR == 8) {
      if (more) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fr[buf][0]), "+v"(fr[buf][1]), "+v"(fr[buf][2]), "+v"(fr[buf][3]), "+v"(fr[buf][4]), "+v"(fr[buf][5]), "+v"(fr[buf][NR - 2]), "+v"(fr[buf][NR - 1]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[buf][0]), "+v"(fr[buf][1]), "+v"(fr[buf][2]), "+v"(fr[buf][3]), "+v"(fr[buf][4]), "+v"(fr[buf][5]), "+v"(fr[buf][NR - 2]), "+v"(fr[buf][NR - 1]));
    } else {
      if (more) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fr[buf][0]), "+v"(fr[buf][1]), "+v"(fr[buf][2]), "+v"(fr[buf][3]), "+v"(fr[buf][NR - 2]), "+v"(fr[buf][NR - 1]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[buf][0]), "+v"(fr[buf][1]), "+v"(fr[buf][2]), "+v"(fr[buf][3]), "+v"(fr[buf][NR - 2]), "+v"(fr[buf][NR - 1]));
    }
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  auto mfmas = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[buf][TM + j]), __builtin_bit_cast(bf16x8, fr[buf][i]),
                                                            acc[i][j], 0, 0, 0);
  };
  const int nk = K / BK;
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  issue(nk > 1 ? 1 : 0, 1);
  reads(0, 0, 0);
  int cur = 0;
  const long long t0 = clock64();
  for (int kt = 0; kt < nk; ++kt) {
    // slices 0..2: request the next slice, wait for this one, multiply
    reads(cur, 1, 1); wait_set(0, true); mfmas(0);
    reads(cur, 2, 0); wait_set(1, true); mfmas(1);
    reads(cur, 3, 1); wait_set(0, true); mfmas(0);
    // last slice: its fragments have arrived -> every wave is done with this stage once it passes the barrier
    wait_set(1, false);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the next stage's DMA (issued a whole k-tile ago) has landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(kt + 2 < nk ? kt + 2 : kt, cur);                      // refill the stage just finished
    reads(cur ^ 1, 0, 0);                                       // first slice of the next k-tile, under the MFMAs below
    mfmas(1);
    cur ^= 1;
  }
  const long long t1 = clock64();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[(size_t)blockIdx.x * blockDim.x + tid] = s + __uint_as_float(fr[0][0][0] & 0u);
  if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int WAVES_M, int WAVES_N>
__global__ void __launch_bounds__(WAVES_M* WAVES_N * 64) probe(const unsigned short* a, const unsigned short* b, float* out, int M,
                                                             int N, int K, long long* clk) {
  constexpr int BM = 256, BN = 256, BK = 64, NW = WAVES_M * WAVES_N;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N, TM = WTM / 32, TN = WTN / 32;
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
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[(size_t)blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int WM, int WN, bool PIPE = false>
static void run(const unsigned short* a, const unsigned short* b, float* out, long long* clk, int M, int N, int K, const char* name) {
  const size_t lds = 2 * (256 + 256) * 128;
  auto fn = PIPE ? probe_pipe<WM, WN> : probe<WM, WN>;
  hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = (M / 256) * (N / 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(fn, dim3(grid), dim3(WM * WN * 64), lds, 0, a, b, out, M, N, K, clk);
  hipEventRecord(e0);
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(fn, dim3(grid), dim3(WM * WN * 64), lds, 0, a, b, out, M, N, K, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  long long c = 0;
  hipMemcpy(&c, clk, sizeof(c), hipMemcpyDeviceToHost);
  const double us = ms * 1e3 / reps;
  printf("%-28s grid %4d  %8.1f us  %7.1f TFLOP/s  %6.0f cycles per k-tile (workgroup 0)  err=%s\n", name, grid, us,
         2.0 * M * N * K / us * 1e-6, (double)c / (K / 64), hipGetErrorString(hipGetLastError()));
  // PROBE_SUSTAIN_MS: the same launch back to back for that long, bracketed by CLOCK_MONOTONIC stamps -- tools/power_probe.py
  // lines its clock / power samples (Python's perf_counter is the same clock) up with the window
  if (const char* sm = getenv("PROBE_SUSTAIN_MS")) {
    const double want = atof(sm) * 1e-3;
    auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
    hipDeviceSynchronize();
    const double t0 = now();
    long n = 0;
    while (now() - t0 < want) {
      for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(fn, dim3(grid), dim3(WM * WN * 64), lds, 0, a, b, out, M, N, K, clk);
      hipDeviceSynchronize();
      n += 20;
    }
    const double t1 = now();
    printf("SUSTAIN|%s|%.6f|%.6f|%.1f\n", name, t0, t1, 2.0 * M * N * K * n / (t1 - t0) * 1e-12);
    fflush(stdout);
  }
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 1;            // tiles per CU
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int M = 256 * 16 * rounds, N = 256 * (cus / 16), K = 4096;
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
  run<2, 4>(a, b, out, clk, M, N, K, "8 waves (128 x 64 per wave)");
  run<2, 2>(a, b, out, clk, M, N, K, "4 waves (128 x 128 per wave)");
  run<2, 4, true>(a, b, out, clk, M, N, K, "8 waves, hand-pipelined");
  run<2, 2, true>(a, b, out, clk, M, N, K, "4 waves, hand-pipelined");
  run<2, 4>(a, b, out, clk, M, N, K, "8 waves again");
  run<2, 2, true>(a, b, out, clk, M, N, K, "4 waves, hand-pipelined again");
  return 0;
}
