// Synthetic neighbours for tools/tha_coresident_probe.py (profiles/NOTES_r04.md section 1): workgroups of four waves with
// 80 KiB of LDS -- the footprint of the one GEMM tile that disturbs the H = 4 talking-heads kernel -- that do ONE class of
// work each, so that the disturbing instruction class can be named without touching the GEMM kernel.
//   mode bit 0: LDS-DMA (buffer_load_dwordx4 ... lds) into the whole allocation
//        bit 1: buffer loads into VGPRs
//        bit 2: ds_write_b128 / ds_read_b128 over the whole allocation
//        bit 3: MFMA on AccVGPR accumulators (the kernel holds 64 of them either way)
//        bit 4: buffer stores
//        bit 5: LDS-DMA whose lanes are ALL out of range of the descriptor (zeros arrive; what the GEMM's last prefetch does)
//        bit 6: LDS-DMA with the odd lanes out of range
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC neighbour_kernels.hip -o bin/libneighbour.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

extern "C" __global__ void __launch_bounds__(256) neighbour_kernel(const char* src, char* dst, unsigned bytes, int mode, int iters,
                                                                    int lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t rs = rsrc_of(src, bytes), rd = rsrc_of(dst, bytes);
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  u32x4 keep = {0u, 0u, 0u, 0u};
  const int pieces = lds_bytes / 1024;          // 1 KiB per wave instruction
  unsigned off = (unsigned)((blockIdx.x * 256 + tid) * 16) % (bytes - 65536);
  for (int it = 0; it < iters; ++it) {
    if (mode & 1) {
      for (int pc = wave; pc < pieces; pc += 4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + pc * 1024), 16, (int)(off + lane * 0), (pc & 15) * 4096, 0, 0);
    }
    if (mode & 96) {
      for (int pc = wave; pc < pieces; pc += 4) {
        const unsigned o = ((mode & 32) || (lane & 1)) ? 0x80000000u : off;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + pc * 1024), 16, (int)o, 0, 0, 0);
      }
    }
    if (mode & 2) {
      for (int j = 0; j < 8; ++j) {
        const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, j * 4096, 0));
        keep ^= v;
      }
    }
    if (mode & 4) {
      for (int pc = wave; pc < pieces; pc += 4) {
        u32x4* q = reinterpret_cast<u32x4*>(smem + pc * 1024 + lane * 16);
        *q = keep + (unsigned)it;
      }
      for (int pc = wave; pc < pieces; pc += 4) keep ^= *reinterpret_cast<const u32x4*>(smem + pc * 1024 + (lane ^ 1) * 16);
    }
    if (mode & 8) {
      const bf16x8 a = __builtin_bit_cast(bf16x8, keep), b = __builtin_bit_cast(bf16x8, keep + 1u);
      for (int r = 0; r < 8; ++r)
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    if (mode & 16) {
      for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(keep, rd, (int)off, j * 4096, 0);
    }
    if (mode & 97) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    off = (off + 1048576u) % (bytes - 65536);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 123.456f || keep.x == 0x12345u) dst[tid] = (char)keep.y;   // keep everything alive
}

// The victim of the hypothesis (NOTES_r04 section 1): a ds_bpermute_b32 that is still in flight when the wave narrows EXEC.
// Every lane supplies lane + 1; lanes 0..15 read lane ^ 32 (so 33..48 must arrive).  Variant 0: s_and_saveexec right behind
// the bpermute, the wait inside the narrowed region (what hipcc generated in tha_kernel: the add that consumes the shuffle was
// sunk into the `if (g == 0)` block).  Variant 1: the wait in front of the EXEC write.  counts[0] += lanes that received
// something else than lane ^ 32 + 1, counts[1] += those that received exactly 0.
extern "C" __global__ void __launch_bounds__(256) bperm_victim_kernel(unsigned* counts, int iters, int variant) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const unsigned addr = (unsigned)((lane ^ 32) * 4);
  const unsigned val = (unsigned)lane + 1u;
  const unsigned long long g0 = 0xffffull;
  unsigned bad = 0, zero = 0;
  // some LDS traffic of its own, so that the allocation is real
  reinterpret_cast<unsigned*>(smem)[threadIdx.x] = val;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    unsigned r = 0xffffffffu;
    unsigned long long saved;
    if (variant == 0) {
      asm volatile("ds_bpermute_b32 %0, %2, %3\n\t"
                   "s_and_saveexec_b64 %1, %4\n\t"
                   "s_waitcnt lgkmcnt(0)\n\t"
                   "s_nop 4\n\t"
                   "s_mov_b64 exec, %1"
                   : "+v"(r), "=&s"(saved) : "v"(addr), "v"(val), "s"(g0) : "memory");
    } else {
      asm volatile("ds_bpermute_b32 %0, %2, %3\n\t"
                   "s_waitcnt lgkmcnt(0)\n\t"
                   "s_and_saveexec_b64 %1, %4\n\t"
                   "s_nop 4\n\t"
                   "s_mov_b64 exec, %1"
                   : "+v"(r), "=&s"(saved) : "v"(addr), "v"(val), "s"(g0) : "memory");
    }
    if (lane < 16) {
      if (r != (unsigned)(lane ^ 32) + 1u) ++bad;
      if (r == 0u) ++zero;
    }
  }
  if (bad) atomicAdd(&counts[0], bad);
  if (zero) atomicAdd(&counts[1], zero);
}

extern "C" int bperm_victim_launch(unsigned* counts, int iters, int variant, int grid, int lds_bytes, void* stream) {
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute((const void*)bperm_victim_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
    done = true;
  }
  hipLaunchKernelGGL(bperm_victim_kernel, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, counts, iters, variant);
  return (int)hipGetLastError();
}

// Second hypothesis (what the per-lane dump of the probe build showed: only lanes 48..63, only the odd mixed heads): an LDS
// read of two dwords whose result is consumed right behind `s_waitcnt lgkmcnt(0)` -- is the HIGH dword of the last lanes
// really there?  Destination registers are poisoned, every lane reads (lane + 1, 1000 + lane) from LDS, waits, and consumes
// the pair NOPS wait states later:
//   form 0  v_pk_add_f32 (both dwords, no operand select)   form 1  v_pk_fma_f32 ... op_sel:[0,1,0]      (broadcast of the high dword,
//   form 2  v_pk_fma_f32 ... op_sel_hi:[1,0,0] (low dword) what hipcc emits for `w[h][hp] * acc` in tha_kernel)
// READ 0 ds_read2_b32, 1 ds_read_b64.  counts[q] += wrong lanes of lane quarter q (4 + q: the wrong value was the poison).
template <int FORM, int NOPS, int READ>
__global__ void __launch_bounds__(256) ldsret_victim_kernel(unsigned* counts, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((ext_vector_type(2))) float f2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* tab = reinterpret_cast<float*>(smem) + wave * 128;
  tab[lane * 2] = (float)(lane + 1);
  tab[lane * 2 + 1] = (float)(1000 + lane);
  __syncthreads();
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(tab + lane * 2);
  const f2 ones = {1.f, 1.f}, zeros = {0.f, 0.f};
  unsigned bad = 0, poison = 0;
  for (int it = 0; it < iters; ++it) {
    f2 r = {-7.f, -7.f}, d = {0.f, 0.f};
    asm volatile("" : "+v"(r));
#define LDSRET_BODY(RD, USE)                                                                         \
    if (NOPS == 0) asm volatile(RD "\n\ts_waitcnt lgkmcnt(0)\n\t" USE : "+v"(r), "=&v"(d) : "v"(addr), "v"(ones), "v"(zeros) : "memory");             \
    if (NOPS == 1) asm volatile(RD "\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 0\n\t" USE : "+v"(r), "=&v"(d) : "v"(addr), "v"(ones), "v"(zeros) : "memory");   \
    if (NOPS == 2) asm volatile(RD "\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 1\n\t" USE : "+v"(r), "=&v"(d) : "v"(addr), "v"(ones), "v"(zeros) : "memory");   \
    if (NOPS == 4) asm volatile(RD "\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 3\n\t" USE : "+v"(r), "=&v"(d) : "v"(addr), "v"(ones), "v"(zeros) : "memory");   \
    if (NOPS == 8) asm volatile(RD "\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 7\n\t" USE : "+v"(r), "=&v"(d) : "v"(addr), "v"(ones), "v"(zeros) : "memory");
#define LDSRET_READ(USE)                                                   \
    if (READ == 0) { LDSRET_BODY("ds_read2_b32 %0, %2 offset1:1", USE) }   \
    else { LDSRET_BODY("ds_read_b64 %0, %2", USE) }
    float e0, e1;
    if (FORM == 0) {
      LDSRET_READ("v_pk_add_f32 %1, %0, %4")
      e0 = (float)(lane + 1); e1 = (float)(1000 + lane);
    } else if (FORM == 1) {
      LDSRET_READ("v_pk_fma_f32 %1, %3, %0, %4 op_sel:[0,1,0]")
      e0 = e1 = (float)(1000 + lane);
    } else if (FORM == 2) {
      LDSRET_READ("v_pk_fma_f32 %1, %3, %0, %4 op_sel_hi:[1,0,0]")
      e0 = e1 = (float)(lane + 1);
    } else if (FORM == 4) {
      LDSRET_READ("v_pk_mul_f32 %1, %0, %3 op_sel:[1,0]")
      e0 = e1 = (float)(1000 + lane);
    } else if (FORM == 5) {
      LDSRET_READ("v_pk_fma_f32 %1, %0, %3, %4 op_sel:[1,0,0]")
      e0 = e1 = (float)(1000 + lane);
    } else if (FORM == 6) {
      LDSRET_READ("v_pk_fma_f32 %1, %4, %4, %0 op_sel:[0,0,1]")
      e0 = e1 = (float)(1000 + lane);
    } else if (FORM == 8) {
      LDSRET_READ("v_pk_mov_b32 %1, %0, %0 op_sel:[1,0]")        // D.lo = src0.hi, D.hi = src1.lo
      e0 = (float)(1000 + lane); e1 = (float)(lane + 1);
    } else if (FORM == 9) {
      LDSRET_READ("v_pk_mul_f32 %1, %3, %0 op_sel:[0,1]")          // src1 high -> low
      e0 = e1 = (float)(1000 + lane);
    } else if (FORM == 10) {
      LDSRET_READ("v_pk_add_f32 %1, %4, %0 op_sel:[0,1]")          // src1 high -> low
      e0 = e1 = (float)(1000 + lane);
    } else if (FORM == 7) {
      // no LDS at all: the pair comes out of VALU instructions
      r = f2{(float)(lane + 1), (float)(1000 + lane)};
      asm volatile("" : "+v"(r));
      asm volatile("v_pk_fma_f32 %1, %2, %0, %3 op_sel:[0,1,0]" : "+v"(r), "=&v"(d) : "v"(ones), "v"(zeros));
      e0 = e1 = (float)(1000 + lane);
    } else {
      // form 3: the ADDRESS register of the read is overwritten right behind its issue (hipcc does that in tha_kernel:
      // `ds_read2_b32 v[52:53], v13 offset1:1` / `v_mov_b32 v13, v12`); NOPS counts the wait states in front of the overwrite
      unsigned a2 = addr;
      asm volatile("" : "+v"(a2));
      if (NOPS == 0) asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tv_mov_b32 %2, 0x7ff0\n\ts_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 %1, %0, %3" : "+v"(r), "=&v"(d), "+v"(a2) : "v"(zeros) : "memory");
      else if (NOPS == 1) asm volatile("ds_read2_b32 %0, %2 offset1:1\n\ts_nop 0\n\tv_mov_b32 %2, 0x7ff0\n\ts_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 %1, %0, %3" : "+v"(r), "=&v"(d), "+v"(a2) : "v"(zeros) : "memory");
      else if (NOPS == 2) asm volatile("ds_read2_b32 %0, %2 offset1:1\n\ts_nop 1\n\tv_mov_b32 %2, 0x7ff0\n\ts_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 %1, %0, %3" : "+v"(r), "=&v"(d), "+v"(a2) : "v"(zeros) : "memory");
      else if (NOPS == 4) asm volatile("ds_read2_b32 %0, %2 offset1:1\n\ts_nop 3\n\tv_mov_b32 %2, 0x7ff0\n\ts_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 %1, %0, %3" : "+v"(r), "=&v"(d), "+v"(a2) : "v"(zeros) : "memory");
      else asm volatile("ds_read2_b32 %0, %2 offset1:1\n\ts_nop 7\n\tv_mov_b32 %2, 0x7ff0\n\ts_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 %1, %0, %3" : "+v"(r), "=&v"(d), "+v"(a2) : "v"(zeros) : "memory");
      e0 = (float)(lane + 1); e1 = (float)(1000 + lane);
    }
    if (d[0] != e0 || d[1] != e1) {
      if (!bad) { counts[8] = __float_as_uint(d[0]); counts[9] = __float_as_uint(d[1]); counts[10] = (unsigned)lane; }   // one sample
      ++bad;
      if (d[0] == -7.f || d[1] == -7.f) ++poison;
    }
  }
  if (bad) atomicAdd(&counts[lane >> 4], bad);
  if (poison) atomicAdd(&counts[4 + (lane >> 4)], poison);
}

extern "C" int ldsret_victim_launch(unsigned* counts, int iters, int form, int nops, int read, int grid, int lds_bytes, void* stream) {
  void (*fn)(unsigned*, int) = nullptr;
#define PICK(F, N, R) if (form == F && nops == N && read == R) fn = ldsret_victim_kernel<F, N, R>;
#define PICKN(F, R) PICK(F, 0, R) PICK(F, 1, R) PICK(F, 2, R) PICK(F, 4, R) PICK(F, 8, R)
  PICKN(0, 0) PICKN(1, 0) PICKN(2, 0) PICKN(3, 0) PICKN(4, 0) PICKN(5, 0) PICKN(6, 0) PICKN(7, 0) PICKN(8, 0) PICKN(9, 0) PICKN(10, 0) PICKN(0, 1) PICKN(1, 1) PICKN(2, 1)
  if (!fn) return -2;
  if (hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, counts, iters);
  return (int)hipGetLastError();
}

extern "C" int neighbour_launch(const void* src, void* dst, unsigned bytes, int mode, int iters, int grid, int lds_bytes, void* stream) {
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute((const void*)neighbour_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
    done = true;
  }
  hipLaunchKernelGGL(neighbour_kernel, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, (const char*)src, (char*)dst, bytes, mode,
                     iters, lds_bytes);
  return (int)hipGetLastError();
}
