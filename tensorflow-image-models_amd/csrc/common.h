// Shared device helpers for the tfimm_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/tfimm_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef uint16_t bf16_t;

// Measurement / ablation switches inside kernels (TFIMM_GEMM_DBG, TFIMM_CHAIN_DBG, TFIMM_ATTN_DBG, ...: skip a phase, redirect an
// operand, record s_memtime stamps) exist only in probe builds (-DTFIMM_PROBE_HOOKS, tools/probes/build_dbg_libs.sh): in the
// product library the flag word reads as 0 at compile time and the code behind it is gone.
#if defined(TFIMM_PROBE_HOOKS) || defined(TFIMM_STREAM_DBG)
#define TFIMM_PROBE(x) (x)
#else
#define TFIMM_PROBE(x) 0
#endif

// ---- error plumbing (host) ---------------------------------------------------------
void tfimm_set_error(const char* fmt, ...);
#define TFIMM_FAIL(code, ...)        \
  do {                               \
    tfimm_set_error(__VA_ARGS__);    \
    return (code);                   \
  } while (0)
#define TFIMM_HIP_CHECK(expr)                                                      \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      tfimm_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                      __FILE__, __LINE__);                                         \
      return (int)_e;                                                              \
    }                                                                              \
  } while (0)
#define TFIMM_LAUNCH_CHECK() TFIMM_HIP_CHECK(hipGetLastError())
// hipGetLastError() is sticky per thread: an unrelated earlier failure (e.g. a device probe made
// by another library before the runtime was initialised) would otherwise be blamed on our launch.
#define TFIMM_LAUNCH(...)                 \
  do {                                    \
    (void)hipGetLastError();              \
    hipLaunchKernelGGL(__VA_ARGS__);      \
    TFIMM_LAUNCH_CHECK();                 \
  } while (0)

// ---- one-time setup per (call site, device) ---------------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a property of a kernel ON ONE DEVICE: a host that drives several devices from one
// process has to set it on each of them, and two host threads may arrive at the same call site at once.  A bit per device
// ordinal (mod 64), set after the setup succeeded; a second thread that runs the setup concurrently repeats an idempotent call.
#include <atomic>
struct tfimm_once_t {
  std::atomic<unsigned long long> done{0};
  static unsigned long long bit() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return 1ull << (dev & 63);
  }
  bool need() const { return !(done.load(std::memory_order_acquire) & bit()); }
  void mark() { done.fetch_or(bit(), std::memory_order_release); }
};

// ---- a workgroup barrier behind which OTHER waves overwrite LDS that THIS wave has been reading ----------------------
// (an epilogue staging block that aliases the operand stage consumed last).  The wave's own LDS reads must be COMPLETE when it
// signals arrival.  __builtin_amdgcn_s_barrier() does not order them: it is not a memory operation for hipcc, which issues
// the last k-slice's ds_read_b128 in front of it and waits for them (s_waitcnt lgkmcnt) where their values are consumed --
// the scheduler is free to put those MFMAs, and the wait, BEHIND the barrier.  Another wave's staging writes then race the
// reads still in flight: one fragment row of a tile multiplied with staged output values (found in round 4 on the K = 64 cases
// of the four-wave tiles, 1-7 of 8 launches, once the s_memtime stamp hook between loop and epilogue -- a branch, i.e. a
// scheduling boundary -- was compiled out of product builds: profiles/NOTES_r04.md section 2).
__device__ __forceinline__ void tfimm_lds_reuse_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ---- bf16 <-> fp32 -----------------------------------------------------------------
__device__ __forceinline__ float bf2f(uint32_t h) { return __uint_as_float(h << 16); }
// fp32 -> bf16, round-to-nearest-even (hardware v_cvt_pk_bf16_f32 on gfx950; NaN stays NaN)
typedef __attribute__((ext_vector_type(2))) float tfimm_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 tfimm_bf16x2;
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  const tfimm_f32x2 f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, tfimm_bf16x2));
}
__device__ __forceinline__ uint32_t f2bf(float f) { return pack_bf2(f, 0.f) & 0xffffu; }
__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
  f[0] = bf2f(u.x & 0xffffu); f[1] = bf2f(u.x >> 16);
  f[2] = bf2f(u.y & 0xffffu); f[3] = bf2f(u.y >> 16);
  f[4] = bf2f(u.z & 0xffffu); f[5] = bf2f(u.z >> 16);
  f[6] = bf2f(u.w & 0xffffu); f[7] = bf2f(u.w >> 16);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 u;
  u.x = pack_bf2(f[0], f[1]); u.y = pack_bf2(f[2], f[3]);
  u.z = pack_bf2(f[4], f[5]); u.w = pack_bf2(f[6], f[7]);
  return u;
}

// ---- squeeze sums (SE): per-(image, channel) sums of a layer's output, accumulated by many workgroups.  Float atomics add in
// whatever order the workgroups arrive, so the last bits -- and through bf16 roundings further down, the logits -- changed from
// launch to launch.  The accumulators are therefore 64-bit FIXED POINT (2^-20 units: +-8.8e12 of range): every thread
// converts the partial sum it formed in a fixed order once, and integer adds commute -- results are bit-reproducible.
typedef long long tfimm_sq_t;
#define TFIMM_SQ_SCALE 1048576.0f
__device__ __forceinline__ tfimm_sq_t sq_from_float(float v) { return __float2ll_rn(v * TFIMM_SQ_SCALE); }
__device__ __forceinline__ void sq_add(tfimm_sq_t* p, tfimm_sq_t q) {
  atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)q);
}
__device__ __forceinline__ float sq_to_float(tfimm_sq_t q) { return (float)((double)q * (1.0 / 1048576.0)); }

// ---- activations (reference tfimm/layers/factory.py:6-13) -----------------------------
__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case TFIMM_ACT_RELU: return fmaxf(v, 0.f);
    case TFIMM_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    case TFIMM_ACT_SWISH: return v / (1.f + __expf(-v));
    case TFIMM_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    case TFIMM_ACT_RELU6: return fminf(fmaxf(v, 0.f), 6.f);
    case TFIMM_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// ---- branch-light activation with wave-uniform parameters (GEMM epilogue, depthwise conv)
// Activation, branch-light (reference act_layer_factory, layers/factory.py:6-13):
//   clamp class   none / relu / relu6 :   v = min(max(v, lo), hi)                 (always executed)
//   sigmoid class swish / sigmoid / tanh: s = 1/(1+exp(-k v)); v = a v s + b s + c
//   gelu (exact erf form; Phi(v) - 1/2 by a degree-9 polynomial of the clamped argument, |GELU error| <= 1.6e-5: see gelu_erf)
struct ActParams {
  float lo, hi, k, a, b, c;
  int cls;
  int clamp;   // lo / hi are finite bounds (relu, relu6)
};
__device__ __forceinline__ ActParams make_act(int act) {
  ActParams q;
  q.lo = -__builtin_inff(); q.hi = __builtin_inff(); q.k = 1.f; q.a = 0.f; q.b = 0.f; q.c = 0.f; q.cls = 0; q.clamp = 0;
  switch (act) {
    case TFIMM_ACT_RELU: q.lo = 0.f; q.clamp = 1; break;
    case TFIMM_ACT_RELU6: q.lo = 0.f; q.hi = 6.f; q.clamp = 1; break;
    case TFIMM_ACT_SWISH: q.cls = 1; q.a = 1.f; break;
    case TFIMM_ACT_SIGMOID: q.cls = 1; q.b = 1.f; break;
    case TFIMM_ACT_TANH: q.cls = 1; q.k = 2.f; q.b = 2.f; q.c = -1.f; break;
    case TFIMM_ACT_GELU: q.cls = 2; break;
    default: break;
  }
  return q;
}
// exact (erf) GELU, v Phi(v) = 0.5 v (1 + erf(v / sqrt 2)), with NO transcendental instruction:
//     Phi(v) - 1/2 = t P(t^2 - c^2 / 2)      on the clamped argument t = med3(v, -c, c), c = 4.5,
// P a degree-9 polynomial (minimax fit of the GELU error itself, tools/fit_gelu_poly.py; Horner in r = t^2 - c^2 / 2, one
// FMA away from t, centred so that the fp32 Horner does not cancel), pinned to Phi(c) = 1 so that beyond the clamp the result
// is v (or 0) up to c (1 - Phi(c)) = 1.5e-5.  max |GELU error| 1.6e-5 over all v, relative error 1.2e-5 around 0 -- two
// orders below the bf16 resolution of the values it produces (tests/test_gelu_poly.py evaluates these very coefficients).
// 12 packable FMAs / multiplies + the clamp per value; the previous form (erf(z) = z g(z^2), degree 10, 16 operations) was
// 20 % of the VALU time of every GELU epilogue, and the Abramowitz-Stegun form before it spent 2 quarter-rate
// transcendentals (v_rcp, v_exp) per element and made the K = 768 fc1 layers of ViT-B VALU-bound in their epilogue.
#define TFIMM_GELU_CLAMP 4.5f
#define TFIMM_GELU_CENTRE 10.125f   /* c^2 / 2 */
#define TFIMM_GELU_C0 1.569035798e-01f
#define TFIMM_GELU_C1 -7.623877842e-03f
#define TFIMM_GELU_C2 5.338436458e-04f
#define TFIMM_GELU_C3 -3.877354538e-05f
#define TFIMM_GELU_C4 2.691573627e-06f
#define TFIMM_GELU_C5 -1.703820232e-07f
#define TFIMM_GELU_C6 1.003279149e-08f
#define TFIMM_GELU_C7 -6.429004551e-10f
#define TFIMM_GELU_C8 3.842302865e-11f
#define TFIMM_GELU_C9 -1.143696032e-12f
#define TFIMM_GELU_POLY(R)                                                                                                    \
  (((((((((TFIMM_GELU_C9 * (R) + TFIMM_GELU_C8) * (R) + TFIMM_GELU_C7) * (R) + TFIMM_GELU_C6) * (R) + TFIMM_GELU_C5) * (R) +   \
       TFIMM_GELU_C4) * (R) + TFIMM_GELU_C3) * (R) + TFIMM_GELU_C2) * (R) + TFIMM_GELU_C1) * (R) + TFIMM_GELU_C0)
__device__ __forceinline__ float gelu_erf(float v) {
  const float t = __builtin_amdgcn_fmed3f(v, -TFIMM_GELU_CLAMP, TFIMM_GELU_CLAMP);
  const float r = fmaf(t, t, -TFIMM_GELU_CENTRE);
  const float g = TFIMM_GELU_POLY(r);
  return v * fmaf(t, g, 0.5f);
}
__device__ __forceinline__ float act1(float v, const ActParams& q) {
  v = fminf(fmaxf(v, q.lo), q.hi);
  // the empty asm statements keep hipcc from if-converting the (wave-uniform) class branches into
  // selects, which would evaluate exp/rcp of BOTH classes for every element of every activation
  if (q.cls == 1) {
    asm volatile("");
    const float sgm = __builtin_amdgcn_rcpf(1.f + __expf(-q.k * v));
    v = q.a * v * sgm + (q.b * sgm + q.c);
  } else if (q.cls == 2) {
    asm volatile("");
    v = gelu_erf(v);
  }
  return v;
}
__device__ __forceinline__ void act8(float* v, const ActParams& q) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(v[e], q.lo), q.hi);
  if (q.cls == 1) {   // wave-uniform branches, kept as branches (see act1)
    asm volatile("");
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float sgm = __builtin_amdgcn_rcpf(1.f + __expf(-q.k * v[e]));
      v[e] = q.a * v[e] * sgm + (q.b * sgm + q.c);
    }
  } else if (q.cls == 2) {
    asm volatile("");
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
  }
}

// the same on a pair: every operation but the clamp is a packed instruction
__device__ __forceinline__ tfimm_f32x2 gelu_erf2(tfimm_f32x2 v) {
  const tfimm_f32x2 t = {__builtin_amdgcn_fmed3f(v.x, -TFIMM_GELU_CLAMP, TFIMM_GELU_CLAMP),
                         __builtin_amdgcn_fmed3f(v.y, -TFIMM_GELU_CLAMP, TFIMM_GELU_CLAMP)};
  const tfimm_f32x2 r = t * t - TFIMM_GELU_CENTRE;
  const tfimm_f32x2 g = TFIMM_GELU_POLY(r);
  return v * (t * g + 0.5f);
}

// Same on four packed pairs (v_pk_* arithmetic, v_med3_f32 clamp); every class is a wave-uniform
// branch, so an epilogue pays only for the activation it has.
__device__ __forceinline__ void act8p(tfimm_f32x2* v, const ActParams& q) {
  if (q.clamp) {
    asm volatile("");
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e].x = __builtin_amdgcn_fmed3f(v[e].x, q.lo, q.hi);
      v[e].y = __builtin_amdgcn_fmed3f(v[e].y, q.lo, q.hi);
    }
  } else if (q.cls == 1) {
    asm volatile("");
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const tfimm_f32x2 z = v[e] * (-q.k * 1.4426950408889634f);          // exp(-k v) = exp2(-k v log2 e)
      const tfimm_f32x2 en = {__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)};
      const tfimm_f32x2 dn = 1.f + en;
      const tfimm_f32x2 sg = {__builtin_amdgcn_rcpf(dn.x), __builtin_amdgcn_rcpf(dn.y)};
      v[e] = q.a * v[e] * sg + (q.b * sg + q.c);
    }
  } else if (q.cls == 2) {
    asm volatile("");
    // the four pairs' Horner chains advance in lockstep: a chain by itself issues one dependent v_pk_fma_f32 after the
    // other (hipcc emitted exactly that, with a wait state between each), four interleaved keep the VALU busy
    tfimm_f32x2 t[4], r[4], g[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      t[e] = tfimm_f32x2{__builtin_amdgcn_fmed3f(v[e].x, -TFIMM_GELU_CLAMP, TFIMM_GELU_CLAMP),
                         __builtin_amdgcn_fmed3f(v[e].y, -TFIMM_GELU_CLAMP, TFIMM_GELU_CLAMP)};
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = t[e] * t[e] - TFIMM_GELU_CENTRE;
    constexpr float cf[10] = {TFIMM_GELU_C0, TFIMM_GELU_C1, TFIMM_GELU_C2, TFIMM_GELU_C3, TFIMM_GELU_C4,
                              TFIMM_GELU_C5, TFIMM_GELU_C6, TFIMM_GELU_C7, TFIMM_GELU_C8, TFIMM_GELU_C9};
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = r[e] * cf[9] + cf[8];
#pragma unroll
    for (int k = 7; k >= 0; --k) {
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = g[e] * r[e] + cf[k];
      // keep the four chains side by side (the scheduler otherwise re-serialises them)
      asm volatile("" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] * (t[e] * g[e] + 0.5f);
  }
}
// eight bf16 (one 16-byte row segment) -> four fp32 pairs
__device__ __forceinline__ void unpack8p(const uint4& u, tfimm_f32x2* f) {
  f[0].x = __uint_as_float(u.x << 16); f[0].y = __uint_as_float(u.x & 0xffff0000u);
  f[1].x = __uint_as_float(u.y << 16); f[1].y = __uint_as_float(u.y & 0xffff0000u);
  f[2].x = __uint_as_float(u.z << 16); f[2].y = __uint_as_float(u.z & 0xffff0000u);
  f[3].x = __uint_as_float(u.w << 16); f[3].y = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8p(const tfimm_f32x2* f) {
  uint4 u;
  u.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(f[0], tfimm_bf16x2));
  u.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(f[1], tfimm_bf16x2));
  u.z = __builtin_bit_cast(uint32_t, __builtin_convertvector(f[2], tfimm_bf16x2));
  u.w = __builtin_bit_cast(uint32_t, __builtin_convertvector(f[3], tfimm_bf16x2));
  return u;
}

// ---- wave64 reductions ---------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
