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
//   gelu (exact erf form; erf by a degree-10 polynomial, |err| <= 3.2e-6: see gelu_erf)
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
// exact (erf) GELU, 0.5 v (1 + erf(v / sqrt 2)), with NO transcendental instruction: erf(z) = z g(z^2) on |z| <= 3.2 with g a
// degree-10 polynomial in s = z^2 / 5.12 - 1 (Chebyshev fit, Horner in s so the coefficients stay O(1)); beyond 3.2 the
// clamped argument gives +-0.999997.  max |erf error| 3.2e-6, max |GELU error| 1.2e-5 (at |v| = 8, i.e. 1.5e-6 relative) --
// two orders below the bf16 resolution of the values it produces.  11 FMAs, all packable (v_pk_fma_f32): the
// Abramowitz-Stegun form it replaces spent 2 quarter-rate transcendentals (v_rcp, v_exp) per element and made the
// K = 768 fc1 layers of ViT-B VALU-bound in their epilogue (615 us vs 500 us for the same GEMM without activation).
#define TFIMM_GELU_POLY(S)                                                                                              \
  ((((((((((2.982273698e-03f * (S) - 7.046153303e-03f) * (S) + 7.957076654e-03f) * (S) - 1.521942858e-02f) * (S) +       \
         3.318292275e-02f) * (S) - 5.471928790e-02f) * (S) + 8.062700182e-02f) * (S) - 1.136467382e-01f) * (S) +         \
      1.543549746e-01f) * (S) - 2.173077315e-01f) * (S) + 4.413341880e-01f)
__device__ __forceinline__ float gelu_erf(float v) {
  const float z = __builtin_amdgcn_fmed3f(v * 0.70710678118654752f, -3.2f, 3.2f);
  const float sv = fmaf(z * z, 0.1953125f, -1.f);
  const float g = TFIMM_GELU_POLY(sv);
  const float h = 0.5f * v;
  return fmaf(h, z * g, h);
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
  const tfimm_f32x2 zz = v * 0.70710678118654752f;
  const tfimm_f32x2 z = {__builtin_amdgcn_fmed3f(zz.x, -3.2f, 3.2f), __builtin_amdgcn_fmed3f(zz.y, -3.2f, 3.2f)};
  const tfimm_f32x2 sv = z * z * 0.1953125f - 1.f;
  const tfimm_f32x2 g = TFIMM_GELU_POLY(sv);
  const tfimm_f32x2 h = 0.5f * v;
  return h * (z * g) + h;
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
    tfimm_f32x2 z[4], sv[4], g[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const tfimm_f32x2 zz = v[e] * 0.70710678118654752f;
      z[e] = tfimm_f32x2{__builtin_amdgcn_fmed3f(zz.x, -3.2f, 3.2f), __builtin_amdgcn_fmed3f(zz.y, -3.2f, 3.2f)};
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sv[e] = z[e] * z[e] * 0.1953125f - 1.f;
    constexpr float cf[11] = {4.413341880e-01f, -2.173077315e-01f, 1.543549746e-01f, -1.136467382e-01f, 8.062700182e-02f,
                              -5.471928790e-02f, 3.318292275e-02f, -1.521942858e-02f, 7.957076654e-03f, -7.046153303e-03f,
                              2.982273698e-03f};
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = sv[e] * cf[10] + cf[9];
#pragma unroll
    for (int k = 8; k >= 0; --k) {
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = g[e] * sv[e] + cf[k];
      // keep the four chains side by side (the scheduler otherwise re-serialises them)
      asm volatile("" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const tfimm_f32x2 h = 0.5f * v[e];
      v[e] = h * (z[e] * g[e]) + h;
    }
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
