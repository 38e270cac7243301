// Shared device helpers for the satt HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/satt_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ---- counter-based dropout / zoneout mask: must match oracle/rng.py bit-for-bit ----
__device__ __forceinline__ uint32_t satt_hash(uint32_t seed, uint32_t stream, uint32_t idx) {
  uint32_t x = idx ^ (seed * 0x9E3779B1u);
  x += stream * 0x85EBCA6Bu;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool satt_keep(uint32_t seed, uint32_t stream, uint32_t idx, uint32_t thresh) {
  return satt_hash(seed, stream, idx) >= thresh;
}

__device__ __forceinline__ uint16_t f2bf(float f) {  // round-to-nearest-even fp32 -> bf16
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)(u >> 16);  // inf / nan passthrough
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// Workgroup barrier for LDS hand-offs inside the persistent kernels.  __syncthreads() is a workgroup-scope
// release/acquire fence + s_barrier, and the fence drains EVERY outstanding global load and store of the wave
// (s_waitcnt vmcnt(0)) -- i.e. each barrier would expose one L2 round trip for the prefetched loads and the
// fire-and-forget result stores of the step.  The kernels only hand LDS data across these barriers (global data is
// consumed by later launches or travels through the tagged-granule exchange), so only LDS traffic is waited for.
// Register consumers of in-flight global loads are still protected by the compiler's own s_waitcnt insertion.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// fast forms for the latency-critical recurrent kernels: v_exp_f32 + v_rcp_f32 (1 ulp each; abs error ~2e-7).
// NB: __fdividef() lowers to the full IEEE division sequence (v_div_scale/fmas/fixup, ~11 VALU ops) on gfx950.
__device__ __forceinline__ float exp2f_(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float sigmoidf_(float x) {
  return __builtin_amdgcn_rcpf(1.0f + exp2f_(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(exp2f_(2.8853900817779268f * x) + 1.0f);
}

// wave64 sum, result in every lane: 4 DPP stages inside each row of 16 lanes (quad_perm xor1, xor2,
// row_half_mirror, row_mirror) + 4 v_readlane for the 4 rows — no LDS round trips (ds_bpermute) at all.
#define SATT_DPP_ADD(v, ctrl) \
  (v) += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xF, 0xF, false))
__device__ __forceinline__ float wave_sum(float v) {
  SATT_DPP_ADD(v, 0xB1);    // quad_perm [1,0,3,2]
  SATT_DPP_ADD(v, 0x4E);    // quad_perm [2,3,0,1]
  SATT_DPP_ADD(v, 0x141);   // row_half_mirror
  SATT_DPP_ADD(v, 0x140);   // row_mirror
  return (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)) +
          __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)) +
          __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48)));
}
#define SATT_DPP_MAX(v, ctrl) \
  (v) = fmaxf((v), __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xF, 0xF, false)))
__device__ __forceinline__ float wave_max(float v) {
  SATT_DPP_MAX(v, 0xB1); SATT_DPP_MAX(v, 0x4E); SATT_DPP_MAX(v, 0x141); SATT_DPP_MAX(v, 0x140);
  const int b = __float_as_int(v);
  return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 0)), __int_as_float(__builtin_amdgcn_readlane(b, 16))),
               fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 32)), __int_as_float(__builtin_amdgcn_readlane(b, 48))));
}

// N independent wave-wide sums with their butterfly stages interleaved (hides the cross-lane latency)
template <int N>
__device__ __forceinline__ void wave_sum_multi(float (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) SATT_DPP_ADD(v[i], 0xB1);
#pragma unroll
  for (int i = 0; i < N; ++i) SATT_DPP_ADD(v[i], 0x4E);
#pragma unroll
  for (int i = 0; i < N; ++i) SATT_DPP_ADD(v[i], 0x141);
#pragma unroll
  for (int i = 0; i < N; ++i) SATT_DPP_ADD(v[i], 0x140);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int b = __float_as_int(v[i]);
    v[i] = (__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48)));
  }
}

#define SATT_LAUNCH_CHECK()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return SATT_E_LAUNCH;              \
  } while (0)
