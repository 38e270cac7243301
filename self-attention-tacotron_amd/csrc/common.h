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

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

#define SATT_LAUNCH_CHECK()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return SATT_E_LAUNCH;              \
  } while (0)
