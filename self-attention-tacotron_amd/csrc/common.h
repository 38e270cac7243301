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

// two fp32 -> packed bf16 pair (lo = a, hi = b) with the gfx950 conversion instruction v_cvt_pk_bf16_f32
// (round-to-nearest-even: bitwise equal to f2bf for finite values)
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  const f32x2_ v = {a, b};
  const bf16x2_ r = __builtin_convertvector(v, bf16x2_);
  return __builtin_bit_cast(uint32_t, r);
}
// round-to-nearest-even fp32 -> bf16: one v_cvt_pk_bf16_f32 (the manual bit trick cost ~7 VALU operations, and the
// exact 3-way operand split of the recurrent mat-vecs converts three times per value per step)
__device__ __forceinline__ uint16_t f2bf(float f) { return (uint16_t)pack_bf16x2(f, f); }
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

// Element idx of a row whose pointer is wave-uniform, with the byte offset formed in 32 bits before it meets the pointer: the access
// takes the scalar base + a 32-bit vector offset.  Written as `base[(size_t)row * n + idx]` the compiler forms a 64-bit VECTOR
// address per access - three to eight VALU instructions each, in recurrent kernels whose phases are instruction-issue bound (r5).
__device__ __forceinline__ float ld_su(const float* base, unsigned idx) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + idx * 4u);
}
// CONTRACT of st_su: the store is an asm statement WITHOUT a "memory" clobber (with one, the compiler keeps every LDS operand read of
// the encoder LSTM's step behind it: +4 exposed LDS latencies per step, the r5 finding of DESIGN.md 3.3).  The compiler therefore
// neither sees the write nor orders ordinary loads / stores against it.  It is valid ONLY for write-only outputs that the kernel
// never re-reads and never signals to another workgroup (saved tensors of csrc/lstm.hip, consumed by LATER launches); it must not
// appear in the exchange paths (cluster_xchg.h has gput_s / gst_s / pst_s for those - tests/test_mfma_hazard_cpu.py checks both).
__device__ __forceinline__ void st_su(float* base, unsigned idx, float v) {     // (an asm store: not seen by the compiler's wait counts,
  asm volatile("global_store_dword %0, %1, %2" ::"v"(idx * 4u), "v"(v), "s"(base));    //  which only makes its waits conservative)
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
  // the four row sums -> every lane: the gfx950 row / half swaps (r5; one VALU instruction per stage and value instead of four lane
  // reads into scalar registers, three adds and the copies back): (row 0 + row 1) + (row 2 + row 3), the order of the lane reads
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const unsigned b = __float_as_uint(v[i]);
    const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const unsigned b = __float_as_uint(v[i]);
    const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
}

// Transposing wave reduction: N (power of two <= 16) independent sums over the 64 lanes in ~3N instructions instead
// of ~11N.  Every butterfly stage halves the number of live values: lanes whose stage bit is 0 keep the even
// member of each pair and send the odd one, and vice versa.  Returns, in EVERY lane l, the wave-wide total of
// value index (l & (N-1)).  xor 1 / xor 2 exchange by DPP quad_perm, xor 4 / 8 / 16 by ds_swizzle (bit-mask mode,
// no LDS memory), xor 32 by ds_bpermute.
__device__ __forceinline__ float swz_xor(float v, int mask) {    // mask must be a compile-time 4 / 8 / 16
  const int b = __float_as_int(v);
  if (mask == 4) return __int_as_float(__builtin_amdgcn_ds_swizzle(b, 0x101F));
  if (mask == 8) return __int_as_float(__builtin_amdgcn_ds_swizzle(b, 0x201F));
  return __int_as_float(__builtin_amdgcn_ds_swizzle(b, 0x401F));
}
template <int N>
__device__ __forceinline__ float wave_sum_transpose(float (&v)[N]) {
  static_assert(N == 2 || N == 4 || N == 8 || N == 16, "N must be a power of two <= 16");
  const int lane = (int)__lane_id();
  float w[N];
#pragma unroll
  for (int i = 0; i < N; ++i) w[i] = v[i];
  int n = N;
  if (n > 1) {          // bit 0: xor 1
    const bool hi = lane & 1;
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      const float keep = hi ? w[2 * i + 1] : w[2 * i], send = hi ? w[2 * i] : w[2 * i + 1];
      w[i] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1, 0xF, 0xF, false));
    }
    n >>= 1;
  } else { SATT_DPP_ADD(w[0], 0xB1); }
  if (N >= 4) {         // bit 1: xor 2
    const bool hi = lane & 2;
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
      const float keep = hi ? w[2 * i + 1] : w[2 * i], send = hi ? w[2 * i] : w[2 * i + 1];
      w[i] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x4E, 0xF, 0xF, false));
    }
  } else { SATT_DPP_ADD(w[0], 0x4E); }
  if (N >= 8) {         // bit 2: xor 4
    const bool hi = lane & 4;
#pragma unroll
    for (int i = 0; i < N / 8; ++i) {
      const float keep = hi ? w[2 * i + 1] : w[2 * i], send = hi ? w[2 * i] : w[2 * i + 1];
      w[i] = keep + swz_xor(send, 4);
    }
  } else { w[0] += swz_xor(w[0], 4); }
  if (N >= 16) {        // bit 3: xor 8
    const bool hi = lane & 8;
    const float keep = hi ? w[1] : w[0], send = hi ? w[0] : w[1];
    w[0] = keep + swz_xor(send, 8);
  } else { w[0] += swz_xor(w[0], 8); }
  w[0] += swz_xor(w[0], 16);
  w[0] += __int_as_float(__builtin_amdgcn_ds_bpermute(((lane ^ 32) << 2), __float_as_int(w[0])));
  return w[0];
}

#define SATT_LAUNCH_CHECK()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return SATT_E_LAUNCH;              \
  } while (0)
