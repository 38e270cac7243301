// Helpers shared by the single-workgroup (attn_rnn.hip) and cluster (attn_cluster.hip) attention-RNN kernels.
#pragma once
#include "matvec.h"

#ifdef SATT_PROFILE
// per-phase wall-clock (100 MHz) accumulators of workgroup 0 / thread 0; read back with satt_prof_read()
static __device__ unsigned long long satt_prof_acc[32];   // one copy per translation unit
#define PROF_DECL unsigned long long prof_t0 = wall_clock64(), prof_a[16] = {0}
#ifdef SATT_TRACE_ONLY   // exchange traces: the accumulators would slow member 0 of sample 0 down and show up as skew
#define PROF(i)
#else
#define PROF(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { unsigned long long n_ = wall_clock64(); prof_a[i] += n_ - prof_t0; prof_t0 = n_; } } while (0)
#endif
#define PROF_STORE(base) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) for (int i_ = 0; i_ < 16; ++i_) satt_prof_acc[(base) + i_] = prof_a[i_]; } while (0)
// raw time stamps of the 4 members of sample 0 for the first 128 steps of a launch: satt_prof_trace[member][step][slot]
static __device__ unsigned long long satt_prof_trace[8 * 128 * 16];
#ifndef SATT_TRACE_B
#define SATT_TRACE_B 0       // the sample whose members are traced
#endif
#define TRACE(step, slot) do { if (blockIdx.x == SATT_TRACE_B && threadIdx.x == 0 && (step) < 128 && blockIdx.y < 8) satt_prof_trace[(blockIdx.y * 128 + (step)) * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define PROF_DECL
#ifdef SATT_ASM_MARKS    // listing aid (tools/asm_segments.py): a comment in the assembly at every phase boundary
#define PROF(i) asm volatile("; SATT_MARK " #i)
#else
#define PROF(i)
#endif
#define PROF_STORE(base)
#define TRACE(step, slot)
#endif

namespace {

constexpr int ANT = 512;
constexpr int AW = ANT / 64;  // waves
constexpr int NQ = 4;         // each lane owns 4 consecutive units / channels: U1, V1 <= 256, % 4 == 0
constexpr int MVU = 4;        // mat-vec rows in flight per thread (x2 with the double buffer)
constexpr int RB = 4;         // memory rows processed per wave iteration (interleaved reductions)

// softmax over v[0..len) by ONE wave (in place), zeros beyond len up to n
__device__ __forceinline__ void wave_softmax(float* v, int len, int n, int lane) {
  float m = -INFINITY;
  for (int t = lane; t < len; t += 64) m = fmaxf(m, v[t]);
  m = wave_max(m);
  float s = 0.f;
  for (int t = lane; t < len; t += 64) { float e = exp2f_(1.4426950408889634f * (v[t] - m)); v[t] = e; s += e; }
  s = wave_sum(s);
  const float inv = __builtin_amdgcn_rcpf(s);
  for (int t = lane; t < n; t += 64) v[t] = (t < len) ? v[t] * inv : 0.f;
}

// 4 consecutive key units of memory row tt for this lane
template <bool KLDS>
__device__ __forceinline__ void load_key4(const float* __restrict__ kglob, const uint16_t* __restrict__ klds, int tt,
                                          int U, int d0, bool act, float (&kk)[NQ]) {
  if (KLDS) {
    uint2 w = make_uint2(0u, 0u);
    if (act) w = *reinterpret_cast<const uint2*>(klds + tt * U + d0);
    kk[0] = __uint_as_float(w.x << 16); kk[1] = __uint_as_float(w.x & 0xFFFF0000u);
    kk[2] = __uint_as_float(w.y << 16); kk[3] = __uint_as_float(w.y & 0xFFFF0000u);
  } else {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) v = *reinterpret_cast<const float4*>(kglob + (size_t)tt * U + d0);
    kk[0] = v.x; kk[1] = v.y; kk[2] = v.z; kk[3] = v.w;
  }
}
template <bool KLDS>
__device__ __forceinline__ float load_key1(const float* __restrict__ kglob, const uint16_t* __restrict__ klds, int tt,
                                           int U, int d, bool act) {
  if (!act) return 0.f;
  if (KLDS) return bf2f(klds[tt * U + d]);
  return kglob[(size_t)tt * U + d];
}

// Branch-free variants for the cluster kernels: the lane's unit index is clamped into the row, so the load is issued
// unconditionally (a predicated load costs a branch plus an immediate wait).  Lanes beyond U read valid units whose
// contribution the caller cancels (zero weight in its parameter table, or a select on the result).
template <bool KLDS>
__device__ __forceinline__ void load_key4u(const float* __restrict__ kglob, const uint16_t* __restrict__ klds, int tt,
                                           int U, int d0, float (&kk)[NQ]) {
  const int dc = min(d0, U - NQ);
  if (KLDS) {
    const uint2 w = *reinterpret_cast<const uint2*>(klds + tt * U + dc);
    kk[0] = __uint_as_float(w.x << 16); kk[1] = __uint_as_float(w.x & 0xFFFF0000u);
    kk[2] = __uint_as_float(w.y << 16); kk[3] = __uint_as_float(w.y & 0xFFFF0000u);
  } else {
    const float4 v = *reinterpret_cast<const float4*>(kglob + (size_t)tt * U + dc);
    kk[0] = v.x; kk[1] = v.y; kk[2] = v.z; kk[3] = v.w;
  }
}
template <bool KLDS>
__device__ __forceinline__ float load_key1u(const float* __restrict__ kglob, const uint16_t* __restrict__ klds, int tt,
                                            int U, int d) {
  const int dc = max(min(d, U - 1), 0);        // U == 0 (no second source): element 0 of the stand-in row
  if (KLDS) return bf2f(klds[tt * U + dc]);
  return kglob[(size_t)tt * U + dc];
}

// Single-source form (U2 == 0 and V2 == 0: the baseline Tacotron decoder, reference modules/module.py:530-623): the
// second mechanism degenerates to zero energies, uniform alignments and an empty context.  Its pointers may be NULL;
// the kernels' clamped (always issued) loads are pointed at the first source, where they read finite values that
// zero weights cancel.
inline void single_source_fixup(satt_attn_rnn_params& p) {
  if (p.U2 == 0) { p.keys2 = p.keys1; p.v2 = p.v1; }
  if (p.V2 == 0) p.values2 = p.values1;
}

}  // namespace
