// Cross-workgroup exchange primitives of the cluster kernels (attention cluster, LSTM cluster): 8-byte {tag, value}
// granules in global memory - the data is the flag - published with single stores and gathered by bounded polling.
#pragma once
#include <mutex>
#include "common.h"

namespace {

constexpr int XW = 8;       // waves per workgroup of the cluster kernels (512 threads)
constexpr int GQ_MAP = 6;   // granules per lane at most in the mapped gather (as GQ below)
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

// Publish one granule.  same_xcd: every member of the cluster runs on the same XCD (verified at kernel start from
// HW_REG_XCC_ID), so a PLAIN store - which stays in that XCD's L2 - is visible to the members' sc1 polling loads;
// measured round trip 0.55 us vs 0.92 us for the agent-scope (write-through, sc1) store.  Otherwise agent scope.
__device__ __forceinline__ void gput(u64* g, uint32_t tag, float v, bool same_xcd) {
  const u64 x = ((u64)tag << 32) | (u64)__float_as_uint(v);
  if (same_xcd) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(g), "v"(x));
  else __hip_atomic_store((gu64*)g, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The same with the address as (uniform base, 32-bit element index): the store instruction takes the scalar base, the lane pays
// one shift instead of a 64-bit vector address (three to eight VALU instructions per store in phases that are issue bound).
__device__ __forceinline__ void gput_s(u64* base, unsigned idx, uint32_t tag, float v, bool same_xcd) {
  const u64 x = ((u64)tag << 32) | (u64)__float_as_uint(v);
  const unsigned off = idx * 8u;
  if (same_xcd) asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(off), "v"(x), "s"(base));
  else asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(off), "v"(x), "s"(base));
}
// agent-scope (write-through) float store, (uniform base, 32-bit element index)
__device__ __forceinline__ void gst_s(float* base, unsigned idx, float v) {
#ifdef SATT_CHUNK_FENCE      // (A/B switch of attn_cluster.hip: plain stores + a fence at the chunk boundary)
  asm volatile("global_store_dword %0, %1, %2" ::"v"(idx * 4u), "v"(v), "s"(base));
#else
  asm volatile("global_store_dword %0, %1, %2 sc1" ::"v"(idx * 4u), "v"(v), "s"(base));
#endif
}
// plain float store, (uniform base, 32-bit element index)
__device__ __forceinline__ void pst_s(float* base, unsigned idx, float v) {
  asm volatile("global_store_dword %0, %1, %2" ::"v"(idx * 4u), "v"(v), "s"(base));
}
__device__ __forceinline__ int xcc_id() { return (int)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }   // HW_REG_XCC_ID[3:0]
constexpr uint32_t XCC_TAG = 0xFFFFFFFFu;

// The polling loop shared by the gathers: N granules per lane per poll, every load of a poll issued before the first result is
// inspected (branch-free, so the compiler keeps all N in flight: a poll costs ONE L2 round trip; with a branch per granule it
// serialises them).
// Returns false on the bounded-spin timeout (the caller raises the error word).
// r5, tried and NOT kept (-DSATT_POLL_PIPELINED): TWO polls in flight - poll B issued before poll A's results are inspected (the
// vector-memory counter is in-order: the wait for A leaves B's N loads outstanding), so that a fresh poll reaches L2 every half
// round trip instead of one per round trip + sleep.  The ISA is as intended (no copies, vmcnt(N) waits), and the step got SLOWER:
// 7.48 -> 7.63 ms, attention backward launch 3.04 -> 3.15 ms (same box, tools/ab_bench.sh).  More polls in flight is not what
// the exchanges lack: the polling waves share the CU's vector-memory path with the waves that still have to publish.
#ifndef SATT_POLL_SLEEP
#define SATT_POLL_SLEEP 1      // s_sleep argument between two polls (units of 64 clocks)
#endif
template <int N>
__device__ __forceinline__ bool poll_until(const gu64* const (&g)[N], uint32_t tag, u64 (&x)[N]) {
#ifndef SATT_POLL_PIPELINED
  for (unsigned spins = 0;; ++spins) {
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] = __hip_atomic_load(g[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool all_ok = true;
#pragma unroll
    for (int q = 0; q < N; ++q) all_ok &= (uint32_t)(x[q] >> 32) == tag;
    if (__all(all_ok)) return true;
    if (spins > (1u << 21)) return false;
    __builtin_amdgcn_s_sleep(SATT_POLL_SLEEP);
  }
#else
  // (no copy between the two register sets inside the loop - a copy of an in-flight poll costs s_waitcnt vmcnt(0) per
  //  iteration, the drained-ring pattern of DESIGN.md 3.4 -: which set holds the result is a flag, selected behind the loop)
  u64 y[N];
  bool from_y = false, done = true;
#pragma unroll
  for (int q = 0; q < N; ++q) x[q] = __hip_atomic_load(g[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (unsigned spins = 0;; ++spins) {
#pragma unroll
    for (int q = 0; q < N; ++q) y[q] = __hip_atomic_load(g[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = true;
#pragma unroll
    for (int q = 0; q < N; ++q) ok &= (uint32_t)(x[q] >> 32) == tag;
    if (__all(ok)) break;
#pragma unroll
    for (int q = 0; q < N; ++q) x[q] = __hip_atomic_load(g[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok = true;
#pragma unroll
    for (int q = 0; q < N; ++q) ok &= (uint32_t)(y[q] >> 32) == tag;
    if (__all(ok)) { from_y = true; break; }
    if (spins > (1u << 20)) { done = false; break; }
  }
#pragma unroll
  for (int q = 0; q < N; ++q) x[q] = from_y ? y[q] : x[q];
  return done;
#endif
}

// poll_until with the kernels' time-out protocol: raises the error word and marks the workgroup dead (later gathers return at once)
template <int N>
__device__ __forceinline__ void poll_or_die(const gu64* const (&g)[N], uint32_t tag, u64 (&x)[N], int lane, unsigned int* err_word,
                                            int* dead) {
  if (!*dead) {
    if (!poll_until<N>(g, tag, x)) {
      if (lane == 0) __hip_atomic_store((gu32*)err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *dead = 1;
    }
  }
}

// N granules per lane (poll_until); lanes / slots beyond cnt re-read the last valid granule and are ignored.
template <int N, class St>
__device__ __forceinline__ void gather_poll(u64* src, int cnt, uint32_t tag, int lane, St store, unsigned int* err_word,
                                            int* dead) {
  if (cnt <= 0) return;
  u64 x[N];
  const gu64* g[N];
#pragma unroll
  for (int q = 0; q < N; ++q) { g[q] = (const gu64*)(src + min(lane + 64 * q, cnt - 1)); x[q] = 0; }
  if (!*dead) {
    if (!poll_until<N>(g, tag, x)) {
      if (lane == 0) __hip_atomic_store((gu32*)err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef SATT_XCHG_DEBUG      // who waited for what (first reporter wins): tail words 3.. = tag, count, block, first missing slot, its tag, XCC
      const unsigned long long okm = __ballot(((uint32_t)(x[0] >> 32) == tag) || lane >= cnt);
      if (lane == 0 && atomicAdd(err_word + 3, 1u) == 0u) {
        err_word[14] = (unsigned)okm;
        err_word[4] = tag; err_word[5] = (unsigned)cnt; err_word[6] = blockIdx.x | (blockIdx.y << 16);
        int miss = -1; unsigned mt = 0;
        for (int q = 0; q < N; ++q) if ((uint32_t)(x[q] >> 32) != tag && miss < 0) { miss = 64 * q; mt = (uint32_t)(x[q] >> 32); }
        err_word[7] = (unsigned)miss; err_word[8] = mt; err_word[9] = (unsigned)xcc_id();
        err_word[10] = __hip_atomic_load((gu32*)(err_word + 12), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // workgroups started so far
        err_word[11] = __hip_atomic_load((gu32*)(err_word + 13), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... and finished
      }
#endif
      *dead = 1;
    }
  }
#pragma unroll
  for (int q = 0; q < N; ++q) { const int i = lane + 64 * q; if (i < cnt) store(i, __uint_as_float((uint32_t)x[q])); }
}
// ONE wave gathers granules src[0..count) (count <= 256) carrying `tag`, calling store(i, value) for each.
template <class St>
__device__ __forceinline__ void gather_chunk(u64* src, int count, uint32_t tag, int lane, St store,
                                             unsigned int* err_word, int* dead) {
  if (count <= 64) gather_poll<1>(src, count, tag, lane, store, err_word, dead);
  else if (count <= 128) gather_poll<2>(src, count, tag, lane, store, err_word, dead);
  else gather_poll<4>(src, count, tag, lane, store, err_word, dead);
}
// The same over a MAPPED index set: virtual granule j lives at src[map(j)] (a member that needs only part of every peer's vector
// polls only that part); store receives the PHYSICAL index.
template <int N, class Map, class St>
__device__ __forceinline__ void gather_poll_map(u64* src, int beg, int cnt, Map map, uint32_t tag, int lane, St store,
                                                unsigned int* err_word, int* dead) {
  if (cnt <= 0) return;
  u64 x[N];
  int phys[N];
  const gu64* g[N];
#pragma unroll
  for (int q = 0; q < N; ++q) { phys[q] = map(beg + min(lane + 64 * q, cnt - 1)); g[q] = (const gu64*)(src + phys[q]); x[q] = 0; }
  if (!*dead) {
    if (!poll_until<N>(g, tag, x)) {
      if (lane == 0) __hip_atomic_store((gu32*)err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *dead = 1;
    }
  }
#pragma unroll
  for (int q = 0; q < N; ++q) { const int i = lane + 64 * q; if (i < cnt) store(phys[q], __uint_as_float((uint32_t)x[q])); }
}
template <class Map, class St>
__device__ __forceinline__ void gather_span_map(u64* src, int n, Map map, uint32_t tag, int part, int nparts, int lane, St store,
                                                unsigned int* err_word, int* dead) {
  const int per = (((n + nparts - 1) / nparts) + 63) & ~63;
  const int beg = part * per, cnt = min(per, n - beg);
  if (per <= 64) gather_poll_map<1>(src, beg, cnt, map, tag, lane, store, err_word, dead);
  else if (per <= 128) gather_poll_map<2>(src, beg, cnt, map, tag, lane, store, err_word, dead);
  else if (per <= 192) gather_poll_map<3>(src, beg, cnt, map, tag, lane, store, err_word, dead);
  else if (per <= 256) gather_poll_map<4>(src, beg, cnt, map, tag, lane, store, err_word, dead);
  else gather_poll_map<GQ_MAP>(src, beg, cnt, map, tag, lane, store, err_word, dead);
}

// all AW waves cooperate: chunk k (256 granules) is gathered by wave k % AW
template <class St>
__device__ __forceinline__ void gather_all(u64* src, int n, uint32_t tag, int wave, int lane, St store,
                                           unsigned int* err_word, int* dead) {
  for (int c0 = wave * 256; c0 < n; c0 += XW * 256)
    gather_chunk(src + c0, min(256, n - c0), tag, lane, [&](int i, float v) { store(c0 + i, v); }, err_word, dead);
}

// Wave `part` of `nparts` gathers its even share of src[0..n): ceil(n / nparts) granules rounded up to 64 lanes,
// at most 64*GQ of them (one poll loop, all loads of a lane in flight together).
constexpr int GQ = 6;
template <class St>
__device__ __forceinline__ void gather_span(u64* src, int n, uint32_t tag, int part, int nparts, int lane, St store,
                                            unsigned int* err_word, int* dead) {
  const int per = (((n + nparts - 1) / nparts) + 63) & ~63;
  const int beg = part * per, cnt = min(per, n - beg);      // cnt <= 0: nothing to do
  auto st = [&](int i, float v) { store(beg + i, v); };
  if (per <= 64) gather_poll<1>(src + beg, cnt, tag, lane, st, err_word, dead);
  else if (per <= 128) gather_poll<2>(src + beg, cnt, tag, lane, st, err_word, dead);
  else if (per <= 192) gather_poll<3>(src + beg, cnt, tag, lane, st, err_word, dead);
  else if (per <= 256) gather_poll<4>(src + beg, cnt, tag, lane, st, err_word, dead);
  else gather_poll<GQ>(src + beg, cnt, tag, lane, st, err_word, dead);
}


// ---- timing-robustness build (-DSATT_CLUSTER_JITTER=<max sleeps>; r6): behind every workgroup barrier of the cluster kernels each WAVE
// sleeps, with probability 1/8, a pseudo-random number of 3.4 us units - skew between the members of a cluster and between the
// waves of a member far beyond what the hardware produces.  The exchanges (tags, single-buffered granules, chunk hand-offs) must
// give the same results and no time-out under it.  Include-order: the .hip files redefine lds_barrier() AFTER their includes.
#ifdef SATT_CLUSTER_JITTER
__device__ __forceinline__ void cluster_jitter() {
  unsigned h = (unsigned)wall_clock64() ^ ((blockIdx.x + 977u * blockIdx.y) * 2654435761u) ^ ((threadIdx.x >> 6) * 2246822519u);
  h ^= h >> 15; h *= 2654435761u; h ^= h >> 13;
  h = __builtin_amdgcn_readfirstlane(h);
  if ((h & 7u) == 0u)
    for (unsigned i = 0, n = (h >> 8) % (unsigned)(SATT_CLUSTER_JITTER); i < n; ++i) __builtin_amdgcn_s_sleep(127);
}
#endif

// ---- host side: resident capacity of a cluster kernel --------------------------------------------------------------------------
// Every member of a cluster spins (bounded) for its peers: a launch is only correct if ALL its workgroups can be resident at once.
// capacity = (active workgroups per CU the occupancy calculator gives THIS kernel at THIS LDS size) x (CUs of the current device),
// asked once per (kernel, LDS size) and cached.  The launchers refuse a grid above it instead of assuming one workgroup per CU on
// 256 CUs; the engine sizes its layer pipeline from the same numbers (satt_*_cluster_residency).
struct ClusterCap { const void* fn; size_t smem; int per_cu, cus; };
inline int cluster_capacity(const void* fn, int threads, size_t smem, int* per_cu = nullptr, int* cus = nullptr) {
  static ClusterCap cache[32];
  static int n = 0;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  for (int i = 0; i < n; ++i)
    if (cache[i].fn == fn && cache[i].smem == smem) {
      if (per_cu) *per_cu = cache[i].per_cu;
      if (cus) *cus = cache[i].cus;
      return cache[i].per_cu * cache[i].cus;
    }
  int dev = 0, per = 0, ncu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, fn, threads, smem) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  if (n < 32) cache[n++] = ClusterCap{fn, smem, per, ncu};
  if (per_cu) *per_cu = per;
  if (cus) *cus = ncu;
  return per * ncu;
}

}  // namespace
