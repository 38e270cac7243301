// Persistent form of the autoregressive decode step (BASELINE config 5; reference modules/module.py:762-778,
// modules/rnn_wrappers.py:47-124,188-214, StopTokenBasedInferenceHelper): ONE launch runs `nsteps` whole decoder steps.
// csrc/decode.hip walks a step as 9 dependent launches of >= 4.7 us each (57 us per step at B = 1: launch bound, not arithmetic
// bound); here the step is walked by NWG persistent workgroups that meet at SIX device-wide barriers per step:
//   A  pre-net 0 -> pre-net 1 (every workgroup, redundantly) -> own 8 units of the attention LSTM cell            | barrier 1
//   B  processed query h W_q (redundant) -> energies of both mechanisms for the own slice of memory rows           | barrier 2
//   C  softmax + forward recursion + both contexts (redundant) -> own 8 units of LSTM 1                            | barrier 3
//   D  own 8 units of LSTM 2                                                                                       | barrier 4
//   E  own 32 columns of the K | V | Q projection -> row t of the cache                                            | barrier 5
//   F  own (head, key chunk) of the causal self-attention over the cache: partial (max, sum, P V)                  | barrier 6
//   G  merge of the partials, folded output transform + tanh + residual, mel | stop projection (redundant): every
//      workgroup ends the step holding the fed-back frame - the next step starts without a barrier
// "Redundant" layers are the small ones (<= 256 x 256): recomputing them in every workgroup costs one L2 stream of <= 128 KB
// per CU (< 1 us) where splitting them would cost a barrier (~1 us) each.  The math is that of csrc/decode.hip (bf16 weight
// shadows, fp32 accumulation, ZoneoutLSTMCell in interpolation mode, masked softmax, forward-attention recursion).
// Exchange: what another workgroup reads within a step is written with agent-scope (write-through) stores and read with
// agent-scope loads; the barrier is one agent-scope atomic counter, bounded spins, sticky error word (cluster_xchg.h rules).
// The active workgroups are blockIdx.x % 8 == 0 of an 8 x NWG grid: the dispatcher places workgroup i on XCD i % 8, so all of
// them share ONE L2 (32 CUs) - an optimisation only: nothing relies on the placement.
// Supported: dual-source model, plain two-layer pre-net, no transition agent, no forced alignments, bf16 weights, B <= 4,
// Ti <= 256, one causal self-attention hop; everything else takes the launch-per-layer path (inference.DecodeSession).
#include <cstdlib>
#include "common.h"

#ifdef SATT_MEGA_PROF      // per-phase wall-clock sums (100 MHz) of workgroup 0: tools/build_variant.sh + tools/decode_mega_prof.py
static __device__ unsigned long long satt_mega_prof[32];
#define MPROF(i) do { if (wg == 0 && threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); satt_mega_prof[i] += n_ - mp_last; mp_last = n_; } } while (0)
#else
#define MPROF(i)
#endif

namespace {

constexpr int MNT = 512, MNW = MNT / 64;        // threads / waves per workgroup (256 registers per thread: 1024 threads spilled)
constexpr int MWG = 32;                         // persistent workgroups
#ifdef SATT_MEGA_ONE_XCD
constexpr int MEGA_GRID = 8 * MWG;
#else
constexpr int MEGA_GRID = MWG;                  // workgroup i runs on XCD i % 8: four per XCD
#endif
constexpr int MWN = 256;                        // widest redundant layer (columns), also the widest K of one
constexpr int MKS = 1024;                       // largest K of a sliced (LSTM / K|V|Q) product
constexpr int MTI = 256, MCT = 320, MNO = 164;
typedef unsigned int u32;
typedef __attribute__((address_space(1))) u32 gu32m;
typedef __attribute__((address_space(1))) float gf32m;

__device__ __forceinline__ void ast(float* p, float v) { __hip_atomic_store((gf32m*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ald(const float* p) { return __hip_atomic_load((const gf32m*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// device-wide barrier in two halves: ARRIVE (every thread's exchange stores have left - vmcnt -, one arrival per workgroup) and WAIT
// (bounded spin).  Loads that do not depend on the other workgroups - the next phase's weights - are issued BETWEEN the halves:
// in front of the arrival they would delay it (the counter is in-order: vmcnt(0) waits for them too), behind the wait they would
// cost their round trip on the step's dependency chain.
// r5b: one flag WORD per workgroup instead of one shared counter - 32 atomic adds on one address serialise in the L2's atomic unit;
// a workgroup stores its epoch into its own slot (write-through) and wave 0 of every workgroup polls all 32 slots with one load per lane.
__device__ __forceinline__ void bar_arrive(u32* flags, u32& epoch, const int* dead, int wg) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  epoch += 1;
  if (threadIdx.x == 0 && !*dead) __hip_atomic_store((gu32m*)(flags + wg), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void bar_wait(u32* flags, u32 epoch, u32* err, int* dead) {
  if (threadIdx.x < 64 && !*dead) {
    const int l = threadIdx.x & 63;
    unsigned spins = 0;
    for (;;) {
      const u32 v = __hip_atomic_load((const gu32m*)(flags + (l & (MWG - 1))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all((int)(v - epoch) >= 0)) break;
      if (++spins > (1u << 22)) {
        if (l == 0) { __hip_atomic_store((gu32m*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *dead = 1; }
        break;
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void unpack4(uint2 v, float (&w)[4]) {
  w[0] = __uint_as_float(v.x << 16); w[1] = __uint_as_float(v.x & 0xFFFF0000u);
  w[2] = __uint_as_float(v.y << 16); w[3] = __uint_as_float(v.y & 0xFFFF0000u);
}

// ---- redundant ("wide") layer: out[b][n] = act(sum_k x[b][k] W[k][n] + bias[n]) (+ res[b][n]) for ALL n < N <= 256, K <= 256.
// lane = column group (4 columns), wave = k lane (rows wave, wave + 16, ...); partials of the 16 waves meet in `red`.
__host__ __device__ constexpr int red_floats(int nb) { return 2 * MNW * nb * MWN > 4096 ? 2 * MNW * nb * MWN : 4096; }
__host__ inline size_t mega_lds_bytes(int NB) {
  const size_t fl = (size_t)red_floats(NB) + (size_t)NB * (MKS + MNO + MCT + (MTI + 16) + MTI + MWN * 3 + 2 * MTI + 32) + 64 + 4 + 8 * MWN + 17 * 8 + 3 * MWN +
                    (size_t)NB * 8 * (MWN + 64) + NB + 4;
  return fl * sizeof(float);
}
// 16-byte weight loads (8 bf16 columns per lane): half the vector-memory instructions of the 8-byte form.  The instructions in
// flight per CU are bounded: with ~1800 wave-level loads per step the waves stalled at ISSUE for several round-trip generations.
constexpr int WKI = MWN / (2 * MNW);    // weight rows per thread of a wide layer: lane = (row parity l >> 5, column group l & 31 of 8)
struct WideW { uint4 v[WKI]; };
__device__ __forceinline__ void unpack8(uint4 v, float (&w)[8]) {
  w[0] = __uint_as_float(v.x << 16); w[1] = __uint_as_float(v.x & 0xFFFF0000u); w[2] = __uint_as_float(v.y << 16); w[3] = __uint_as_float(v.y & 0xFFFF0000u);
  w[4] = __uint_as_float(v.z << 16); w[5] = __uint_as_float(v.z & 0xFFFF0000u); w[6] = __uint_as_float(v.w << 16); w[7] = __uint_as_float(v.w & 0xFFFF0000u);
}
// the loads of a wide layer's weights: issued EARLY (before the barrier / the staging that precedes the product - the weights do not
// depend on the step's data; a load issued where it is consumed costs a whole L2 / MALL round trip on the step's dependency chain)
// row of (wave, i, half): 2 * (wave + MNW * i) + half
__device__ __forceinline__ void wide_load(WideW& w, const uint16_t* __restrict__ W, int ldw, int K, int tid) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5;
  const int nc = min(8 * (lane & 31), max(ldw - 8, 0));
#pragma unroll
  for (int i = 0; i < WKI; ++i) w.v[i] = *reinterpret_cast<const uint4*>(W + (int64_t)min(2 * (wave + MNW * i) + half, K - 1) * ldw + nc);
}
template <int NB>
__device__ __forceinline__ void wide_compute(const WideW& wv, const float* x, int xs_, int K, int N, const float* __restrict__ bias, int act,
                                             const float* res, int rs_, float* out, int os_, float* red, int tid) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, cgp = lane & 31;
  const bool colok = 8 * cgp < N;
  float acc[NB][8];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[b][j] = 0.f;
  // BRANCH-FREE, every LDS read of the input rows requested before the first product (a guard per weight row became a basic block
  // per row with its own LDS wait); rows beyond K read a valid element against zero weights
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float xv[WKI];
#pragma unroll
    for (int i = 0; i < WKI; ++i) xv[i] = x[b * xs_ + min(2 * (wave + MNW * i) + half, K - 1)];
#pragma unroll
    for (int i = 0; i < WKI; ++i) {
      float w[8];
      unpack8(wv.v[i], w);
      const float xm = (2 * (wave + MNW * i) + half < K && colok) ? xv[i] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[b][j] += xm * w[j];
    }
  }
  // partial of k lane (wave, half): red[(2 wave + half)][b][256]
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float* dst = red + ((2 * wave + half) * NB + b) * MWN + 8 * cgp;
    *reinterpret_cast<float4*>(dst) = make_float4(acc[b][0], acc[b][1], acc[b][2], acc[b][3]);
    *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[b][4], acc[b][5], acc[b][6], acc[b][7]);
  }
  const float bv = bias ? bias[min(tid & (MWN - 1), N - 1)] : 0.f;      // (MNT is a multiple of MWN: a thread's column is fixed)
  lds_barrier();
  for (int e = tid; e < NB * MWN; e += MNT) {
    const int b = e / MWN, n = e - b * MWN;
    if (n < N) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 2 * MNW; ++w) s += red[(w * NB + b) * MWN + n];
      s += bv;
      if (act == SATT_ACT_RELU) s = fmaxf(s, 0.f);
      else if (act == SATT_ACT_TANH) s = tanhf_(s);
      if (res) s += res[b * rs_ + n];
      out[b * os_ + n] = s;
    }
  }
  lds_barrier();
}

// ---- sliced product: the workgroup's 32 columns [n0, n0 + 32) of x W, K <= 1024; z[b][32] (LDS) receives the sums.
// thread = (column group tid & 3 of 8 columns, k lane tid >> 2 of MNT / 4): rows kl, kl + MNT / 4, ...; the 16 k lanes of a wave fold
// by shuffles.
constexpr int SKL = MNT / 4, SKI = MKS / SKL;
struct SliceW { uint4 v[SKI]; };
__device__ __forceinline__ void slice_load(SliceW& w, const uint16_t* __restrict__ W, int ldw, int n0, int K, int tid) {
  const int cg = tid & 3, kl = tid >> 2;
#pragma unroll
  for (int i = 0; i < SKI; ++i) w.v[i] = *reinterpret_cast<const uint4*>(W + (int64_t)min(kl + SKL * i, K - 1) * ldw + n0 + 8 * cg);
}
template <int NB>
__device__ __forceinline__ void slice_compute(const SliceW& wv, const float* xs, int K, float* z, float* red, int tid) {
  const int cg = tid & 3, kl = tid >> 2, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float acc[NB][8];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[b][j] = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b) {            // branch-free, input rows requested first (see wide_compute)
    float xv[SKI];
#pragma unroll
    for (int i = 0; i < SKI; ++i) xv[i] = xs[b * MKS + min(kl + SKL * i, K - 1)];
#pragma unroll
    for (int i = 0; i < SKI; ++i) {
      float w[8];
      unpack8(wv.v[i], w);
      const float xm = kl + SKL * i < K ? xv[i] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[b][j] += xm * w[j];
    }
  }
  // the 16 k lanes of a wave (lanes cg + 4 q): xor 4, 8, 16 by ds_swizzle, xor 32 by ds_bpermute
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = acc[b][j];
      v += swz_xor(v, 4); v += swz_xor(v, 8); v += swz_xor(v, 16);
      v += __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(v)));
      acc[b][j] = v;
    }
  if (lane < 4) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float* dst = red + (wave * NB + b) * 32 + 8 * cg;
      *reinterpret_cast<float4*>(dst) = make_float4(acc[b][0], acc[b][1], acc[b][2], acc[b][3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[b][4], acc[b][5], acc[b][6], acc[b][7]);
    }
  }
  lds_barrier();
  if (tid < NB * 32) {
    const int b = tid >> 5, n = tid & 31;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < MNW; ++w) s += red[(w * NB + b) * 32 + n];
    z[b * 32 + n] = s;
  }
  lds_barrier();
}

// own-unit state and gate biases of a cell: requested early (see wide_load); consumed by lstm_cell
struct CellIn { float c_old, h_old, b4[4]; };
template <int NB>
__device__ __forceinline__ void cell_load(CellIn& ci, const float* __restrict__ bias, int H, int wg, const float* c_state, const float* h_state,
                                          int par, int B, int tid) {
  const int b = min(tid >> 3, B - 1), eu = min(8 * wg + (tid & 7), H - 1);
  const int64_t oi = (int64_t)par * B * H + (int64_t)b * H + eu;
  ci.c_old = ald(c_state + oi); ci.h_old = ald(h_state + oi);
#pragma unroll
  for (int g = 0; g < 4; ++g) ci.b4[g] = bias[g * H + eu];
}

// ZoneoutLSTMCell (inference mode) of the workgroup's 8 units from z[b][gate * 8 + u] (regrouped columns: csrc/decode.hip)
template <int NB>
__device__ __forceinline__ void lstm_cell(const float* z, const CellIn& ci, int H, int wg, float* c_state, float* h_state, int par,
                                          int B, float zc, float zh, float* hn_out, int tid) {
  if (tid < NB * 8) {
    const int b = tid >> 3, u = tid & 7, eu = 8 * wg + u;
    if (b < B && eu < H) {
      const float* zb = z + b * 32;
      const float zi = zb[u] + ci.b4[0], zj = zb[8 + u] + ci.b4[1], zf = zb[16 + u] + ci.b4[2], zo = zb[24 + u] + ci.b4[3];
      const int64_t oo = (int64_t)(par ^ 1) * B * H + (int64_t)b * H + eu;
      const float cn = sigmoidf_(zf + 1.f) * ci.c_old + sigmoidf_(zi) * tanhf_(zj);
      const float hn = sigmoidf_(zo) * tanhf_(cn);
      ast(c_state + oo, (1.f - zc) * cn + zc * ci.c_old);
      ast(h_state + oo, (1.f - zh) * hn + zh * ci.h_old);             // read by every workgroup one step later
      ast(hn_out + (int64_t)b * H + eu, hn);                          // the cell output BEFORE zoneout
    }
  }
}

template <int NB>
__global__ __launch_bounds__(MNT) void dec_mega_k(const satt_dec_mega_params p) {
#ifdef SATT_MEGA_ONE_XCD      // (measured slower: the step's 4.9 MB of weights do not fit ONE 4 MB L2 - see the header comment)
  if ((blockIdx.x & 7) != 0) return;
  const int wg = blockIdx.x >> 3, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#else
  const int wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#endif
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // ---- LDS (floats)
  float* red = smem;                                 // [8 waves][NB][256] floats (at least 4096): reductions; also phase scratch
  float* xs = red + red_floats(NB);                  // [NB][1024]: input rows of the sliced products
  float* yv = xs + NB * MKS;                         // [NB][164]: this step's output row (the fed-back frame)
  float* ctx = yv + NB * MNO;                        // [NB][320]
  float* aprev = ctx + NB * MCT;                     // [NB][256 + 16]: location-conv input with zero borders (PL in front)
  float* alpha = aprev + NB * (MTI + 16);            // [NB][256]
  float* hq = alpha + NB * MTI;                      // [NB][256]: attention cell output (query)
  float* va = hq + NB * MWN;                         // [NB][256] scratch vectors
  float* vb = va + NB * MWN;
  float* e1 = vb + NB * MWN;                         // [NB][256] energies / softmax numerators
  float* e2 = e1 + NB * MTI;
  float* zs = e2 + NB * MTI;                         // [NB][32]
  float* sm = zs + NB * 32;                          // [64] block reductions
  int* dead = reinterpret_cast<int*>(sm + 64);
  const int B = p.B, Ti = p.Ti, A = p.A, D = p.D, Ds = p.Ds, U1 = p.U1, U2 = p.U2, UQ = U1 + U2, V1 = p.V1, V2 = p.V2, CT = V1 + V2;
  const int NO = p.NO, KW = p.kernel, F = p.filters, PL = (KW - 1) / 2, heads = p.heads, hd = Ds / heads;
  float* Us = reinterpret_cast<float*>(dead + 4);     // [8][256]: location-feature map U (filters <= 8)
  float* Fs = Us + 8 * MWN;                           // [16 taps][8] location filters | [8] their biases
  float* tab = Fs + 17 * 8;                           // [3][256]: v1 | b1 | v2
  float* kls = tab + 3 * MWN;                         // [NB * 8 own rows][256 + 64]: keys of both mechanisms for the own memory rows
  int* lens = reinterpret_cast<int*>(kls + NB * 8 * (MWN + 64));   // [NB]
  if (tid == 0) *dead = 0;
  for (int i = tid; i < 8 * MWN; i += MNT) { const int f = i / MWN, u = i - f * MWN; Us[i] = (f < F && u < U1) ? p.locU[f * U1 + u] : 0.f; }
  for (int i = tid; i < 3 * MWN; i += MNT) {
    const int w = i / MWN, u = i - w * MWN;
    tab[i] = w == 0 ? (u < U1 ? p.v1[u] : 0.f) : (w == 1 ? (u < U1 ? p.b1[u] : 0.f) : (u < U2 ? p.v2[u] : 0.f));
  }
  if (tid < NB) lens[tid] = tid < B ? (int)p.lengths[tid] : 0;
  {   // keys of the own rows (constant over the steps of an utterance)
    const int Rk = (Ti + MWG - 1) / MWG;
    for (int i = tid; i < NB * 8 * (MWN + 64); i += MNT) {
      const int row = i / (MWN + 64), u = i - row * (MWN + 64), b = row / 8, rr = row - b * 8, tt = wg * Rk + rr;
      float v = 0.f;
      if (b < B && rr < Rk && tt < Ti) v = u < MWN ? (u < U1 ? p.keys1[((int64_t)b * Ti + tt) * U1 + u] : 0.f) : (u - MWN < U2 ? p.keys2[((int64_t)b * Ti + tt) * U2 + u - MWN] : 0.f);
      kls[i] = v;
    }
  }
  for (int i = tid; i < 17 * 8; i += MNT) { const int j = i >> 3, f = i & 7; Fs[i] = f < F ? (j < KW ? p.locF[j * F + f] : (j == 16 ? p.locFb[f] : 0.f)) : 0.f; }
  for (int i = tid; i < NB * (MTI + 16); i += MNT) aprev[i] = 0.f;
  __syncthreads();
  // state that every workgroup carries redundantly: restored from the global copies (workgroup 0 keeps them current)
  int t = *p.step;
  {
    const int par = t & 1;
    for (int i = tid; i < NB * Ti; i += MNT) {
      const int b = i / Ti, r = i - b * Ti;
      if (b < B) {
        aprev[b * (MTI + 16) + PL + r] = p.a_state[((int64_t)par * B + b) * Ti + r];
        alpha[b * MTI + r] = p.alpha_state[((int64_t)par * B + b) * Ti + r];
      }
    }
    for (int i = tid; i < NB * CT; i += MNT) { const int b = i / CT, c = i - b * CT; ctx[b * MCT + c] = b < B ? p.ctx[((int64_t)(par ^ 1) * B + b) * CT + c] : 0.f; }
    for (int i = tid; i < NB * NO; i += MNT) { const int b = i / NO, c = i - b * NO; yv[b * MNO + c] = b < B ? p.yout[((int64_t)b * (p.Td + 1) + t) * NO + c] : 0.f; }
  }
  u32 target = *p.bar_base;                          // barrier epoch at launch
  __syncthreads();
  const int R = (Ti + MWG - 1) / MWG, r0 = wg * R;   // own slice of memory rows
  const int NCH = MWG / heads;                       // key chunks per head in the self-attention
  const int nsteps = p.nsteps;
  WideW wn; SliceW sn; CellIn cn;               // operands requested ahead of their phase (see wide_load)
  wide_load(wn, p.Wp0, p.P0, p.feed, threadIdx.x);
#ifdef SATT_MEGA_PROF
  unsigned long long mp_last = wall_clock64();
#endif
  for (int s = 0; s < nsteps; ++s, ++t) {
    // the ~70 kernel arguments are re-read from the kernarg segment inside every step (scalar loads next to their use) instead of
    // living - spilled - in scalar registers for the whole loop (the idiom of csrc/attn_cluster.hip)
    typedef const __attribute__((address_space(4))) satt_dec_mega_params KArgsM;
    KArgsM* kq = (KArgsM*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kq));
    const auto& p = *kq;
    // an opaque per-step zero keeps every thread-index expression INSIDE the step: left to itself the compiler hoists the address
    // arithmetic of all nine products out of the step loop and spills it (767 scalar + 350 vector spills before this)
    int oz = 0;
    asm volatile("" : "+v"(oz));
    const int tid = (int)threadIdx.x + oz, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(tid >= 0 && tid < MNT && wave >= 0 && wave < MNW);
    const int par = t & 1;
    // =========================================================== A: pre-nets + attention LSTM
    {
      // fed frame: the last `feed` values of the previous output row (free run) or the teacher's row t
      for (int i = tid; i < NB * p.feed; i += MNT) {
        const int b = i / p.feed, k = i - b * p.feed;
        float v = 0.f;
        if (b < B) v = p.tin ? p.tin[((int64_t)b * p.Td + t) * p.feed + k] : yv[b * MNO + (NO - 1 - p.feed) + k];
        va[b * MWN + k] = v;
      }
      WideW w1; SliceW sa; CellIn ci;
      wide_load(w1, p.Wp1, p.P1, p.P0, tid);
      cell_load<NB>(ci, p.ba, A, wg, p.ca, p.ha, par, B, tid);
      lds_barrier();
      MPROF(13);
      wide_compute<NB>(wn, va, MWN, p.feed, p.P0, p.bp0, SATT_ACT_RELU, nullptr, 0, vb, MWN, red, tid);             // (wn: requested at the end of the previous step)
      MPROF(14);
      slice_load(sa, p.Wa, 4 * A, 32 * wg, p.P1 + CT + A, tid);
      wide_compute<NB>(w1, vb, MWN, p.P0, p.P1, p.bp1, SATT_ACT_RELU, nullptr, 0, xs, MKS, red, tid);               // -> xs[b][0 .. P1)
      MPROF(15);
      for (int i = tid; i < NB * (CT + A); i += MNT) {
        const int b = i / (CT + A), k = i - b * (CT + A);
        float v = 0.f;
        if (b < B) v = k < CT ? ctx[b * MCT + k] : ald(p.ha + ((int64_t)par * B + b) * A + (k - CT));
        xs[b * MKS + p.P1 + k] = v;
      }
      lds_barrier();
      MPROF(16);
      slice_compute<NB>(sa, xs, p.P1 + CT + A, zs, red, tid);
      MPROF(17);
      lstm_cell<NB>(zs, ci, A, wg, p.ca, p.ha, par, B, p.zc, p.zh, p.hq, tid);
    }
    MPROF(0);
    bar_arrive(p.bar, target, dead, wg);
      wide_load(wn, p.Wq, UQ, A, tid);              // the query layer of phase B
    bar_wait(p.bar, target, p.err, dead);
    MPROF(1);
    // =========================================================== B: processed query + energies of the own rows
    {
      for (int i = tid; i < NB * A; i += MNT) { const int b = i / A, k = i - b * A; hq[b * MWN + k] = b < B ? ald(p.hq + (int64_t)b * A + k) : 0.f; }
      lds_barrier();
      wide_compute<NB>(wn, hq, MWN, A, UQ, nullptr, SATT_ACT_NONE, nullptr, 0, va, MWN, red, tid);                 // pq -> va
      MPROF(22);
      slice_load(sn, p.W1, 4 * D, 32 * wg, A + CT + D, tid);      // LSTM 1 of phase C: in flight across the energies and the barrier
      cell_load<NB>(cn, p.b1l, D, wg, p.c1, p.h1, par, B, tid);      // (these two are issued early in the phase: long done at the arrival)
      // (sample, own row) pairs over the waves; lane: 4 units of mechanism 1, unit `lane` of mechanism 2
      const int d0 = 4 * lane, dc = min(d0, U1 - 4);
      const bool ok1 = d0 < U1;
      float v1r[4], b1r[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { v1r[q] = ok1 ? tab[dc + q] : 0.f; b1r[q] = ok1 ? tab[MWN + dc + q] : 0.f; }
      const float v2r = (U2 && lane < U2) ? tab[2 * MWN + lane] : 0.f;
      for (int pr = wave; pr < B * R; pr += MNW) {
        const int b = pr / R, rr = pr - b * R, tt = r0 + rr;
        if (tt < Ti) {                      // (wave-uniform)
          const int len = lens[b];
          const float* kr = kls + (b * 8 + rr) * (MWN + 64);
          const float4 kk = *reinterpret_cast<const float4*>(kr + dc);
          const float k2 = U2 ? kr[MWN + min(lane, U2 - 1)] : 0.f;
          const float kq[4] = {kk.x, kk.y, kk.z, kk.w};
          // location features of row tt (conv1d SAME of the previous alignments, forward_attention.py:98-100): wave-uniform
          // (one (filter, tap) product per lane, F * KW <= 64, then one wave sum per filter: as nested loops of dependent LDS reads
          //  this was 2.5 us per step)
          float fl[8];
          {
            const int ff = lane / KW, jj = lane - ff * KW;
            const float term = (ff < F) ? aprev[b * (MTI + 16) + tt + jj] * Fs[jj * 8 + min(ff, 7)] : 0.f;
#pragma unroll
            for (int f = 0; f < 8; ++f) fl[f] = f < F ? wave_sum(ff == f ? term : 0.f) + Fs[16 * 8 + f] : 0.f;
          }
          float a = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float x = kq[q] + (ok1 ? b1r[q] + va[b * MWN + dc + q] : 0.f);
#pragma unroll
            for (int f = 0; f < 8; ++f) x += fl[f] * ((f < F && ok1) ? Us[min(f, F - 1) * MWN + dc + q] : 0.f);
            a += v1r[q] * tanhf_(x);
          }
          float a2 = v2r * tanhf_(k2 + (lane < U2 ? va[b * MWN + U1 + lane] : 0.f));
          a = wave_sum(a); a2 = wave_sum(a2);
          if (lane == 0 && tt < len) { ast(p.e1 + (int64_t)b * Ti + tt, a); if (U2) ast(p.e2 + (int64_t)b * Ti + tt, a2); }
        }
      }
    }
    MPROF(2);
    bar_arrive(p.bar, target, dead, wg);
    bar_wait(p.bar, target, p.err, dead);
    MPROF(3);
    // =========================================================== C: softmax, recursion, contexts (redundant) + LSTM 1
    {
      float4 cx0[8];                       // first batch of value rows of sample 0 (see the context block below)
      {
        const int ncg = CT / 4, ngr = MNT / ncg, cg = tid % ncg, rg = min(tid / ncg, ngr - 1), col = 4 * cg;
        const float* vs = col < V1 ? p.values1 + col : p.values2 + (col - V1);
        const int ld = col < V1 ? V1 : V2;
#pragma unroll
        for (int u = 0; u < 8; ++u) cx0[u] = *reinterpret_cast<const float4*>(vs + (int64_t)min(rg + ngr * u, Ti - 1) * ld);
      }
      for (int i = tid; i < NB * Ti; i += MNT) {
        const int b = i / Ti, r = i - b * Ti;
        const int len = lens[b];
        e1[b * MTI + r] = r < len ? ald(p.e1 + (int64_t)b * Ti + r) : -INFINITY;
        e2[b * MTI + r] = (U2 && r < len) ? ald(p.e2 + (int64_t)b * Ti + r) : -INFINITY;
      }
      lds_barrier();
      // one wave per (sample, mechanism): masked softmax; mechanism 1 continues with the forward recursion
      if (wave < 2 * NB) {
        const int b = wave >> 1, mech = wave & 1;
        if (b < B && (mech == 0 || U2)) {
          const int len = lens[b];
          float* e = (mech ? e2 : e1) + b * MTI;
          float m = -INFINITY;
          for (int i = lane; i < len; i += 64) m = fmaxf(m, e[i]);
          m = wave_max(m);
          float sacc = 0.f;
          for (int i = lane; i < Ti; i += 64) { const float x = i < len ? __expf(e[i] - m) : 0.f; e[i] = x; sacc += x; }
          sacc = wave_sum(sacc);
          const float rs = 1.f / sacc;
          if (mech == 1) {
            for (int i = lane; i < Ti; i += 64) e[i] *= rs;
          } else {
            float* ap = aprev + b * (MTI + 16) + PL;
            float* al = alpha + b * MTI;
            float sa = 0.f;
            float keep[MTI / 64];
#pragma unroll
            for (int q = 0; q < MTI / 64; ++q) {
              const int i = lane + 64 * q;
              float v = 0.f;
              if (i < Ti) {
                const float a = e[i] * rs;
                ap[i] = p.cumulative ? a + ap[i] : a;                 // next location-conv input
                v = a;
                if (p.att1_mode == 0) { v = (0.5f * al[i] + 0.5f * (i > 0 ? al[i - 1] : 0.f) + 1e-7f) * a; sa += v; }
              }
              keep[q] = v;
            }
            if (p.att1_mode == 0) {
              sa = wave_sum(sa);
              const float r2 = 1.f / sa;
#pragma unroll
              for (int q = 0; q < MTI / 64; ++q) keep[q] *= r2;
            }
            // (every lane has read its al[i - 1] above: the wave runs in lock step up to the reduction)
#pragma unroll
            for (int q = 0; q < MTI / 64; ++q) { const int i = lane + 64 * q; if (i < Ti) { al[i] = keep[q]; e[i] = keep[q]; } }
          }
        }
      }
      lds_barrier();
      MPROF(23);
      if (wg == 0) {       // histories and the global copies of the carried state
        for (int i = tid; i < NB * Ti; i += MNT) {
          const int b = i / Ti, r = i - b * Ti;
          if (b < B) {
            p.align1[((int64_t)b * p.Td + t) * Ti + r] = e1[b * MTI + r];
            if (p.align2) p.align2[((int64_t)b * p.Td + t) * Ti + r] = e2[b * MTI + r];
            p.a_state[((int64_t)(par ^ 1) * B + b) * Ti + r] = aprev[b * (MTI + 16) + PL + r];
            p.alpha_state[((int64_t)(par ^ 1) * B + b) * Ti + r] = e1[b * MTI + r];
          }
        }
      }
      // contexts: thread = (float4 column group of CT / 4, row group of MNT / (CT / 4)); the FIRST batch of value rows of sample 0
      // was requested before the softmax (cx0: the values do not depend on the alignments)
      {
        const int ncg = CT / 4, ngr = MNT / ncg, cg = tid % ncg, rg = tid / ncg;
        const int col = 4 * cg;
        const bool s1c = col < V1, act = rg < ngr;
        for (int b = 0; b < B; ++b) {
          const int len = lens[b];
          const float* vs = s1c ? p.values1 + (int64_t)b * Ti * V1 + col : p.values2 + (int64_t)b * Ti * V2 + (col - V1);
          const int ld = s1c ? V1 : V2;
          const float* al = (s1c ? e1 : e2) + b * MTI;
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int rb = rg; rb < len; rb += 8 * ngr) {
            float4 x[8];
            if (b == 0 && rb == rg) {
#pragma unroll
              for (int u = 0; u < 8; ++u) x[u] = cx0[u];
            } else {
#pragma unroll
              for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const float4*>(vs + (int64_t)min(rb + ngr * u, Ti - 1) * ld);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int r = rb + ngr * u;
              const float w = (act && r < len) ? al[min(r, Ti - 1)] : 0.f;
              acc.x += w * x[u].x; acc.y += w * x[u].y; acc.z += w * x[u].z; acc.w += w * x[u].w;
            }
          }
          if (act) *reinterpret_cast<float4*>(red + (rg * ncg + cg) * 4) = acc;
          lds_barrier();
          if (tid < CT) {
            float sacc = 0.f;
            for (int g = 0; g < ngr; ++g) sacc += red[(g * ncg + (tid >> 2)) * 4 + (tid & 3)];
            ctx[b * MCT + tid] = sacc;
            if (wg == 0) p.ctx[((int64_t)par * B + b) * CT + tid] = sacc;
          }
          lds_barrier();
        }
      }
      MPROF(24);
      // LSTM 1 on [h_att | context | h1]
      for (int i = tid; i < NB * (A + CT + D); i += MNT) {
        const int b = i / (A + CT + D), k = i - b * (A + CT + D);
        float v = 0.f;
        if (b < B) v = k < A ? hq[b * MWN + k] : (k < A + CT ? ctx[b * MCT + k - A] : ald(p.h1 + ((int64_t)par * B + b) * D + (k - A - CT)));
        xs[b * MKS + k] = v;
      }
      lds_barrier();
      slice_compute<NB>(sn, xs, A + CT + D, zs, red, tid);
      lstm_cell<NB>(zs, cn, D, wg, p.c1, p.h1, par, B, p.zc, p.zh, p.h1n, tid);
    }
    MPROF(4);
    bar_arrive(p.bar, target, dead, wg);
      slice_load(sn, p.W2, 4 * D, 32 * wg, 2 * D, tid);           // LSTM 2 of phase D
      cell_load<NB>(cn, p.b2l, D, wg, p.c2, p.h2, par, B, tid);
    bar_wait(p.bar, target, p.err, dead);
    MPROF(5);
    // =========================================================== D: LSTM 2 on [h1_new | h2]
    {
      for (int i = tid; i < NB * 2 * D; i += MNT) {
        const int b = i / (2 * D), k = i - b * 2 * D;
        float v = 0.f;
        if (b < B) v = k < D ? ald(p.h1n + (int64_t)b * D + k) : ald(p.h2 + ((int64_t)par * B + b) * D + (k - D));
        xs[b * MKS + k] = v;
      }
      lds_barrier();
      slice_compute<NB>(sn, xs, 2 * D, zs, red, tid);
      lstm_cell<NB>(zs, cn, D, wg, p.c2, p.h2, par, B, p.zc, p.zh, p.dout, tid);
    }
    MPROF(6);
    bar_arrive(p.bar, target, dead, wg);
      slice_load(sn, p.Wkvq, 3 * Ds, min(32 * wg, 3 * Ds - 32), D, tid);          // K | V | Q of phase E (clamped: idle workgroups read valid columns)
    bar_wait(p.bar, target, p.err, dead);
    MPROF(7);
    // =========================================================== E: K | V | Q row of the cache (own 32 columns)
    {
      for (int i = tid; i < NB * D; i += MNT) { const int b = i / D, k = i - b * D; const float v = b < B ? ald(p.dout + (int64_t)b * D + k) : 0.f; xs[b * MKS + k] = v; vb[b * MWN + k] = v; }
      lds_barrier();
      if (32 * wg < 3 * Ds) {                // (uniform per workgroup; its barriers are workgroup barriers)
        slice_compute<NB>(sn, xs, D, zs, red, tid);
        if (tid < NB * 32) {
          const int b = tid >> 5, n = 32 * wg + (tid & 31);
          if (b < B && n < 3 * Ds) ast(p.kvq + ((int64_t)b * p.Td + t) * 3 * Ds + n, zs[b * 32 + (tid & 31)] + p.bkvq[n]);
        }
      }
    }
    MPROF(8);
    bar_arrive(p.bar, target, dead, wg);
      wide_load(wn, p.Wot, Ds, Ds, tid);           // the folded output transform of phase G: in flight across two barriers
    bar_wait(p.bar, target, p.err, dead);
    MPROF(9);
    // =========================================================== F: causal self-attention, own (head, key chunk)
    {
      // One pass, every load in flight together: 16 lanes per key (8 dims each for head depth 128: two 16-byte loads), 32 keys per
      // pass over the workgroup - a chunk is at most ceil(Td / NCH) keys; then P V with thread = (4 dims, key group).
      const int h = wg % heads, ch = wg / heads;
      // chunks of >= 32 keys (one pass of the workgroup): early steps use few chunks - fewer partials for phase G to fold
      const int nk = t + 1, per = max(MNT / 16, (nk + NCH - 1) / NCH), j0 = ch * per, j1 = min(j0 + per, nk), nkc = max(j1 - j0, 0);
      const float scale = rsqrtf((float)hd);
      const int kg = tid >> 4, dl = tid & 15, dpl = hd / 16;            // key group / 16-lane dim slice (hd % 64 == 0: dpl % 4 == 0)
      for (int b = 0; b < B; ++b) {
        const float* base = p.kvq + (int64_t)b * p.Td * 3 * Ds + h * hd;
        float* sc = red;                     // [<= MTI] scores, then numerators
        float* part = red + MTI + 64;        // [MNT] partial P V
        // the new query row, the chunk's key rows and (first batch) value rows: ONE round trip - none of the addresses depends
        // on the scores
        const float qd = ald(base + (int64_t)t * 3 * Ds + 2 * Ds + min(tid, hd - 1));
        float4 kv[2];
        // (keys beyond the chunk are clamped to an OLD row where there is one: only the real row t takes the agent-scope loads)
        const int jsafe = min(j0, max(t - 1, 0));
        const int jk = j0 + kg, jkc = jk < j1 ? jk : jsafe;
        {
          const float* kp = base + (int64_t)jkc * 3 * Ds + dl * dpl;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if (jkc == t) kv[i] = make_float4(ald(kp + 4 * i), ald(kp + 4 * i + 1), ald(kp + 4 * i + 2), ald(kp + 4 * i + 3));     // this step's row: written by other workgroups
            else kv[i] = *reinterpret_cast<const float4*>(kp + 4 * i);
          }
        }
        const int nc4 = hd / 4, ng = MNT / nc4, c4 = tid % nc4, g = tid / nc4;
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int jv = j0 + g + ng * u, jc = jv < j1 ? jv : jsafe;
          const float* vp = base + (int64_t)jc * 3 * Ds + Ds + 4 * c4;
          if (jc == t) v[u] = make_float4(ald(vp), ald(vp + 1), ald(vp + 2), ald(vp + 3));
          else v[u] = *reinterpret_cast<const float4*>(vp);
        }
        if (tid < hd) va[tid] = qd;
        lds_barrier();
        {   // scores of the chunk's first MNT / 16 keys from the registers; later passes (chunks longer than one pass) load their rows
          for (int jb = 0; jb < nkc; jb += MNT / 16) {
            const int j = j0 + jb + kg, jc = j < j1 ? j : jsafe;
            float acc = 0.f;
            for (int i = 0; i < dpl; i += 4) {
              float4 kk;
              if (jb == 0 && i < 8) kk = kv[i >> 2];
              else {
                const float* kp = base + (int64_t)jc * 3 * Ds + dl * dpl + i;
                if (jc == t) kk = make_float4(ald(kp), ald(kp + 1), ald(kp + 2), ald(kp + 3));
                else kk = *reinterpret_cast<const float4*>(kp);
              }
              const float* qp = va + dl * dpl + i;
              acc += qp[0] * kk.x + qp[1] * kk.y + qp[2] * kk.z + qp[3] * kk.w;
            }
            SATT_DPP_ADD(acc, 0xB1); SATT_DPP_ADD(acc, 0x4E); SATT_DPP_ADD(acc, 0x141); SATT_DPP_ADD(acc, 0x140);   // 16-lane row sum
            if (dl == 0 && j < j1) sc[jb + kg] = acc * scale;
          }
        }
        lds_barrier();
        // softmax statistics of the chunk: one wave
        if (wave == 0) {
          float m = -INFINITY;
          for (int j = lane; j < nkc; j += 64) m = fmaxf(m, sc[j]);
          m = wave_max(m);
          float z = 0.f;
          for (int j = lane; j < nkc; j += 64) { const float e = __expf(sc[j] - m); sc[j] = e; z += e; }
          z = wave_sum(z);
          if (lane == 0) { sm[0] = nkc > 0 ? m : -INFINITY; sm[1] = nkc > 0 ? z : 0.f; }
        }
        lds_barrier();
        // partial P V: thread = (4 dims, key group of MNT / (hd / 4)); the first batch of value rows is in registers already
        {
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int jb = g; jb < nkc; jb += 4 * ng) {
            if (jb != g) {
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int jv = j0 + jb + ng * u, jc = jv < j1 ? jv : jsafe;
                const float* vp = base + (int64_t)jc * 3 * Ds + Ds + 4 * c4;
                if (jc == t) v[u] = make_float4(ald(vp), ald(vp + 1), ald(vp + 2), ald(vp + 3));
                else v[u] = *reinterpret_cast<const float4*>(vp);
              }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int jj = jb + ng * u;
              const float pj = jj < nkc ? sc[jj] : 0.f;
              acc.x += pj * v[u].x; acc.y += pj * v[u].y; acc.z += pj * v[u].z; acc.w += pj * v[u].w;
            }
          }
          *reinterpret_cast<float4*>(part + g * hd + 4 * c4) = acc;
          lds_barrier();
          float* dst = p.part + (((int64_t)b * heads + h) * NCH + ch) * (hd + 2);
          if (tid < hd) {
            float o = 0.f;
            for (int gg = 0; gg < ng; ++gg) o += part[gg * hd + tid];
            ast(dst + 2 + tid, o);
          }
          if (tid == 0) { ast(dst, sm[0]); ast(dst + 1, sm[1]); }
        }
        lds_barrier();
      }
    }
    MPROF(10);
    bar_arrive(p.bar, target, dead, wg);
    bar_wait(p.bar, target, p.err, dead);
    MPROF(11);
    // =========================================================== G: merge, output transform, projection (redundant)
    {
      // every thread folds its output column itself: the (max, sum) pairs and the partial values of ALL chunks are requested in
      // one go (one round trip); the rescaling factors are recomputed per thread (<= 16 exponentials) instead of travelling
      // through LDS behind a first round trip
      const int nkg = t + 1, perg = max(MNT / 16, (nkg + NCH - 1) / NCH), nch = min((nkg + perg - 1) / perg, 16);      // chunks phase F filled
      for (int i = tid; i < NB * Ds; i += MNT) {
        const int b = i / Ds, c = i - b * Ds, h = c / hd, d = c - h * hd;
        float o = 0.f;
        if (b < B) {
          const float* src = p.part + ((int64_t)b * heads + h) * NCH * (hd + 2);
          float mv[16], zv[16], ov[16];
#pragma unroll
          for (int c2 = 0; c2 < 16; ++c2) {
            const float* q = src + (int64_t)min(c2, nch - 1) * (hd + 2);
            mv[c2] = ald(q); zv[c2] = ald(q + 1); ov[c2] = ald(q + 2 + d);
          }
          float M = -INFINITY;
#pragma unroll
          for (int c2 = 0; c2 < 16; ++c2) M = fmaxf(M, c2 < nch ? mv[c2] : -INFINITY);
          float zt = 0.f;
#pragma unroll
          for (int c2 = 0; c2 < 16; ++c2) {
            const float f = (c2 < nch && zv[c2] > 0.f) ? __expf(mv[c2] - M) : 0.f;
            zt += f * zv[c2];
            o += f * ov[c2];
          }
          o /= zt;
        }
        va[b * MWN + c] = o;
      }
      MPROF(18);
      lds_barrier();
      MPROF(19);
      WideW wo;
      wide_load(wo, p.Wout, p.ldout, Ds, tid);
      wide_compute<NB>(wn, va, MWN, Ds, Ds, p.bot, SATT_ACT_TANH, vb, MWN, hq, MWN, red, tid);        // (hq is free until the next step's B)
      MPROF(20);
      wide_load(wn, p.Wp0, p.P0, p.feed, tid);     // pre-net 0 of the NEXT step
      wide_compute<NB>(wo, hq, MWN, Ds, NO, p.bout, SATT_ACT_NONE, nullptr, 0, yv, MNO, red, tid);
      MPROF(21);
      if (wg == 0) {
        for (int i = tid; i < NB * NO; i += MNT) { const int b = i / NO, c = i - b * NO; if (b < B) p.yout[((int64_t)b * (p.Td + 1) + t + 1) * NO + c] = yv[b * MNO + c]; }
        if (tid == 0) {
          // stop rule of StopTokenBasedInferenceHelper (csrc/decode.hip dec_bookkeeping evaluates it one launch later)
          if (p.flag && !p.tin) {
            bool all = true;
            for (int b = 0; b < B; ++b) all = all && (1.f / (1.f + __expf(-yv[b * MNO + NO - 1])) > p.stop_threshold);
            if (all && t > p.min_steps && *p.flag == 0) *p.flag = t + 1;
          }
          *p.step = t + 1; p.step[1] = t + 1;
        }
      }
      lds_barrier();
    }
    MPROF(12);
  }
  if (wg == 0 && tid == 0) *p.bar_base = target;
}

}  // namespace

#ifdef SATT_MEGA_PROF
extern "C" int satt_dec_mega_prof_read(unsigned long long* host16, int reset) {
  if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(satt_mega_prof), sizeof(unsigned long long) * 32) != hipSuccess) return -3;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(satt_mega_prof), z, sizeof(z)) != hipSuccess) return -3; }
  return 0;
}
#endif

// csrc/decode_mega2.hip: the register-resident, granule-exchange form of the same step (B <= 2)
int satt_dec_mega2_launch(const satt_dec_mega_params& p, hipStream_t s);
int64_t satt_dec_mega2_scratch_floats(int B, int heads);
bool satt_dec_mega2_takes(const satt_dec_mega_params& p);
static bool mega_first_form() { static const bool v = getenv("SATT_MEGA_V1") != nullptr; return v; }

extern "C" int64_t satt_dec_mega_scratch_floats(int B, int heads, int hd) {
  const int64_t v1 = (int64_t)B * heads * (MWG / (heads > 0 ? heads : 1)) * (hd + 2);
  const int64_t v2 = (hd > 0 && heads * hd == MWN) ? satt_dec_mega2_scratch_floats(B, heads) : 0;
  return v1 > v2 ? v1 : v2;
}

extern "C" int satt_dec_mega_supported(const satt_dec_mega_params* p) {
  if (!p) return 0;
  const int UQ = p->U1 + p->U2, CT = p->V1 + p->V2;
  return p->B >= 1 && p->B <= 4 && p->Ti >= 1 && p->Ti <= MTI && (p->Ti + MWG - 1) / MWG <= 8 && p->A == 8 * MWG && p->D == 8 * MWG && UQ <= MWN && p->U1 % 4 == 0 &&
         p->U2 <= 64 && CT <= MCT && CT % 4 == 0 && p->V1 % 4 == 0 && p->V2 % 4 == 0 && p->V2 > 0 && p->U2 > 0 && p->P0 <= MWN && p->P1 <= MWN &&
         p->P0 % 8 == 0 && p->P1 % 8 == 0 && UQ % 8 == 0 && p->ldout % 8 == 0 && p->feed <= MWN && p->feed + 1 <= p->NO && p->NO <= MNO && p->ldout % 4 == 0 && p->ldout >= p->NO &&
         p->Ds == MWN && p->heads >= 2 && MWG % p->heads == 0 && p->Ds % p->heads == 0 && (p->Ds / p->heads) <= MNT &&
         MNT % (p->Ds / p->heads) == 0 && 3 * p->Ds <= 32 * MWG && p->P1 + CT + p->A <= MKS && p->A + CT + p->D <= MKS &&
         p->kernel >= 1 && p->kernel <= 16 && p->filters >= 1 && p->filters <= 8 && p->kernel * p->filters <= 64 && p->Td >= 1 &&
         mega_lds_bytes(p->B <= 1 ? 1 : (p->B <= 2 ? 2 : 4)) <= 160 * 1024;
}

extern "C" int satt_dec_mega(const satt_dec_mega_params* pp, void* stream) {
  if (!pp || !satt_dec_mega_supported(pp) || pp->nsteps < 1) return SATT_E_UNSUPPORTED;
  const satt_dec_mega_params& p = *pp;
  if (!p.Wp0 || !p.Wp1 || !p.Wa || !p.Wq || !p.W1 || !p.W2 || !p.Wkvq || !p.Wot || !p.Wout || !p.bar || !p.err || !p.step || !p.part ||
      !p.bar_base) return SATT_E_BADARG;
  const int NB = p.B <= 1 ? 1 : (p.B <= 2 ? 2 : 4);
  const size_t smem = mega_lds_bytes(NB);
  if (smem > 160 * 1024) return SATT_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (!mega_first_form() && satt_dec_mega2_takes(p)) return satt_dec_mega2_launch(p, s);
#define SATT_MEGA(NBV)                                                                                                     \
  do {                                                                                                                       \
    (void)hipFuncSetAttribute((const void*)dec_mega_k<NBV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);         \
    hipLaunchKernelGGL(dec_mega_k<NBV>, dim3(MEGA_GRID), dim3(MNT), smem, s, p);                                             \
  } while (0)
  if (NB == 1) SATT_MEGA(1); else if (NB == 2) SATT_MEGA(2); else SATT_MEGA(4);
#undef SATT_MEGA
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}
