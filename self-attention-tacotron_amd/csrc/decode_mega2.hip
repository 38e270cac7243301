// Persistent decode step, second form (r5): the training recipe applied to csrc/decode_mega.hip.
//   * EVERY weight is register resident for the whole launch: the sliced products (attention LSTM, LSTM 1, LSTM 2, K|V|Q: 32 columns
//     per workgroup) hold their [K x 32] bf16 slice in 76 registers per thread, and the small layers (pre-net 0 / 1, query layer,
//     folded output transform, mel | stop projection) are SPLIT 32 ways - 8 columns per workgroup, one 16-byte register per thread -
//     instead of being recomputed by every workgroup from a 128 KB weight stream;
//   * exchanges are {tag, value} granules (cluster_xchg.h: the data is the flag): a consumer polls the vector it needs, nobody waits
//     for a write-through acknowledgement and there is no separate barrier.  Eleven exchanges per step:
//       p0 -> p1 -> [attention LSTM] hq -> pq -> [energies] e1 | e2 -> [softmax, contexts, LSTM 1] h1n -> [LSTM 2] dout ->
//       [K|V|Q] row t -> [cached self-attention partials] -> [merge, output transform] tr -> [projection] y
// Same buffers, same math, same selection as the first form (satt_dec_mega_supported); granule tags are step + 1, the caller zeroes
// the granule buffer when it resets the step counter.  Single-buffered granules are safe: between the consumption of X(t) and the
// production of X(t+1) lies at least one exchange every workgroup contributes to (hq, pq, h1n, dout, tr).
#include "cluster_xchg.h"

#ifdef SATT_MEGA_PROF      // per-phase wall-clock sums (100 MHz) of workgroup 0: tools/build_variant.sh + tools/decode_mega_prof.py
static __device__ unsigned long long satt_mega2_prof[32];
#define MPROF(i) do { if (wg == 0 && threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); satt_mega2_prof[i] += n_ - mp_last; mp_last = n_; } } while (0)
#else
#define MPROF(i)
#endif

namespace {

constexpr int M2T = 512;                         // threads (8 waves: cluster_xchg.h's gathers are written for XW = 8)
constexpr int M2G = 32;                          // persistent workgroups
constexpr int M2N = 256;                         // width of the split layers / of A, D, Ds
constexpr int M2K = 1024;                        // largest K of a sliced product
// Publishing and polling are kept in DIFFERENT waves: the vector-memory counter is in order and counts stores, so a wave that has just
// published (agent-scope store: acknowledged only after the write-through) would wait for that acknowledgement in its first poll.
constexpr int PUTW = 7, GATW = 6;                // the wave that publishes; the waves [0, GATW) that poll
constexpr int M2TI = 256, M2CT = 320, M2NO = 168;
constexpr int M2RED = 4352;                      // floats of the reduction / phase scratch (the gathered partials: 32 x (hd + 2))
typedef __attribute__((address_space(1))) float gf32q;
__device__ __forceinline__ void ast2(float* p, float v) { __hip_atomic_store((gf32q*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ald2(const float* p) { return __hip_atomic_load((const gf32q*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void unpack8q(uint4 v, float (&w)[8]) {
  w[0] = __uint_as_float(v.x << 16); w[1] = __uint_as_float(v.x & 0xFFFF0000u); w[2] = __uint_as_float(v.y << 16); w[3] = __uint_as_float(v.y & 0xFFFF0000u);
  w[4] = __uint_as_float(v.z << 16); w[5] = __uint_as_float(v.z & 0xFFFF0000u); w[6] = __uint_as_float(v.w << 16); w[7] = __uint_as_float(v.w & 0xFFFF0000u);
}

// granule layout (u64 words) per sample
struct GL { int p0, p1, hq, pq, e, h1, dout, kvq, part, tr, y, total; };
__host__ __device__ inline GL gl_of(int Ti, int Ds, int heads, int hd) {
  GL g; int o = 0;
  g.p0 = o; o += M2N; g.p1 = o; o += M2N; g.hq = o; o += M2N; g.pq = o; o += M2N; g.e = o; o += 2 * M2TI; g.h1 = o; o += M2N; g.dout = o; o += M2N;
  g.kvq = o; o += 3 * Ds; g.part = o; o += M2G * (hd + 2); g.tr = o; o += M2N; g.y = o; o += M2NO;
  g.total = o;
  (void)Ti; (void)heads;
  return g;
}

// ---- resident slice of a sliced product: thread = (column group tid & 3 of 8 columns, k lane tid >> 2 of 128): rows kl + 128 i
template <int KI> struct SliceR { uint4 v[KI]; };
// The packed registers are made opaque once per step: otherwise the compiler hoists the bf16 -> fp32 unpacking out of the step loop,
// which doubles the resident footprint and spills it.
__device__ __forceinline__ void pin(uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
template <int KI> __device__ __forceinline__ void pin(SliceR<KI>& w) {
#pragma unroll
  for (int i = 0; i < KI; ++i) pin(w.v[i]);
}
template <int KI>
__device__ __forceinline__ void slice_fill(SliceR<KI>& w, const uint16_t* __restrict__ W, int ldw, int n0, int K, int tid) {
  const int cg = tid & 3, kl = tid >> 2;
#pragma unroll
  for (int i = 0; i < KI; ++i) {
    const int k = kl + 128 * i;
    uint4 v = *reinterpret_cast<const uint4*>(W + (int64_t)min(k, K - 1) * ldw + n0 + 8 * cg);
    if (k >= K) v = make_uint4(0u, 0u, 0u, 0u);
    w.v[i] = v;
  }
}
// z[b][32] = the workgroup's 32 columns of xs[b][0..K) W
template <int NB, int KI>
__device__ __forceinline__ void slice_mul(const SliceR<KI>& wv, const float* xs, int K, float* z, float* red, int tid) {
  const int cg = tid & 3, kl = tid >> 2, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float acc[NB][8];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float xv[KI];
#pragma unroll
    for (int i = 0; i < KI; ++i) xv[i] = xs[b * M2K + min(kl + 128 * i, K - 1)];      // (rows beyond K carry zero weights)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[b][j] = 0.f;
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      float w[8];
      unpack8q(wv.v[i], w);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[b][j] += xv[i] * w[j];
    }
  }
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int j = 0; j < 8; ++j) {      // the 16 k lanes of a wave (lanes cg + 4 q)
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
    for (int w = 0; w < XW; ++w) s += red[(w * NB + b) * 32 + n];
    z[b * 32 + n] = s;
  }
  lds_barrier();
}

// ---- split layer: the workgroup's 8 columns [8 wg, 8 wg + 8) of act(x W + bias) (+ res); thread k < K holds row k of them.
// The 8 results of sample b are published as granules dst[b * bs + 8 wg + j] (columns >= N are not published).  bias8: the 8 bias
// values (an LDS table filled once per launch: a global load here would sit on the step's dependency chain).  `red` is free again
// after the caller's next workgroup barrier (every use is followed by the gather of the published vector).
template <int NB>
__device__ __forceinline__ void split_mul(uint4 wr, const float* x, int xs_, int K, int N, const float* bias8, int act,
                                          const float* res, int rs_, u64* dst, int64_t bs, uint32_t tag, int wg, int B, float* red, int tid) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float w[8];
  unpack8q(wr, w);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const float xv = tid < K ? x[b * xs_ + tid] : 0.f;
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = xv * w[j];
    wave_sum_multi<8>(a);
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[(b * XW + wave) * 8 + j] = a[j];
    }
  }
  lds_barrier();
  if (wave == PUTW && lane < NB * 8) {
    const int b = lane >> 3, j = lane & 7, n = 8 * wg + j;
    if (b < B && n < N) {
      float s = 0.f;
#pragma unroll
      for (int wv = 0; wv < XW; ++wv) s += red[(b * XW + wv) * 8 + j];
      s += bias8[j];
      if (act == SATT_ACT_RELU) s = fmaxf(s, 0.f);
      else if (act == SATT_ACT_TANH) s = tanhf_(s);
      if (res) s += res[b * rs_ + n];
      gput(dst + b * bs + n, tag, s, false);
    }
  }
}
__device__ __forceinline__ uint4 split_fill(const uint16_t* __restrict__ W, int ldw, int K, int wg, int tid) {
  const int c0 = min(8 * wg, max(ldw - 8, 0));
  uint4 v = *reinterpret_cast<const uint4*>(W + (int64_t)min(tid, K - 1) * ldw + c0);
  if (tid >= K || 8 * wg >= ldw) v = make_uint4(0u, 0u, 0u, 0u);
  return v;
}

// waves [0, GATW) gather n granules of every sample into LDS rows dst[b * ds_ + i]
template <int NB>
__device__ __forceinline__ void gather_rows(u64* src, int64_t bs, int n, uint32_t tag, float* dst, int ds_, int B, int tid, unsigned int* err, int* dead) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int b = 0; b < B; ++b)
    if (wave < GATW) gather_span(src + b * bs, n, tag, wave, GATW, lane, [&](int i, float v) { dst[b * ds_ + i] = v; }, err, dead);
  lds_barrier();
}

template <int NB>
__global__ __launch_bounds__(M2T) void dec_mega2_k(const satt_dec_mega_params p) {
  const int wg = blockIdx.x;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* red = smem;                                 // [M2RED] reductions / phase scratch
  float* xs = red + M2RED;                           // [NB][1024]
  float* yv = xs + NB * M2K;                         // [NB][168]
  float* ctx = yv + NB * M2NO;                       // [NB][320]
  float* aprev = ctx + NB * M2CT;                    // [NB][256 + 16]
  float* alpha = aprev + NB * (M2TI + 16);           // [NB][256]
  float* hq = alpha + NB * M2TI;                     // [NB][256]
  float* va = hq + NB * M2N;                         // [NB][256] scratch vectors
  float* vb = va + NB * M2N;
  float* vc = vb + NB * M2N;
  float* e1 = vc + NB * M2N;                         // [NB][256]
  float* e2 = e1 + NB * M2TI;
  float* zs = e2 + NB * M2TI;                        // [NB][32]
  float* sm = zs + NB * 32;                          // [64]
  int* dead = reinterpret_cast<int*>(sm + 64);
  float* Us = reinterpret_cast<float*>(dead + 4);    // [8][256]
  float* Fs = Us + 8 * M2N;                          // [17][8]
  float* tab = Fs + 17 * 8;                          // [3][256]: v1 | b1 | v2
  float* kls = tab + 3 * M2N;                        // [NB * 8][256 + 64]
  float* bt = kls + NB * 8 * (M2N + 64);             // [5][8] split-layer biases | [3][32] cell biases (gate-major) | [32] K|V|Q bias
  int* lens = reinterpret_cast<int*>(bt + 40 + 96 + 32);
  const int B = p.B, Ti = p.Ti, A = p.A, D = p.D, Ds = p.Ds, U1 = p.U1, U2 = p.U2, UQ = U1 + U2, V1 = p.V1, V2 = p.V2, CT = V1 + V2;
  const int NO = p.NO, KW = p.kernel, F = p.filters, PL = (KW - 1) / 2, heads = p.heads, hd = Ds / heads;
  const GL G = gl_of(Ti, Ds, heads, hd);
  u64* gr = reinterpret_cast<u64*>(p.part);
  const int64_t gbs = G.total;                       // granules per sample
  {
    const int tid = threadIdx.x;
    if (tid == 0) *dead = 0;
    for (int i = tid; i < 8 * M2N; i += M2T) { const int f = i / M2N, u = i - f * M2N; Us[i] = (f < F && u < U1) ? p.locU[f * U1 + u] : 0.f; }
    for (int i = tid; i < 3 * M2N; i += M2T) {
      const int w = i / M2N, u = i - w * M2N;
      tab[i] = w == 0 ? (u < U1 ? p.v1[u] : 0.f) : (w == 1 ? (u < U1 ? p.b1[u] : 0.f) : (u < U2 ? p.v2[u] : 0.f));
    }
    if (tid < NB) lens[tid] = tid < B ? (int)p.lengths[tid] : 0;
    const int Rk = (Ti + M2G - 1) / M2G;
    for (int i = tid; i < NB * 8 * (M2N + 64); i += M2T) {
      const int row = i / (M2N + 64), u = i - row * (M2N + 64), b = row / 8, rr = row - b * 8, tt = wg * Rk + rr;
      float v = 0.f;
      if (b < B && rr < Rk && tt < Ti) v = u < M2N ? (u < U1 ? p.keys1[((int64_t)b * Ti + tt) * U1 + u] : 0.f) : (u - M2N < U2 ? p.keys2[((int64_t)b * Ti + tt) * U2 + u - M2N] : 0.f);
      kls[i] = v;
    }
    for (int i = tid; i < 17 * 8; i += M2T) { const int j = i >> 3, f = i & 7; Fs[i] = f < F ? (j < KW ? p.locF[j * F + f] : (j == 16 ? p.locFb[f] : 0.f)) : 0.f; }
    for (int i = tid; i < NB * (M2TI + 16); i += M2T) aprev[i] = 0.f;
    if (tid < 40) {
      const int l = tid >> 3, n = 8 * wg + (tid & 7);
      const float* bp = l == 0 ? p.bp0 : (l == 1 ? p.bp1 : (l == 2 ? nullptr : (l == 3 ? p.bot : p.bout)));
      const int N = l == 0 ? p.P0 : (l == 1 ? p.P1 : (l == 2 ? 0 : (l == 3 ? p.Ds : p.NO)));
      bt[tid] = (bp && n < N) ? bp[n] : 0.f;
    } else if (tid >= 64 && tid < 64 + 96) {
      const int i = tid - 64, l = i >> 5, g = (i >> 3) & 3, u = i & 7;
      const float* bp = l == 0 ? p.ba : (l == 1 ? p.b1l : p.b2l);
      bt[40 + i] = bp[g * (l == 0 ? p.A : p.D) + 8 * wg + u];
    } else if (tid >= 192 && tid < 224) {
      const int n = 32 * wg + tid - 192;
      bt[136 + tid - 192] = n < 3 * p.Ds ? p.bkvq[n] : 0.f;
    }
  }
  __syncthreads();
  int t = *p.step;
  int stopped = (p.flag && threadIdx.x == 0 && blockIdx.x == 0) ? *p.flag : 0;      // (only workgroup 0 / thread 0 uses it)
#ifdef SATT_MEGA_PROF
  unsigned long long mp_last = wall_clock64(), mp_clk = clock64();
#endif
  {
    const int tid = threadIdx.x, par = t & 1;
    for (int i = tid; i < NB * Ti; i += M2T) {
      const int b = i / Ti, r = i - b * Ti;
      if (b < B) {
        aprev[b * (M2TI + 16) + PL + r] = p.a_state[((int64_t)par * B + b) * Ti + r];
        alpha[b * M2TI + r] = p.alpha_state[((int64_t)par * B + b) * Ti + r];
      }
    }
    for (int i = tid; i < NB * CT; i += M2T) { const int b = i / CT, c = i - b * CT; ctx[b * M2CT + c] = b < B ? p.ctx[((int64_t)(par ^ 1) * B + b) * CT + c] : 0.f; }
    for (int i = tid; i < NB * NO; i += M2T) { const int b = i / NO, c = i - b * NO; yv[b * M2NO + c] = b < B ? p.yout[((int64_t)b * (p.Td + 1) + t) * NO + c] : 0.f; }
  }
  // ---- resident weights (registers for the whole launch)
  SliceR<6> sa; SliceR<7> s1; SliceR<4> s2; SliceR<2> sk;
  slice_fill(sa, p.Wa, 4 * A, 32 * wg, p.P1 + CT + A, (int)threadIdx.x);
  slice_fill(s1, p.W1, 4 * D, 32 * wg, A + CT + D, (int)threadIdx.x);
  slice_fill(s2, p.W2, 4 * D, 32 * wg, 2 * D, (int)threadIdx.x);
  slice_fill(sk, p.Wkvq, 3 * Ds, min(32 * wg, 3 * Ds - 32), D, (int)threadIdx.x);
  uint4 wp0 = split_fill(p.Wp0, p.P0, p.feed, wg, (int)threadIdx.x), wp1 = split_fill(p.Wp1, p.P1, p.P0, wg, (int)threadIdx.x);
  uint4 wqr = split_fill(p.Wq, UQ, A, wg, (int)threadIdx.x), wot = split_fill(p.Wot, Ds, Ds, wg, (int)threadIdx.x);
  uint4 wou = split_fill(p.Wout, p.ldout, Ds, wg, (int)threadIdx.x);
  __syncthreads();
  const int R = (Ti + M2G - 1) / M2G, r0 = wg * R;
  const int NCH = M2G / heads;
  const int nsteps = p.nsteps;
  for (int s = 0; s < nsteps; ++s, ++t) {
    typedef const __attribute__((address_space(4))) satt_dec_mega_params KArgsM;
    KArgsM* kq = (KArgsM*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kq));
    const auto& p = *kq;
    int oz = 0;
    asm volatile("" : "+v"(oz));
    pin(sa); pin(s1); pin(s2); pin(sk); pin(wp0); pin(wp1); pin(wqr); pin(wot); pin(wou);
    const int tid = (int)threadIdx.x + oz, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(tid >= 0 && tid < M2T && wave >= 0 && wave < XW);
    const int par = t & 1;
    const uint32_t tag = (uint32_t)(t + 1);
    unsigned int* err = p.err;
    // ================= A1: pre-net 0 (split) on the fed frame
    const float* fed = yv + (NO - 1 - p.feed);      // free running: the frame this workgroup gathered at the end of the previous step
    int fstr = M2NO;
    if (p.tin) {
      for (int i = tid; i < NB * p.feed; i += M2T) {
        const int b = i / p.feed, k = i - b * p.feed;
        va[b * M2N + k] = b < B ? p.tin[((int64_t)b * p.Td + t) * p.feed + k] : 0.f;
      }
      lds_barrier();
      fed = va; fstr = M2N;
    }
    MPROF(0);
    split_mul<NB>(wp0, fed, fstr, p.feed, p.P0, bt, SATT_ACT_RELU, nullptr, 0, gr + G.p0, gbs, tag, wg, B, red, tid);
    MPROF(1);
    gather_rows<NB>(gr + G.p0, gbs, p.P0, tag, vb, M2N, B, tid, err, dead);
    MPROF(2);
    // ================= A2: pre-net 1 (split)
    split_mul<NB>(wp1, vb, M2N, p.P0, p.P1, bt + 8, SATT_ACT_RELU, nullptr, 0, gr + G.p1, gbs, tag, wg, B, red, tid);
    MPROF(3);
    // own-unit state of the attention cell + [context | h] of the input row: requested while the pre-net exchange runs
    float c_old = 0.f, h_old = 0.f, b4[4] = {0.f, 0.f, 0.f, 0.f};
    {
      const int b = min(tid >> 3, B - 1), eu = 8 * wg + (tid & 7);
      const int64_t oi = (int64_t)par * B * A + (int64_t)b * A + eu;
      c_old = ald2(p.ca + oi); h_old = ald2(p.ha + oi);
#pragma unroll
      for (int g = 0; g < 4; ++g) b4[g] = bt[40 + g * 8 + (tid & 7)];
    }
    for (int i = tid; i < NB * (CT + A); i += M2T) {
      const int b = i / (CT + A), k = i - b * (CT + A);
      float v = 0.f;
      if (b < B) v = k < CT ? ctx[b * M2CT + k] : ald2(p.ha + ((int64_t)par * B + b) * A + (k - CT));
      xs[b * M2K + p.P1 + k] = v;
    }
    gather_rows<NB>(gr + G.p1, gbs, p.P1, tag, xs, M2K, B, tid, err, dead);          // -> xs[b][0 .. P1)
    MPROF(4);
    // ================= A3: attention LSTM slice + cell
    slice_mul<NB, 6>(sa, xs, p.P1 + CT + A, zs, red, tid);
    if (tid < NB * 8) {
      const int b = tid >> 3, u = tid & 7, eu = 8 * wg + u;
      if (b < B) {
        const float* zb = zs + b * 32;
        const float zi = zb[u] + b4[0], zj = zb[8 + u] + b4[1], zf = zb[16 + u] + b4[2], zo = zb[24 + u] + b4[3];
        const int64_t oo = (int64_t)(par ^ 1) * B * A + (int64_t)b * A + eu;
        const float cn = sigmoidf_(zf + 1.f) * c_old + sigmoidf_(zi) * tanhf_(zj);
        const float hn = sigmoidf_(zo) * tanhf_(cn);
        gput(gr + b * gbs + G.hq + eu, tag, hn, false);
        ast2(p.ca + oo, (1.f - p.zc) * cn + p.zc * c_old);
        ast2(p.ha + oo, (1.f - p.zh) * hn + p.zh * h_old);
      }
    }
    MPROF(5);
    gather_rows<NB>(gr + G.hq, gbs, A, tag, hq, M2N, B, tid, err, dead);
    MPROF(6);
    // ================= B1: query layer (split)
    split_mul<NB>(wqr, hq, M2N, A, UQ, bt + 16, SATT_ACT_NONE, nullptr, 0, gr + G.pq, gbs, tag, wg, B, red, tid);
    MPROF(7);
    // LSTM 1's own-unit state: requested now
    float c1o = 0.f, h1o = 0.f, b41[4] = {0.f, 0.f, 0.f, 0.f};
    {
      const int b = min(tid >> 3, B - 1), eu = 8 * wg + (tid & 7);
      const int64_t oi = (int64_t)par * B * D + (int64_t)b * D + eu;
      c1o = ald2(p.c1 + oi); h1o = ald2(p.h1 + oi);
#pragma unroll
      for (int g = 0; g < 4; ++g) b41[g] = bt[72 + g * 8 + (tid & 7)];
    }
    gather_rows<NB>(gr + G.pq, gbs, UQ, tag, va, M2N, B, tid, err, dead);             // pq -> va
    MPROF(8);
    // ================= B2: energies of the own rows
    {
      // lane l handles units l + 64 q of mechanism 1 (stride-1 LDS reads; q beyond U1 is skipped wave-uniformly) and unit l of mechanism 2
      const float v2r = (U2 && lane < U2) ? tab[2 * M2N + lane] : 0.f;
      for (int pr = wave; pr < B * R; pr += XW) {
        const int b = pr / R, rr = pr - b * R, tt = r0 + rr;
        if (tt < Ti) {
          const float* kr = kls + (b * 8 + rr) * (M2N + 64);
          // location features: lane = (filter l & 7, tap group l >> 3): taps jj = group, group + 8; xor-reduced over the groups
          float fl[8];
          {
            const int f = lane & 7, jg = lane >> 3;
            const float* ap = aprev + b * (M2TI + 16) + tt;
            float part = ap[jg] * Fs[jg * 8 + f] + ap[jg + 8] * Fs[(jg + 8) * 8 + f];      // (Fs rows >= kernel are zero)
            part += swz_xor(part, 8); part += swz_xor(part, 16);
            part += __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(part)));
#pragma unroll
            for (int ff = 0; ff < 8; ++ff) fl[ff] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(part), ff)) + Fs[16 * 8 + ff];
          }
          float a = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (64 * q < U1) {
              const int d = lane + 64 * q;
              float x = kr[d] + tab[M2N + d] + va[b * M2N + min(d, UQ - 1)];
#pragma unroll
              for (int ff = 0; ff < 8; ++ff) x += fl[ff] * Us[ff * M2N + d];
              a += tab[d] * tanhf_(x);          // (v1 is zero beyond U1)
            }
          }
          const float k2 = U2 ? kr[M2N + min(lane, U2 - 1)] : 0.f;
          float a2 = v2r * tanhf_(k2 + (lane < U2 ? va[b * M2N + U1 + lane] : 0.f));
          a = wave_sum(a); a2 = wave_sum(a2);
          // (rows beyond the sample's length are published too: their consumers mask them - every granule of [0, Ti) gets its tag)
          if (lane == 0) { gput(gr + b * gbs + G.e + tt, tag, a, false); gput(gr + b * gbs + G.e + M2TI + tt, tag, a2, false); }
        }
      }
    }
    MPROF(9);
    for (int b = 0; b < B; ++b) {
      if (wave < 4) gather_span(gr + b * gbs + G.e, Ti, tag, wave, 4, lane, [&](int i, float v) { e1[b * M2TI + i] = i < lens[b] ? v : -INFINITY; }, err, dead);
      else gather_span(gr + b * gbs + G.e + M2TI, Ti, tag, wave - 4, 4, lane, [&](int i, float v) { e2[b * M2TI + i] = (U2 && i < lens[b]) ? v : -INFINITY; }, err, dead);
    }
    lds_barrier();
    MPROF(10);
    // ================= C: softmax, recursion, contexts (redundant) + LSTM 1
    if (wave < 2 * NB) {
      const int b = wave >> 1, mech = wave & 1;
      if (b < B && (mech == 0 || U2)) {
        const int len = lens[b];
        float* e = (mech ? e2 : e1) + b * M2TI;
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
          float* ap = aprev + b * (M2TI + 16) + PL;
          float* al = alpha + b * M2TI;
          float sa_ = 0.f;
          float keep[M2TI / 64];
#pragma unroll
          for (int q = 0; q < M2TI / 64; ++q) {
            const int i = lane + 64 * q;
            float v = 0.f;
            if (i < Ti) {
              const float a = e[i] * rs;
              ap[i] = p.cumulative ? a + ap[i] : a;
              v = a;
              if (p.att1_mode == 0) { v = (0.5f * al[i] + 0.5f * (i > 0 ? al[i - 1] : 0.f) + 1e-7f) * a; sa_ += v; }
            }
            keep[q] = v;
          }
          if (p.att1_mode == 0) {
            sa_ = wave_sum(sa_);
            const float r2 = 1.f / sa_;
#pragma unroll
            for (int q = 0; q < M2TI / 64; ++q) keep[q] *= r2;
          }
#pragma unroll
          for (int q = 0; q < M2TI / 64; ++q) { const int i = lane + 64 * q; if (i < Ti) { al[i] = keep[q]; e[i] = keep[q]; } }
        }
      }
    }
    lds_barrier();
    MPROF(11);
    if (wg == 0) {
      for (int i = tid; i < NB * Ti; i += M2T) {
        const int b = i / Ti, r = i - b * Ti;
        if (b < B) {
          p.align1[((int64_t)b * p.Td + t) * Ti + r] = e1[b * M2TI + r];
          if (p.align2) p.align2[((int64_t)b * p.Td + t) * Ti + r] = e2[b * M2TI + r];
          p.a_state[((int64_t)(par ^ 1) * B + b) * Ti + r] = aprev[b * (M2TI + 16) + PL + r];
          p.alpha_state[((int64_t)(par ^ 1) * B + b) * Ti + r] = e1[b * M2TI + r];
        }
      }
    }
    {
      const int ncg = CT / 4, ngr = M2T / ncg, cg = tid % ncg, rg = tid / ncg;
      const int col = 4 * cg;
      const bool s1c = col < V1, act = rg < ngr;
      for (int b = 0; b < B; ++b) {
        const int len = lens[b];
        const float* vs = s1c ? p.values1 + (int64_t)b * Ti * V1 + col : p.values2 + (int64_t)b * Ti * V2 + (col - V1);
        const int ld = s1c ? V1 : V2;
        const float* al = (s1c ? e1 : e2) + b * M2TI;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int rb = rg; rb < len; rb += 8 * ngr) {
          float4 x[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const float4*>(vs + (int64_t)min(rb + ngr * u, Ti - 1) * ld);
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
          ctx[b * M2CT + tid] = sacc;
          if (wg == 0) p.ctx[((int64_t)par * B + b) * CT + tid] = sacc;
        }
        lds_barrier();
      }
    }
    MPROF(12);
    for (int i = tid; i < NB * (A + CT + D); i += M2T) {
      const int b = i / (A + CT + D), k = i - b * (A + CT + D);
      float v = 0.f;
      if (b < B) v = k < A ? hq[b * M2N + k] : (k < A + CT ? ctx[b * M2CT + k - A] : ald2(p.h1 + ((int64_t)par * B + b) * D + (k - A - CT)));
      xs[b * M2K + k] = v;
    }
    lds_barrier();
    MPROF(13);
    slice_mul<NB, 7>(s1, xs, A + CT + D, zs, red, tid);
    if (tid < NB * 8) {
      const int b = tid >> 3, u = tid & 7, eu = 8 * wg + u;
      if (b < B) {
        const float* zb = zs + b * 32;
        const float zi = zb[u] + b41[0], zj = zb[8 + u] + b41[1], zf = zb[16 + u] + b41[2], zo = zb[24 + u] + b41[3];
        const int64_t oo = (int64_t)(par ^ 1) * B * D + (int64_t)b * D + eu;
        const float cn = sigmoidf_(zf + 1.f) * c1o + sigmoidf_(zi) * tanhf_(zj);
        const float hn = sigmoidf_(zo) * tanhf_(cn);
        gput(gr + b * gbs + G.h1 + eu, tag, hn, false);
        ast2(p.c1 + oo, (1.f - p.zc) * cn + p.zc * c1o);
        ast2(p.h1 + oo, (1.f - p.zh) * hn + p.zh * h1o);
      }
    }
    MPROF(14);
    // ================= D: LSTM 2 on [h1_new | h2]
    float c2o = 0.f, h2o = 0.f, b42[4] = {0.f, 0.f, 0.f, 0.f};
    {
      const int b = min(tid >> 3, B - 1), eu = 8 * wg + (tid & 7);
      const int64_t oi = (int64_t)par * B * D + (int64_t)b * D + eu;
      c2o = ald2(p.c2 + oi); h2o = ald2(p.h2 + oi);
#pragma unroll
      for (int g = 0; g < 4; ++g) b42[g] = bt[104 + g * 8 + (tid & 7)];
    }
    for (int i = tid; i < NB * D; i += M2T) { const int b = i / D, k = i - b * D; xs[b * M2K + D + k] = b < B ? ald2(p.h2 + ((int64_t)par * B + b) * D + k) : 0.f; }
    gather_rows<NB>(gr + G.h1, gbs, D, tag, xs, M2K, B, tid, err, dead);               // h1_new -> xs[b][0 .. D)
    MPROF(15);
    slice_mul<NB, 4>(s2, xs, 2 * D, zs, red, tid);
    if (tid < NB * 8) {
      const int b = tid >> 3, u = tid & 7, eu = 8 * wg + u;
      if (b < B) {
        const float* zb = zs + b * 32;
        const float zi = zb[u] + b42[0], zj = zb[8 + u] + b42[1], zf = zb[16 + u] + b42[2], zo = zb[24 + u] + b42[3];
        const int64_t oo = (int64_t)(par ^ 1) * B * D + (int64_t)b * D + eu;
        const float cn = sigmoidf_(zf + 1.f) * c2o + sigmoidf_(zi) * tanhf_(zj);
        const float hn = sigmoidf_(zo) * tanhf_(cn);
        gput(gr + b * gbs + G.dout + eu, tag, hn, false);
        ast2(p.c2 + oo, (1.f - p.zc) * cn + p.zc * c2o);
        ast2(p.h2 + oo, (1.f - p.zh) * hn + p.zh * h2o);
      }
    }
    MPROF(16);
    // ================= E: K | V | Q row (own 32 columns)
    for (int b = 0; b < B; ++b)
      gather_span(gr + b * gbs + G.dout, D, tag, wave, XW, lane, [&](int i, float v) { xs[b * M2K + i] = v; vb[b * M2N + i] = v; }, err, dead);
    lds_barrier();
    MPROF(17);
    if (32 * wg < 3 * Ds) {
      slice_mul<NB, 2>(sk, xs, D, zs, red, tid);
      if (tid < NB * 32) {
        const int b = tid >> 5, n = 32 * wg + (tid & 31);
        if (b < B && n < 3 * Ds) {
          const float v = zs[b * 32 + (tid & 31)] + bt[136 + (tid & 31)];
          gput(gr + b * gbs + G.kvq + n, tag, v, false);
          ast2(p.kvq + ((int64_t)b * p.Td + t) * 3 * Ds + n, v);        // the cache row (write-through): later steps read it with plain loads
        }
      }
    }
    MPROF(18);
    // ================= F: cached self-attention, own (head, key chunk)
    {
      const int h = wg % heads, ch = wg / heads;
      const int nk = t + 1, per = max(M2T / 16, (nk + NCH - 1) / NCH), j0 = ch * per, j1 = min(j0 + per, nk), nkc = max(j1 - j0, 0);
      const float scale = rsqrtf((float)hd);
      const int kg = tid >> 4, dl = tid & 15, dpl = hd / 16;
      const bool has_t = j1 == nk && nkc > 0;           // the chunk that holds this step's row
      for (int b = 0; b < (nkc > 0 ? B : 0); ++b) {      // (a chunk beyond the filled ones has nothing to publish: nobody gathers it)
        const float* base = p.kvq + (int64_t)b * p.Td * 3 * Ds + h * hd;
        float* sc = red;
        float* part = red + M2TI + 64;
        float* krow = vc;                    // [hd] K row of step t | [hd] V row of step t (from the granules)
        // old rows: plain loads, requested before the poll for the new row
        const int jsafe = min(j0, max(t - 1, 0));
        float4 kv[2];
        const int jk = j0 + kg, jkc = (jk < j1 && jk < t) ? jk : jsafe;
        {
          const float* kp = base + (int64_t)jkc * 3 * Ds + dl * dpl;
          kv[0] = *reinterpret_cast<const float4*>(kp); kv[1] = *reinterpret_cast<const float4*>(kp + 4);
        }
        const int nc4 = hd / 4, ng = M2T / nc4, c4 = tid % nc4, g = tid / nc4;
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int jv = j0 + g + ng * u, jc = (jv < j1 && jv < t) ? jv : jsafe;
          v[u] = *reinterpret_cast<const float4*>(base + (int64_t)jc * 3 * Ds + Ds + 4 * c4);
        }
        // the new row: query of this head (every workgroup), key / value of this head (the chunk that holds row t)
        u64* grow = gr + b * gbs + G.kvq + h * hd;
        if (wave < 2) gather_span(grow + 2 * Ds, hd, tag, wave, 2, lane, [&](int i, float x) { va[i] = x; }, err, dead);
        else if (has_t && wave < 4) gather_span(grow, hd, tag, wave - 2, 2, lane, [&](int i, float x) { krow[i] = x; }, err, dead);
        else if (has_t && wave < 6) gather_span(grow + Ds, hd, tag, wave - 4, 2, lane, [&](int i, float x) { krow[hd + i] = x; }, err, dead);
        lds_barrier();
        for (int jb = 0; jb < nkc; jb += M2T / 16) {
          const int j = j0 + jb + kg;
          float acc = 0.f;
          for (int i = 0; i < dpl; i += 4) {
            float4 kk;
            if (j == t) kk = *reinterpret_cast<const float4*>(krow + dl * dpl + i);
            else if (jb == 0 && i < 8) kk = kv[i >> 2];
            else kk = *reinterpret_cast<const float4*>(base + (int64_t)((j < j1) ? j : jsafe) * 3 * Ds + dl * dpl + i);
            const float* qp = va + dl * dpl + i;
            acc += qp[0] * kk.x + qp[1] * kk.y + qp[2] * kk.z + qp[3] * kk.w;
          }
          SATT_DPP_ADD(acc, 0xB1); SATT_DPP_ADD(acc, 0x4E); SATT_DPP_ADD(acc, 0x141); SATT_DPP_ADD(acc, 0x140);
          if (dl == 0 && j < j1) sc[jb + kg] = acc * scale;
        }
        lds_barrier();
        // chunk statistics: every wave computes them for itself (one workgroup barrier less than a single-wave softmax)
        float cm = -INFINITY, cz = 0.f;
        {
          for (int j = lane; j < nkc; j += 64) cm = fmaxf(cm, sc[j]);
          cm = wave_max(cm);
          for (int j = lane; j < nkc; j += 64) cz += __expf(sc[j] - cm);
          cz = wave_sum(cz);
        }
        {
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int jb = g; jb < nkc; jb += 4 * ng) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int jj = jb + ng * u, j = j0 + jj;
              float4 x = v[u];
              if (j == t) x = *reinterpret_cast<const float4*>(krow + hd + 4 * c4);
              else if (jb != g) x = *reinterpret_cast<const float4*>(base + (int64_t)((j < j1) ? j : jsafe) * 3 * Ds + Ds + 4 * c4);
              const float pj = jj < nkc ? __expf(sc[jj] - cm) : 0.f;
              acc.x += pj * x.x; acc.y += pj * x.y; acc.z += pj * x.z; acc.w += pj * x.w;
            }
          }
          *reinterpret_cast<float4*>(part + g * hd + 4 * c4) = acc;
          lds_barrier();
          u64* dst = gr + b * gbs + G.part + (h * NCH + ch) * (hd + 2);
          if (tid < hd) {
            float o = 0.f;
            for (int gg = 0; gg < ng; ++gg) o += part[gg * hd + tid];
            gput(dst + 2 + tid, tag, o, false);
          }
          if (tid == 0) { gput(dst, tag, nkc > 0 ? cm : -INFINITY, false); gput(dst + 1, tag, nkc > 0 ? cz : 0.f, false); }
        }
        lds_barrier();
      }
    }
    MPROF(19);
    // ================= G1: merge of the chunks that were filled (redundant), folded output transform (split)
    {
      const int nkg = t + 1, perg = max(M2T / 16, (nkg + NCH - 1) / NCH), nch = min((nkg + perg - 1) / perg, NCH);
      float* pm = red;                       // [heads][NCH][hd + 2] gathered partials of one sample
      for (int b = 0; b < B; ++b) {
        const int wph = XW / heads;            // waves per head when the heads are gathered side by side (one poll round trip)
        if (wph * heads == XW && (nch * (hd + 2) + wph - 1) / wph <= 64 * GQ) {
          const int h = wave / wph;
          gather_span(gr + b * gbs + G.part + h * NCH * (hd + 2), nch * (hd + 2), tag, wave - h * wph, wph, lane,
                      [&](int i, float x) { pm[h * NCH * (hd + 2) + i] = x; }, err, dead);
        } else {
          for (int h = 0; h < heads; ++h)
            gather_span(gr + b * gbs + G.part + h * NCH * (hd + 2), nch * (hd + 2), tag, wave, XW, lane,
                        [&](int i, float x) { pm[h * NCH * (hd + 2) + i] = x; }, err, dead);
        }
        lds_barrier();
        MPROF(20);
        for (int c = tid; c < Ds; c += M2T) {
          const int h = c / hd, d = c - h * hd;
          const float* q = pm + h * NCH * (hd + 2);
          float M = -INFINITY;
          for (int c2 = 0; c2 < nch; ++c2) M = fmaxf(M, q[c2 * (hd + 2)]);
          float zt = 0.f, o = 0.f;
          for (int c2 = 0; c2 < nch; ++c2) {
            const float zc_ = q[c2 * (hd + 2) + 1];
            const float f = zc_ > 0.f ? __expf(q[c2 * (hd + 2)] - M) : 0.f;
            zt += f * zc_;
            o += f * q[c2 * (hd + 2) + 2 + d];
          }
          va[b * M2N + c] = o / zt;
        }
        lds_barrier();
      }
    }
    MPROF(21);
    split_mul<NB>(wot, va, M2N, Ds, Ds, bt + 24, SATT_ACT_TANH, vb, M2N, gr + G.tr, gbs, tag, wg, B, red, tid);
    MPROF(22);
    gather_rows<NB>(gr + G.tr, gbs, Ds, tag, vc, M2N, B, tid, err, dead);
    MPROF(23);
    // ================= G2: mel | stop projection (split) -> y, the next step's fed frame
    split_mul<NB>(wou, vc, M2N, Ds, NO, bt + 32, SATT_ACT_NONE, nullptr, 0, gr + G.y, gbs, tag, wg, B, red, tid);
    MPROF(24);
    gather_rows<NB>(gr + G.y, gbs, NO, tag, yv, M2NO, B, tid, err, dead);
    MPROF(25);
    if (wg == 0) {
      for (int i = tid; i < NB * NO; i += M2T) { const int b = i / NO, c = i - b * NO; if (b < B) p.yout[((int64_t)b * (p.Td + 1) + t + 1) * NO + c] = yv[b * M2NO + c]; }
      if (tid == 0) {
        if (p.flag && !p.tin) {
          bool all = true;
          for (int b = 0; b < B; ++b) all = all && (1.f / (1.f + __expf(-yv[b * M2NO + NO - 1])) > p.stop_threshold);
          if (all && t > p.min_steps && stopped == 0) { stopped = t + 1; *p.flag = t + 1; }
        }
        *p.step = t + 1; p.step[1] = t + 1;
      }
    }
    lds_barrier();
    MPROF(26);
#ifdef SATT_MEGA_PROF
    if (wg == 0 && threadIdx.x == 0) { const unsigned long long c_ = clock64(); satt_mega2_prof[30] += c_ - mp_clk; mp_clk = c_; }
#endif
  }
}

inline size_t mega2_lds_bytes(int NB) {
  const size_t fl = M2RED + (size_t)NB * (M2K + M2NO + M2CT + (M2TI + 16) + M2TI + M2N * 4 + 2 * M2TI + 32) + 64 + 4 + 8 * M2N + 17 * 8 + 3 * M2N +
                    (size_t)NB * 8 * (M2N + 64) + 168 + NB + 4;
  return fl * sizeof(float);
}

}  // namespace

#ifdef SATT_MEGA_PROF
extern "C" int satt_dec_mega2_prof_read(unsigned long long* host32, int reset) {
  if (hipMemcpyFromSymbol(host32, HIP_SYMBOL(satt_mega2_prof), sizeof(unsigned long long) * 32) != hipSuccess) return -3;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(satt_mega2_prof), z, sizeof(z)) != hipSuccess) return -3; }
  return 0;
}
#endif

// floats of the exchange buffer `part` of satt_dec_mega_params for the granule form (two floats per granule)
extern "C" int64_t satt_dec_mega2_scratch_floats(int B, int Ti, int Ds, int heads) {
  if (heads < 1) return 0;
  return 2 * (int64_t)B * gl_of(Ti, Ds, heads, Ds / heads).total;
}

int satt_dec_mega2_launch(const satt_dec_mega_params& p, hipStream_t s) {
  const int NB = p.B <= 1 ? 1 : 2;
  if (p.B > 2 || p.heads * (M2G / p.heads) * (p.Ds / p.heads + 2) > M2RED || p.Ds / p.heads < 16) return SATT_E_UNSUPPORTED;
  const size_t smem = mega2_lds_bytes(NB);
  if (smem > 160 * 1024) return SATT_E_UNSUPPORTED;
  if (NB == 1) {
    (void)hipFuncSetAttribute((const void*)dec_mega2_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(dec_mega2_k<1>, dim3(M2G), dim3(M2T), smem, s, p);
  } else {
    (void)hipFuncSetAttribute((const void*)dec_mega2_k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(dec_mega2_k<2>, dim3(M2G), dim3(M2T), smem, s, p);
  }
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}
