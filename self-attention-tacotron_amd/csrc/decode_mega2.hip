// Persistent decode step (r5; its first form - 32 workgroups meeting at six device-wide barriers per step, 41 us per step - was
// deleted in r6, docs/DESIGN_HISTORY.md keeps its anatomy): the training recipe applied to the decode step.
//   * EVERY weight is register resident for the whole launch: the sliced products (attention LSTM, LSTM 1, LSTM 2, K|V|Q: 32 columns
//     per workgroup) hold their [K x 32] bf16 slice in registers, and the small layers (pre-net 0 / 1, query layer, folded output
//     transform, mel | stop projection) are SPLIT 32 ways - 8 columns per workgroup, one 16-byte register per thread;
//   * exchanges are {tag, value} granules (cluster_xchg.h: the data is the flag): a consumer polls the vector it needs, nobody waits
//     for a write-through acknowledgement and there is no separate barrier.  Eleven exchanges per step:
//       p0 -> p1 -> [attention LSTM] hq -> pq -> [energies] e1 | e2 -> [softmax, LSTM 1] h1n -> [LSTM 2] dout ->
//       [K|V|Q] row t -> [cached self-attention partials] -> [merge, output transform] tr -> [projection] y
//   * the CONTEXTS are never formed inside a step: what the cells need of them is  ctx W_c = sum_r alpha_r (values_r W_c), and
//     values W_c (`ctab`: [B][Ti][4][1024], built once per utterance by the caller with one GEMM per table) turns the 128 KB stream of
//     the value rows per workgroup per step into 2 x 2 x Ti x 32 floats and the [K x 32] slices of the two cells lose their 320
//     context rows.  The context state of the launch-per-layer path is written at the LAST step of a launch only (hand-over);
//   * recurrent vectors live in LDS (every workgroup gathers the new h anyway and applies the zoneout interpolation itself), the
//     own units' c / h in registers of the publishing wave; the global state buffers are written at the last step of a launch;
//   * a wave-64 VALU instruction takes 4 cycles and two waves share a SIMD: phases are written for few instructions and for ONE
//     batch of LDS reads each (reads first, arithmetic behind them) - a read / wait / use chain per element was the cost of the first
//     version of this file (softmax 1.4 us, energies 1.6 us, merge 1.0 us for a few hundred flops);
//   * publishing and polling are kept in different waves where possible (the vector-memory counter is in order and counts stores).
// Same buffers and same math as the launch-per-layer path (csrc/decode.hip); selection: satt_dec_mega_supported (A = D = Ds = 256,
// B <= 2, ...); granule tags are step + 1, the caller zeroes the granule buffer when it resets the step counter.  Single-buffered granules are
// safe: between the consumption of X(t) and the production of X(t+1) lies at least one exchange every workgroup contributes to.
#include <cstdlib>
#include "cluster_xchg.h"

#ifdef SATT_MEGA_PROF      // per-phase wall-clock sums (100 MHz) of workgroup 0: tools/build_variant.sh + tools/decode_mega_prof.py
static __device__ unsigned long long satt_mega2_prof[32];
static __device__ int satt_mega2_prof_wg;      // the workgroup whose phases are recorded (satt_dec_mega2_prof_select)
#define MPROF(i) do { if (wg == satt_mega2_prof_wg && threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); satt_mega2_prof[i] += n_ - mp_last; mp_last = n_; } } while (0)
#elif defined(SATT_MEGA_JITTER)
// Timing-robustness build (tools/build_variant.sh jitter decode_mega2.hip -DSATT_MEGA_JITTER; tools/decode_stress.py): behind every
// phase mark and every workgroup barrier each WAVE sleeps, with probability 1/8, a pseudo-random 0 .. 200 us (hash of the real-time
// counter, the workgroup and the wave) - three orders of magnitude more skew between workgroups, and between the waves of a
// workgroup, than any cold start produces.  The exchange protocol (tags, single-buffered granules) and the LDS phase structure
// must give bit-identical results under it; a hole in either shows as a deviation or a time-out.
__device__ __forceinline__ void mega_jitter() {
  unsigned h = (unsigned)wall_clock64() ^ (blockIdx.x * 2654435761u) ^ ((threadIdx.x >> 6) * 2246822519u);
  h ^= h >> 15; h *= 2654435761u; h ^= h >> 13;
  h = __builtin_amdgcn_readfirstlane(h);
  if ((h & 7u) == 0u)
    for (unsigned i = 0, n = (h >> 8) & 63u; i < n; ++i) __builtin_amdgcn_s_sleep(127);
}
#define MPROF(i) mega_jitter()
#define lds_barrier() do { lds_barrier(); mega_jitter(); } while (0)
#else
#define MPROF(i)
#endif

namespace {

constexpr int M2T = 512;                         // threads (8 waves: cluster_xchg.h's gathers are written for XW = 8)
constexpr int M2G = 32;                          // persistent workgroups
constexpr int M2N = 256;                         // A = D = Ds = width of the split layers
constexpr int PUTW = 7, AUXW = 6, GATW = 6;      // the wave that publishes, its helper, the waves [0, GATW) that poll
constexpr int M2TI = 256, M2NO = 168, M2HD = 128;
constexpr int M2PM = 32 * (M2HD + 2);            // gathered self-attention partials of one sample
constexpr int M2TR = 112, TLS = 132;              // context tables stay in LDS up to this many memory rows (B = 1); padded row
constexpr int KLS = M2N + 64;                    // row of the key table: mechanism 1 | mechanism 2
typedef __attribute__((address_space(1))) float gf32q;
__device__ __forceinline__ void ast2(float* p, float v) { __hip_atomic_store((gf32q*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void unpack8q(uint4 v, float (&w)[8]) {
  w[0] = __uint_as_float(v.x << 16); w[1] = __uint_as_float(v.x & 0xFFFF0000u); w[2] = __uint_as_float(v.y << 16); w[3] = __uint_as_float(v.y & 0xFFFF0000u);
  w[4] = __uint_as_float(v.z << 16); w[5] = __uint_as_float(v.z & 0xFFFF0000u); w[6] = __uint_as_float(v.w << 16); w[7] = __uint_as_float(v.w & 0xFFFF0000u);
}
__device__ __forceinline__ float lane_xor32(float v, int lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(v))); }
__device__ __forceinline__ float lane_get(float v, int src) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v))); }

// granule layout (u64 words) per sample
struct GL { int p0, p1, hq, pq, e, h1, dout, kvq, part, tr, y, hs, total; };
__host__ __device__ inline GL gl_of(int hd) {
  GL g; int o = 0;
  g.p0 = o; o += M2N; g.p1 = o; o += M2N; g.hq = o; o += M2N; g.pq = o; o += M2N; g.e = o; o += 2 * M2TI; g.h1 = o; o += M2N; g.dout = o; o += M2N;
  g.kvq = o; o += 3 * M2N; g.part = o; o += M2G * (hd + 2); g.tr = o; o += M2N; g.y = o; o += M2NO;
  g.hs = o; o += M2G;            // placement handshake of a launch (sample 0's area): the XCC id of every workgroup
  g.total = o;
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
// rows [0, cut) of the product are rows [0, cut) of W, rows [cut, K) are rows [cut + skip, ...) (the context rows are not resident)
template <int KI>
__device__ __forceinline__ void slice_fill(SliceR<KI>& w, const uint16_t* __restrict__ W, int ldw, int n0, int K, int cut, int skip, int tid) {
  const int cg = tid & 3, kl = tid >> 2;
#pragma unroll
  for (int i = 0; i < KI; ++i) {
    const int k = min(kl + 128 * i, K - 1), kr = k < cut ? k : k + skip;
    uint4 v = *reinterpret_cast<const uint4*>(W + (int64_t)kr * ldw + n0 + 8 * cg);
    if (kl + 128 * i >= K) v = make_uint4(0u, 0u, 0u, 0u);
    w.v[i] = v;
  }
}
// acc[b][0..8) += the thread's rows of x[b][0..K) W (x: LDS row of 512 floats per sample, finite everywhere)
template <int NB, int KI>
__device__ __forceinline__ void slice_acc(const SliceR<KI>& wv, const float* x, float (&acc)[NB][8], int tid) {
  const int kl = tid >> 2;
  float xv[NB][KI];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < KI; ++i) xv[b][i] = x[b * 512 + kl + 128 * i];
#pragma unroll
  for (int i = 0; i < KI; ++i) {
    float w[8];
    unpack8q(wv.v[i], w);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[b][j] += xv[b][i] * w[j];
  }
}
// transposing reduction over the 16 k lanes of a wave (lane bits 2..5): lane l < 32 ends with the wave total of column
// 8 (l & 3) + ((l >> 2) & 7) - 8 exchanges instead of 32
__device__ __forceinline__ float slice_reduce8(const float (&a)[8], int lane) {
  const bool h2 = lane & 4, h3 = lane & 8, h4 = lane & 16;
  float b[4], c[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const float keep = h2 ? a[2 * i + 1] : a[2 * i], send = h2 ? a[2 * i] : a[2 * i + 1]; b[i] = keep + swz_xor(send, 4); }
#pragma unroll
  for (int i = 0; i < 2; ++i) { const float keep = h3 ? b[2 * i + 1] : b[2 * i], send = h3 ? b[2 * i] : b[2 * i + 1]; c[i] = keep + swz_xor(send, 8); }
  const float keep = h4 ? c[1] : c[0], send = h4 ? c[0] : c[1];
  float d = keep + swz_xor(send, 16);
  d += lane_xor32(d, lane);
  return d;
}
// the wave partials of the 32 columns of every sample -> dst[(wave * NB + b) * 32 + column]
template <int NB>
__device__ __forceinline__ void slice_store(const float (&acc)[NB][8], float* dst, int lane, int wave) {
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const float d = slice_reduce8(acc[b], lane);
    if (lane < 32) dst[(wave * NB + b) * 32 + 8 * (lane & 3) + ((lane >> 2) & 7)] = d;
  }
}
// publishing wave, lane l < 32 NB: column l & 31 of sample l >> 5 summed over the 8 wave partials
template <int NB>
__device__ __forceinline__ float slice_total(const float* src, int lane) {
  const int b = min(lane >> 5, NB - 1), n = lane & 31;
  float v[XW];
#pragma unroll
  for (int w = 0; w < XW; ++w) v[w] = src[(w * NB + b) * 32 + n];
  return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}

// ---- split layer: the workgroup's 8 columns [8 wg, 8 wg + 8) of act(x W + bias) (+ res); thread k < 256 (waves 0..3: K <= 256)
// holds row k of them; the publishing wave finishes and publishes granules dst[b * bs + 8 wg + j] (columns >= N are not published).
// bias8: LDS (a global load here would sit on the step's dependency chain).  rs is free again after the caller's next barrier.
template <int NB>
__device__ __forceinline__ void split_mul(uint4 wr, const float* x, int xs_, int N, const float* bias8, int act, const float* res, int rs_,
                                          u64* dst, int64_t bs, uint32_t tag, int wg, int B, float* rs, int tid, bool sx) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < 4) {
    float xv[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) xv[b] = x[b * xs_ + tid];          // (rows beyond K carry zero weights; x is finite up to 256)
    float w[8];
    unpack8q(wr, w);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = xv[b] * w[j];
      const float tot = wave_sum_transpose<8>(a);      // lane l: total of column l & 7
      if (lane < 8) rs[(b * 4 + wave) * 8 + lane] = tot;
    }
  }
  lds_barrier();
  if (wave == PUTW && lane < NB * 8) {
    const int b = lane >> 3, j = lane & 7, n = 8 * wg + j;
    const float s0 = rs[(b * 4 + 0) * 8 + j], s1 = rs[(b * 4 + 1) * 8 + j], s2 = rs[(b * 4 + 2) * 8 + j], s3 = rs[(b * 4 + 3) * 8 + j];
    const float bj = bias8[j], rv = res ? res[b * rs_ + min(n, M2N - 1)] : 0.f;
    float s = ((s0 + s1) + (s2 + s3)) + bj;
    if (act == SATT_ACT_RELU) s = fmaxf(s, 0.f);
    else if (act == SATT_ACT_TANH) s = tanhf_(s);
    s += rv;
    if (b < B && n < N) gput(dst + b * bs + n, tag, s, sx);
  }
}
__device__ __forceinline__ uint4 split_fill(const uint16_t* __restrict__ W, int ldw, int K, int wg, int tid) {
  const int c0 = min(8 * wg, max(ldw - 8, 0));
  uint4 v = *reinterpret_cast<const uint4*>(W + (int64_t)min(tid, K - 1) * ldw + c0);
  if (tid >= K || 8 * wg >= ldw) v = make_uint4(0u, 0u, 0u, 0u);
  return v;
}

// ---- folded feedback (r6): from the SAME vector x (the output transform's result) the workgroup's 8 columns of the mel | stop
// projection (published as y with `tag`) AND of relu(x Wf + bf) = the first pre-net layer of the NEXT step (published as p0 with
// tag + 1; Wf as bf16 hi + lo).  One barrier; wave PUTW publishes y, wave AUXW p0.
template <int NB>
__device__ __forceinline__ void split_mul_fb(uint4 wy, uint4 wh, uint4 wl, const float* x, int NO_, int P0_, const float* by8, const float* bf8,
                                             u64* dy, u64* dp, int64_t bs, uint32_t tag, int wg, float* rs, int tid, bool sx) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < 4) {
    float xv[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) xv[b] = x[b * M2N + tid];
    float w[8], h[8], l[8];
    unpack8q(wy, w); unpack8q(wh, h); unpack8q(wl, l);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float a[8], a2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[j] = xv[b] * w[j]; a2[j] = fmaf(xv[b], l[j], xv[b] * h[j]); }
      const float t1 = wave_sum_transpose<8>(a), t2 = wave_sum_transpose<8>(a2);
      if (lane < 8) { rs[(b * 4 + wave) * 8 + lane] = t1; rs[NB * 32 + (b * 4 + wave) * 8 + lane] = t2; }
    }
  }
  lds_barrier();
  if ((wave == PUTW || wave == AUXW) && lane < NB * 8) {
    const int b = lane >> 3, j = lane & 7, n = 8 * wg + j;
    const float* r = rs + (wave == AUXW ? NB * 32 : 0);
    const float s0 = r[(b * 4 + 0) * 8 + j], s1 = r[(b * 4 + 1) * 8 + j], s2 = r[(b * 4 + 2) * 8 + j], s3 = r[(b * 4 + 3) * 8 + j];
    float s = ((s0 + s1) + (s2 + s3)) + (wave == AUXW ? bf8[j] : by8[j]);
    if (wave == AUXW) { s = fmaxf(s, 0.f); if (n < P0_) gput(dp + b * bs + n, tag + 1u, s, sx); }
    else if (n < NO_) gput(dy + b * bs + n, tag, s, sx);
  }
}

// waves 0..3 gather n <= 256 granules of every sample (one per lane) and hand them to store(b, i, value); workgroup barrier behind
template <int NB, class St>
__device__ __forceinline__ void gather_vec(u64* src, int64_t bs, int n, uint32_t tag, int B, int tid, unsigned int* err, int* dead, St store) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < 4) {
    const int beg = 64 * wave, cnt = min(64, n - beg);
#pragma unroll
    for (int b = 0; b < NB; ++b)
      if (b < B) gather_poll<1>(src + b * bs + beg, cnt, tag, lane, [&](int i, float v) { store(b, beg + i, v); }, err, dead);
  }
  lds_barrier();
}

// ---- context tables: thread (column group tid & 3, row lane tid >> 2) holds rows rl, rl + 128 of the four tables
//      (LSTM 1 x values1, LSTM 1 x values2, attention LSTM x values1, attention LSTM x values2), 8 columns each
struct TabR { float4 v[16]; };
// Row lane rl owns rows 127 - rl and 255 - rl: the low rows belong to the HIGH waves, so that for Ti <= 112 wave 0 - which polls the
// energies right behind these requests, and whose poll waits for every load in front of it (in-order counter) - requests nothing.
// Rows beyond Ti re-read the last row (one line per wave and request) and are not used.  (Exec-masked requests instead: the
// compiler spills 113 registers around the undefined halves.)
__device__ __forceinline__ void tab_load(TabR& t, const float* __restrict__ ctab, int b, int Ti, int wg, int tid) {
  const int cg = tid & 3, rl = tid >> 2;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = min(127 - rl + 128 * i, Ti - 1);      // (rows beyond Ti: one shared line, not used)
    {
      const float* base = ctab + ((int64_t)b * Ti + r) * 4096 + 32 * wg + 8 * cg;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        t.v[(i * 4 + q) * 2] = *reinterpret_cast<const float4*>(base + q * 1024);
        t.v[(i * 4 + q) * 2 + 1] = *reinterpret_cast<const float4*>(base + q * 1024 + 4);
      }
    }
  }
}
__device__ __forceinline__ void fma8(float (&a)[8], float s, const float4& lo, const float4& hi) {
  a[0] += s * lo.x; a[1] += s * lo.y; a[2] += s * lo.z; a[3] += s * lo.w; a[4] += s * hi.x; a[5] += s * hi.y; a[6] += s * hi.z; a[7] += s * hi.w;
}
// a1 / a2: the sample's alignments of the two mechanisms (LDS, zero beyond the length)
__device__ __forceinline__ void tab_mul(const TabR& t, const float* a1, const float* a2, int Ti, int tid, float (&acc1)[8], float (&acca)[8]) {
  const int rl = tid >> 2;
  float w1[2], w2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int r = 127 - rl + 128 * i; w1[i] = a1[r]; w2[i] = a2[r]; }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (127 - rl + 128 * i < Ti) {
      fma8(acc1, w1[i], t.v[(i * 4 + 0) * 2], t.v[(i * 4 + 0) * 2 + 1]);
      fma8(acc1, w2[i], t.v[(i * 4 + 1) * 2], t.v[(i * 4 + 1) * 2 + 1]);
      fma8(acca, w1[i], t.v[(i * 4 + 2) * 2], t.v[(i * 4 + 2) * 2 + 1]);
      fma8(acca, w2[i], t.v[(i * 4 + 3) * 2], t.v[(i * 4 + 3) * 2 + 1]);
    }
  }
}

// the same from the LDS copy of the workgroup's table slices (tl[r][4][32], row stride TLS; Ti <= 128: one row per thread)
__device__ __forceinline__ void tab_mul_lds(const float* tl, const float* a1, const float* a2, int Ti, int tid, float (&acc1)[8], float (&acca)[8]) {
  const int cg = tid & 3, r = min(tid >> 2, Ti - 1);
  const bool ok = (tid >> 2) < Ti;
  const float* row = tl + r * TLS + 8 * cg;
  float4 v[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) { v[2 * q] = *reinterpret_cast<const float4*>(row + 32 * q); v[2 * q + 1] = *reinterpret_cast<const float4*>(row + 32 * q + 4); }
  const float w1 = ok ? a1[r] : 0.f, w2 = ok ? a2[r] : 0.f;
  fma8(acc1, w1, v[0], v[1]); fma8(acc1, w2, v[2], v[3]); fma8(acca, w1, v[4], v[5]); fma8(acca, w2, v[6], v[7]);
}

// ZoneoutLSTMCell, own unit: gates from the publishing wave's column totals (lane l holds column l & 31 of sample l >> 5)
__device__ __forceinline__ float lstm_unit(float tot, const float* bias32, float extra, int lane, float& c, float& h, float zc, float zh) {
  const int b = (lane >> 3) & 1, u = lane & 7;
  const float v = tot + bias32[lane & 31] + extra;
  const float zi = lane_get(v, b * 32 + u), zj = lane_get(v, b * 32 + 8 + u), zf = lane_get(v, b * 32 + 16 + u), zo = lane_get(v, b * 32 + 24 + u);
  const float cn = sigmoidf_(zf + 1.f) * c + sigmoidf_(zi) * tanhf_(zj);
  const float hn = sigmoidf_(zo) * tanhf_(cn);
  c = (1.f - zc) * cn + zc * c;
  h = (1.f - zh) * hn + zh * h;
  return hn;
}

__host__ __device__ inline size_t mega2_lds_bytes(int NB, int Ti) {
  const size_t fl = 2 * 8 * NB * 32 + NB * 32 + NB * 16 + 320 + 16 * M2HD + 3 * M2HD + M2PM + (size_t)NB * (3 * 512 + M2N + M2NO + (M2TI + 16) + 3 * M2TI + 3 * M2N) +
                    8 * M2N + 16 * 8 + 3 * M2N + (size_t)NB * 8 * KLS + 176 + 4 + 4 + (size_t)NB * 2 * 32 * M2HD + ((NB == 1 && Ti <= M2TR) ? (size_t)Ti * TLS : 0);
  return fl * sizeof(float);
}

// LJ (r6): the dimensions of examples/ljspeech/self-attention-tacotron.json as compile-time constants (checked by the launcher).  The
// step body is ~12 000 instructions with ~100 wave-uniform values live across it; with run-time dimensions 1 600 of them were
// v_readlane / v_writelane traffic of SPILLED scalars (486 spilled SGPRs) in phases that are instruction-issue bound.
template <int NB, bool TRES, bool LJ>
__global__ __launch_bounds__(M2T) void dec_mega2_k(const satt_dec_mega_params p, const int spread) {
  // r6: ONE XCD.  Workgroups are dealt to the 8 XCDs round robin in launch order, so with spread = 8 the grid is 8 x 32 and only the
  // workgroups with blockIdx % 8 == 0 stay: all 32 on the same XCD (32 CUs: one each).  Every weight is register resident, so the one
  // L2 only has to carry the exchanges - and granules published with PLAIN stores stay in that L2, where the peers' polling loads find
  // them: 0.55 us per round trip instead of ~0.9 through memory (the training clusters' same-XCD path, cluster_xchg.h).  The
  // placement is VERIFIED per launch (handshake below: every workgroup publishes its XCC id, all must agree); otherwise - and with
  // spread = 1 (SATT_DECODE_ONE_XCD=0) - the exchanges use agent-scope write-through stores as in r5.
  if (blockIdx.x % spread) return;
  const int wg = blockIdx.x / spread;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* rs = smem;                                  // [8 NB 32] wave partials of the slice / split products
  float* ra = rs + 8 * NB * 32;                      // [8 NB 32] ... of the attention LSTM's context term of the NEXT step
  float* zca = ra + 8 * NB * 32;                     // [NB][32] that term, summed
  float* zs = zca + NB * 32;                         // [NB][2][8] the own rows' energies of the step
  float* fsc = zs + NB * 16;                        // [320] self-attention scores of the own chunk
  float* fpt = fsc + 320;                            // [16][128] P V partials per key group
  float* fq = fpt + 16 * M2HD;                       // [3][128] q | k | v of the new row (own head)
  float* pm = fq + 3 * M2HD;                         // [M2PM] gathered partials
  float* XA = pm + M2PM;                             // [NB][512] p1 | h of the attention LSTM
  float* X1 = XA + NB * 512;                         // [NB][512] hq | h of LSTM 1
  float* X2 = X1 + NB * 512;                         // [NB][512] h1_new | h of LSTM 2
  float* XK = X2 + NB * 512;                         // [NB][256] decoder LSTM output
  float* yv = XK + NB * M2N;                         // [NB][168]
  float* aprev = yv + NB * M2NO;                     // [NB][256 + 16]
  float* alpha = aprev + NB * (M2TI + 16);           // [NB][256]
  float* e1 = alpha + NB * M2TI;                     // [NB][256] alignments of the step
  float* e2 = e1 + NB * M2TI;
  float* va = e2 + NB * M2TI;                        // [NB][256] scratch vectors
  float* vb = va + NB * M2N;
  float* vc = vb + NB * M2N;
  float* Us = vc + NB * M2N;                         // [8][256]
  float* Fs = Us + 8 * M2N;                          // [16][8]
  float* tab = Fs + 16 * 8;                          // [3][256]: v1 | b1 (+ the location layer's bias term) | v2
  float* kls = tab + 3 * M2N;                        // [NB * 8][KLS]
  float* bt = kls + NB * 8 * KLS;                    // [5][8] split-layer biases | [3][32] cell biases (gate-major) | [32] K|V|Q bias
  int* lens = reinterpret_cast<int*>(bt + 176);      // (bt[168..176): bias of the folded feedback layer)
  int* dead = lens + 4;
  float* Kc = reinterpret_cast<float*>(dead + 4);    // [NB][32][128] key rows of the own (head, chunk) while a chunk is 32 rows (t < 512)
  float* Vc = Kc + NB * 32 * M2HD;                   // [NB][32][128] value rows
  float* TL = Vc + NB * 32 * M2HD;                   // [Ti][TLS] the workgroup's slices of the context tables (B = 1, Ti <= M2TR)
  constexpr int B = NB;          // (the launcher instantiates NB = B: B is 1 or 2; r6 - as a run-time value it kept a guard per sample loop alive)
  const int Ti = p.Ti;
  const int U1 = LJ ? 224 : p.U1, U2 = LJ ? 32 : p.U2, UQ = U1 + U2, V1 = LJ ? 256 : p.V1, V2 = LJ ? 32 : p.V2, CT = V1 + V2;
  const int NO = LJ ? 161 : p.NO, KW = LJ ? 10 : p.kernel, F = LJ ? 5 : p.filters, PL = (KW - 1) / 2, heads = LJ ? 2 : p.heads, hd = M2N / heads;
  const int P0 = LJ ? 256 : p.P0, P1 = LJ ? 128 : p.P1, FEED = LJ ? 80 : p.feed;
  const GL G = gl_of(hd);
  u64* gr = reinterpret_cast<u64*>(p.part);
  const int64_t gbs = G.total;                       // granules per sample
  const int R = (Ti + M2G - 1) / M2G, r0 = wg * R;
  const int NCH = M2G / heads;
  constexpr bool tres = TRES;            // the context tables are LDS resident (B = 1, Ti <= M2TR)
  int t = *p.step;
  if (p.flag && !p.tin && *p.flag != 0) return;      // the stop rule fired in an earlier launch (every workgroup reads the same word)
  float sx_lds = 0.f;                                // 1: every workgroup of this launch sits on the same XCD (handshake below)
  {
    const int tid = threadIdx.x, par = t & 1;
    // EVERY word of the allocation starts at zero.  LDS keeps what the previous workgroup on this CU left (another kernel's data, or
    // power-on contents on a CU that has not run anything yet), and several products here multiply rows "beyond K" by zero weights
    // on the assumption that the row is FINITE: 0 x NaN (or 0 x Inf) is NaN, the ReLU behind the pre-net turns it into 0, and the
    // workgroup's 8 columns of pre-net 0 were silently zero for the whole launch.  r6 root cause of the first-utterance deviation
    // (DESIGN.md 3.5): the fed frame is read as yv[NO - 1 - feed + k], k < 256, and yv[NO .. M2NO) was never written.
    {
      float4* z4 = reinterpret_cast<float4*>(smem);
      const int n4 = (int)(mega2_lds_bytes(NB, TRES ? p.Ti : M2TR + 1) / 16);      // (no LDS-resident tables unless TRES)
      for (int i = tid; i < n4; i += M2T) z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (spread > 1) {
      // placement handshake (tag unique per launch of an utterance: the launch's first step): write-through granules, gathered by wave 0
      const uint32_t htag = 0x80000000u | (uint32_t)(t + 1);
      if (tid == 0) gput(gr + G.hs + wg, htag, __uint_as_float((uint32_t)xcc_id() + 1u), false);
      if (tid < 64) gather_poll<1>(gr + G.hs, M2G, htag, tid, [&](int i, float v) { rs[i] = v; }, p.err, dead);
      __syncthreads();
      if (tid == 0) {
        bool same = !*dead;
        for (int i = 1; i < M2G; ++i) same = same && __float_as_uint(rs[i]) == __float_as_uint(rs[0]);
        rs[M2G] = same ? 1.f : 0.f;
      }
      __syncthreads();
      sx_lds = rs[M2G];
      __syncthreads();                               // (rs is reused below)
    }
    for (int i = tid; i < 8 * M2N; i += M2T) { const int f = i / M2N, u = i - f * M2N; Us[i] = (f < F && u < U1) ? p.locU[f * U1 + u] : 0.f; }
    for (int i = tid; i < 3 * M2N; i += M2T) {
      const int w = i / M2N, u = i - w * M2N;
      float v = w == 0 ? (u < U1 ? p.v1[u] : 0.f) : (w == 1 ? (u < U1 ? p.b1[u] : 0.f) : (u < U2 ? p.v2[u] : 0.f));
      if (w == 1 && u < U1)
        for (int f = 0; f < F; ++f) v += p.locFb[f] * p.locU[f * U1 + u];           // (bias of the location convolution, through U)
      tab[i] = v;
    }
    if (tid < 4) lens[tid] = tid < B ? (int)p.lengths[tid] : 0;
    for (int i = tid; i < NB * 8 * KLS; i += M2T) {
      const int row = i / KLS, u = i - row * KLS, b = row / 8, rr = row - b * 8, tt = r0 + rr;
      float v = 0.f;
      if (b < B && rr < R && tt < Ti) v = u < M2N ? (u < U1 ? p.keys1[((int64_t)b * Ti + tt) * U1 + u] : 0.f) : (u - M2N < U2 ? p.keys2[((int64_t)b * Ti + tt) * U2 + u - M2N] : 0.f);
      kls[i] = v;
    }
    for (int i = tid; i < 16 * 8; i += M2T) { const int j = i >> 3, f = i & 7; Fs[i] = (f < F && j < KW) ? p.locF[j * F + f] : 0.f; }
    if (tid < 40) {
      const int l = tid >> 3, n = 8 * wg + (tid & 7);
      const float* bp = l == 0 ? p.bp0 : (l == 1 ? p.bp1 : (l == 2 ? nullptr : (l == 3 ? p.bot : p.bout)));
      const int N = l == 0 ? P0 : (l == 1 ? P1 : (l == 2 ? 0 : (l == 3 ? M2N : NO)));
      bt[tid] = (bp && n < N) ? bp[n] : 0.f;
    } else if (tid >= 64 && tid < 64 + 96) {
      const int i = tid - 64, l = i >> 5, g = (i >> 3) & 3, u = i & 7;
      const float* bp = l == 0 ? p.ba : (l == 1 ? p.b1l : p.b2l);
      bt[40 + i] = bp[g * M2N + 8 * wg + u];
    } else if (tid >= 192 && tid < 224) {
      const int n = 32 * wg + tid - 192;
      bt[136 + tid - 192] = n < 3 * M2N ? p.bkvq[n] : 0.f;
    } else if (tid >= 224 && tid < 232) {
      const int n = 8 * wg + tid - 224;
      bt[168 + tid - 224] = (p.bfb && n < P0) ? p.bfb[n] : 0.f;
    }
    __syncthreads();
    // state of step t: location-conv input, forward variable, recurrent vectors, the previous step's alignments, the fed frame
    for (int i = tid; i < NB * Ti; i += M2T) {
      const int b = i / Ti, r = i - b * Ti;
      if (b < B) {
        aprev[b * (M2TI + 16) + PL + r] = p.a_state[((int64_t)par * B + b) * Ti + r];
        alpha[b * M2TI + r] = p.alpha_state[((int64_t)par * B + b) * Ti + r];
        if (t > 0) {
          e1[b * M2TI + r] = p.align1[((int64_t)b * p.Td + t - 1) * Ti + r];
          e2[b * M2TI + r] = p.align2[((int64_t)b * p.Td + t - 1) * Ti + r];
        }
      }
    }
    for (int i = tid; i < NB * M2N; i += M2T) {
      const int b = i >> 8, k = i & 255;
      if (b < B) {
        XA[b * 512 + P1 + k] = p.ha[((int64_t)par * B + b) * M2N + k];
        X1[b * 512 + M2N + k] = p.h1[((int64_t)par * B + b) * M2N + k];
        X2[b * 512 + M2N + k] = p.h2[((int64_t)par * B + b) * M2N + k];
      }
    }
    for (int i = tid; i < NB * NO; i += M2T) { const int b = i / NO, c = i - b * NO; yv[b * M2NO + c] = b < B ? p.yout[((int64_t)b * (p.Td + 1) + t) * NO + c] : 0.f; }
    // rows [j0, t) of the own (head, chunk) of the K|V cache (a chunk is 32 rows while t < 512; beyond that the step reads the cache)
    if ((t + NCH) / NCH <= M2T / 16) {
      const int h = wg % heads, j0 = (wg / heads) * (M2T / 16), c4n = hd / 4;
      for (int i = tid; i < NB * 32 * c4n; i += M2T) {
        const int b = i / (32 * c4n), rr = (i / c4n) & 31, c4 = i % c4n, j = j0 + rr;
        if (b < B && j < t) {
          const float* src = p.kvq + ((int64_t)b * p.Td + j) * 3 * M2N + h * hd + 4 * c4;
          *reinterpret_cast<float4*>(Kc + (b * 32 + rr) * M2HD + 4 * c4) = *reinterpret_cast<const float4*>(src);
          *reinterpret_cast<float4*>(Vc + (b * 32 + rr) * M2HD + 4 * c4) = *reinterpret_cast<const float4*>(src + M2N);
        }
      }
    }
  }
#ifdef SATT_MEGA_PROF
  unsigned long long mp_last = wall_clock64(), mp_clk = clock64();
#endif
  // own units' cell state: lanes < 8 NB of the publishing wave
  float cA = 0.f, hA = 0.f, c1 = 0.f, h1 = 0.f, c2 = 0.f, h2 = 0.f;
  if ((int)threadIdx.x >= 64 * PUTW && (int)threadIdx.x < 64 * PUTW + 8 * NB) {
    const int l = threadIdx.x - 64 * PUTW, b = min(l >> 3, B - 1), eu = 8 * wg + (l & 7);
    const int64_t o = ((int64_t)(t & 1) * B + b) * M2N + eu;
    cA = p.ca[o]; hA = p.ha[o]; c1 = p.c1[o]; h1 = p.h1[o]; c2 = p.c2[o]; h2 = p.h2[o];
  }
  // ---- resident weights (registers for the whole launch)
  SliceR<4> sa, s1, s2; SliceR<2> sk;
  slice_fill(sa, p.Wa, 4 * M2N, 32 * wg, P1 + M2N, P1, CT, (int)threadIdx.x);
  slice_fill(s1, p.W1, 4 * M2N, 32 * wg, 2 * M2N, M2N, CT, (int)threadIdx.x);
  slice_fill(s2, p.W2, 4 * M2N, 32 * wg, 2 * M2N, 2 * M2N, 0, (int)threadIdx.x);
  slice_fill(sk, p.Wkvq, 3 * M2N, min(32 * wg, 3 * M2N - 32), M2N, M2N, 0, (int)threadIdx.x);
  uint4 wp0 = split_fill(p.Wp0, P0, FEED, wg, (int)threadIdx.x), wp1 = split_fill(p.Wp1, P1, P0, wg, (int)threadIdx.x);
  uint4 wqr = split_fill(p.Wq, UQ, M2N, wg, (int)threadIdx.x), wot = split_fill(p.Wot, M2N, M2N, wg, (int)threadIdx.x);
  uint4 wou = split_fill(p.Wout, p.ldout, M2N, wg, (int)threadIdx.x);
  const bool fold = p.Wfh && p.Wfl && p.bfb && !p.tin;      // folded feedback (free running only)
  uint4 wfh = fold ? split_fill(p.Wfh, P0, M2N, wg, (int)threadIdx.x) : make_uint4(0u, 0u, 0u, 0u);
  uint4 wfl = fold ? split_fill(p.Wfl, P0, M2N, wg, (int)threadIdx.x) : make_uint4(0u, 0u, 0u, 0u);
  bool have_p0 = false;                                     // the step's first pre-net layer is already in `vb` (previous step of this launch)
  __syncthreads();
  if constexpr (tres) {
    for (int i = threadIdx.x; i < Ti * 128; i += M2T) { const int r = i >> 7, c = i & 127; TL[r * TLS + c] = p.ctab[(int64_t)r * 4096 + (c >> 5) * 1024 + 32 * wg + (c & 31)]; }
    __syncthreads();
  }
  {      // context term of the attention LSTM at step t, from the alignments of step t - 1 (zero at t = 0)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float a1[NB][8], aa[NB][8];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { a1[b][j] = 0.f; aa[b][j] = 0.f; }
      if (b < B) {
        if constexpr (tres) tab_mul_lds(TL, e1, e2, Ti, tid, a1[b], aa[b]);
        else { TabR tb; tab_load(tb, p.ctab, b, Ti, wg, tid); tab_mul(tb, e1 + b * M2TI, e2 + b * M2TI, Ti, tid, a1[b], aa[b]); }
      }
    }
    slice_store<NB>(aa, ra, lane, wave);
    __syncthreads();
    if (tid < NB * 32) zca[tid] = slice_total<NB>(ra, tid);
    __syncthreads();
  }
  const int nsteps = p.nsteps;
  const bool sx = __builtin_amdgcn_readfirstlane((int)(spread > 1 && sx_lds != 0.f)) != 0;      // plain-store exchanges (same XCD, verified)
  for (int s = 0; s < nsteps; ++s, ++t) {
    typedef const __attribute__((address_space(4))) satt_dec_mega_params KArgsM;
    KArgsM* kq = (KArgsM*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kq));
    const auto& p = *kq;
    int oz = 0;
    asm volatile("" : "+v"(oz));
    pin(sa); pin(s1); pin(s2); pin(sk); pin(wp0); pin(wp1); pin(wqr); pin(wot); pin(wou); pin(wfh); pin(wfl);
    const int tid = (int)threadIdx.x + oz, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(tid >= 0 && tid < M2T && wave >= 0 && wave < XW);
    const int par = t & 1;
    const bool last = s == nsteps - 1;
    const uint32_t tag = (uint32_t)(t + 1);
    unsigned int* err = p.err;
    const float zc = p.zc, zh = p.zh;
    // ================= A1: pre-net 0 (split) on the fed frame
    const float* fed = yv + (NO - 1 - FEED);      // free running: the frame this workgroup gathered at the end of the previous step
    int fstr = M2NO;
    if (p.tin) {
      for (int i = tid; i < NB * M2N; i += M2T) {
        const int b = i >> 8, k = i & 255;
        va[i] = (b < B && k < FEED) ? p.tin[((int64_t)b * p.Td + t) * FEED + k] : 0.f;
      }
      lds_barrier();
      fed = va; fstr = M2N;
    }
    MPROF(0);
    if (!have_p0) {      // (folded feedback: the previous step of this launch published and gathered this step's p0 behind its projection)
      split_mul<NB>(wp0, fed, fstr, P0, bt, SATT_ACT_RELU, nullptr, 0, gr + G.p0, gbs, tag, wg, B, rs, tid, sx);
      MPROF(1);
      gather_vec<NB>(gr + G.p0, gbs, P0, tag, B, tid, err, dead, [&](int b, int i, float v) { vb[b * M2N + i] = v; });
    }
    MPROF(2);
    // ================= A2: pre-net 1 (split)
    split_mul<NB>(wp1, vb, M2N, P1, bt + 8, SATT_ACT_RELU, nullptr, 0, gr + G.p1, gbs, tag, wg, B, rs, tid, sx);
    MPROF(3);
    gather_vec<NB>(gr + G.p1, gbs, P1, tag, B, tid, err, dead, [&](int b, int i, float v) { XA[b * 512 + i] = v; });
    MPROF(4);
    // ================= A3: attention LSTM slice + cell ([p1 | h] W + the context term of the previous step's alignments)
    {
      float acc[NB][8];
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[b][j] = 0.f;
      slice_acc<NB, 4>(sa, XA, acc, tid);
      slice_store<NB>(acc, rs, lane, wave);
    }
    lds_barrier();
    if (wave == PUTW) {
      const float tot = slice_total<NB>(rs, lane);
      const float hn = lstm_unit(tot, bt + 40, zca[min(lane, NB * 32 - 1)], lane, cA, hA, zc, zh);
      if (lane < 8 * B) {
        const int b = lane >> 3, eu = 8 * wg + (lane & 7);
        gput(gr + b * gbs + G.hq + eu, tag, hn, sx);
        if (last) { const int64_t oo = ((int64_t)(par ^ 1) * B + b) * M2N + eu; p.ca[oo] = cA; p.ha[oo] = hA; }
      }
    }
    MPROF(5);
    gather_vec<NB>(gr + G.hq, gbs, M2N, tag, B, tid, err, dead, [&](int b, int i, float v) {
      X1[b * 512 + i] = v;
      float* hs = XA + b * 512 + P1 + i;
      *hs = (1.f - zh) * v + zh * *hs;
    });
    MPROF(6);
    // ================= B1: query layer (split)
    split_mul<NB>(wqr, X1, 512, UQ, bt + 16, SATT_ACT_NONE, nullptr, 0, gr + G.pq, gbs, tag, wg, B, rs, tid, sx);
    MPROF(7);
    gather_vec<NB>(gr + G.pq, gbs, UQ, tag, B, tid, err, dead, [&](int b, int i, float v) { va[b * M2N + i] = v; });
    MPROF(8);
    // ================= B2: energies of the own rows: row pr on wave 7 - pr (the polling waves stay free when there are <= 4 rows)
#ifdef SATT_MEGA_PROF
    const unsigned long long en0 = wall_clock64();
#endif
    for (int pr = 7 - wave; pr < B * R; pr += XW) {
      const int b = pr / R, rr = pr - b * R, tt = r0 + rr;
      if (tt < Ti) {
        const float* kr = kls + (b * 8 + rr) * KLS;
        const float* pq = va + b * M2N;
        // location features: lane = (filter l & 7, tap group l >> 3): taps jj = group, group + 8; xor-reduced over the groups
        const int f = lane & 7, jg = lane >> 3;
        const float* ap = aprev + b * (M2TI + 16) + tt;
        const float t0 = ap[jg], t1 = ap[jg + 8], f0 = Fs[jg * 8 + f], f1 = Fs[(jg + 8) * 8 + f];      // (Fs rows >= kernel are zero)
        const float k2 = kr[M2N + lane], q2 = pq[min(U1 + lane, M2N - 1)], v2r = tab[2 * M2N + lane];
        float part = t0 * f0 + t1 * f1;
        part += swz_xor(part, 8); part += swz_xor(part, 16);
        part += lane_xor32(part, lane);
        float fl[8];
#pragma unroll
        for (int ff = 0; ff < 8; ++ff) fl[ff] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(part), ff));
        float a[2] = {0.f, v2r * tanhf_(k2 + q2)};          // (v2 is zero beyond U2, every operand finite)
#pragma unroll
        for (int q0 = 0; q0 < 4; q0 += 2) {
          if (64 * q0 < U1) {
            const int d0 = lane + 64 * q0, d1 = d0 + 64;
            const float ka = kr[d0], kb = kr[d1], ba_ = tab[M2N + d0], bb_ = tab[M2N + d1], pa = pq[d0], pb = pq[d1], wa_ = tab[d0], wb_ = tab[d1];
            float ua[8], ub[8];
#pragma unroll
            for (int ff = 0; ff < 8; ++ff) { ua[ff] = Us[ff * M2N + d0]; ub[ff] = Us[ff * M2N + d1]; }
            float xa = ka + ba_ + pa, xb = kb + bb_ + pb;
#pragma unroll
            for (int ff = 0; ff < 8; ++ff) { xa += fl[ff] * ua[ff]; xb += fl[ff] * ub[ff]; }
            a[0] += wa_ * tanhf_(xa) + wb_ * tanhf_(xb);          // (v1, U and the keys are zero beyond U1; pq there is finite)
          }
        }
        wave_sum_multi<2>(a);
        if (lane == 0) { zs[(b * 2 + 0) * 8 + rr] = a[0]; zs[(b * 2 + 1) * 8 + rr] = a[1]; }
      }
    }
    lds_barrier();
    // The workgroup's 2 R energies of a sample are ONE contiguous run of granules [wg][mechanism][row], published by one store
    // instruction: 200 separate 8-byte write-throughs into the same 26 lines (one per row and mechanism, as the first version did)
    // serialise at the memory side - that exchange took 2.1 us where the others take 0.8.
    // (rows beyond the sample's length are published too: their consumers mask them - every granule of [0, Ti) gets its tag)
    if (wave == PUTW && lane < 2 * R * NB) {
      const int b = lane / (2 * R), l = lane - b * 2 * R, mech = l / R, rr = l - mech * R;
      if (b < B && r0 + rr < Ti) gput(gr + b * gbs + G.e + wg * 2 * R + l, tag, zs[(b * 2 + mech) * 8 + rr], sx);
    }
#ifdef SATT_MEGA_PROF
    if (wg == satt_mega2_prof_wg && threadIdx.x == 64 * PUTW) satt_mega2_prof[31] += wall_clock64() - en0;
#endif
    MPROF(9);
    // the context tables of sample 0 do not depend on the alignments: requested before the energy exchange
    TabR tb;
    if constexpr (!tres) tab_load(tb, p.ctab, 0, Ti, wg, tid);
    // ================= C: softmax + forward recursion, one wave per (sample, mechanism): polls its energies into registers
    if (wave < 2 * B) {
      const int b = wave >> 1, mech = wave & 1, len = lens[b];
      float* al = alpha + b * M2TI;
      float* ap = aprev + b * (M2TI + 16) + PL;
      float alv[4], alm[4], apv[4], ev[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {      // (what the recursion needs of the previous step: requested before the poll)
        const int i = lane + 64 * q;
        alv[q] = al[i]; alm[q] = al[max(i - 1, 0)]; apv[q] = ap[i]; ev[q] = 0.f;
      }
      {
        const gu64* g[4];
        u64 x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = min(lane + 64 * q, Ti - 1), w = i / R;
          g[q] = (const gu64*)(gr + b * gbs + G.e + w * 2 * R + mech * R + (i - w * R));
          x[q] = 0;
        }
        if (!*dead && !poll_until<4>(g, tag, x)) {
          if (lane == 0) __hip_atomic_store((gu32*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          *dead = 1;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) ev[q] = __uint_as_float((uint32_t)x[q]);
      }
      MPROF(24);
      float m = -INFINITY;
#pragma unroll
      for (int q = 0; q < 4; ++q) m = fmaxf(m, lane + 64 * q < len ? ev[q] : -INFINITY);
      m = wave_max(m);
      float x[4], sacc = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) { x[q] = lane + 64 * q < len ? __expf(ev[q] - m) : 0.f; sacc += x[q]; }
      const float rsum = 1.f / wave_sum(sacc);
      if (mech == 1) {
        float* e = e2 + b * M2TI;
#pragma unroll
        for (int q = 0; q < 4; ++q) e[lane + 64 * q] = x[q] * rsum;
      } else {
        float* e = e1 + b * M2TI;
        float keep[4], s2_ = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = lane + 64 * q;
          const float a = x[q] * rsum;
          if (i < Ti) ap[i] = p.cumulative ? a + apv[q] : a;
          keep[q] = a;
          if (p.att1_mode == 0) { keep[q] = i < Ti ? (0.5f * alv[q] + 0.5f * (i > 0 ? alm[q] : 0.f) + 1e-7f) * a : 0.f; s2_ += keep[q]; }
        }
        if (p.att1_mode == 0) {
          const float r2 = 1.f / wave_sum(s2_);
#pragma unroll
          for (int q = 0; q < 4; ++q) keep[q] *= r2;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { al[lane + 64 * q] = keep[q]; e[lane + 64 * q] = keep[q]; }
      }
    }
    MPROF(25);
    lds_barrier();
    MPROF(10);
    // histories (one workgroup each, stores only)
    if (wg == 1 % M2G) {
      for (int i = tid; i < NB * Ti; i += M2T) {
        const int b = i / Ti, r = i - b * Ti;
        if (b < B) {
          p.align1[((int64_t)b * p.Td + t) * Ti + r] = e1[b * M2TI + r];
          p.align2[((int64_t)b * p.Td + t) * Ti + r] = e2[b * M2TI + r];
        }
      }
    }
    if (last && wg == 2 % M2G) {
      for (int i = tid; i < NB * Ti; i += M2T) {
        const int b = i / Ti, r = i - b * Ti;
        if (b < B) {
          p.a_state[((int64_t)(par ^ 1) * B + b) * Ti + r] = aprev[b * (M2TI + 16) + PL + r];
          p.alpha_state[((int64_t)(par ^ 1) * B + b) * Ti + r] = e1[b * M2TI + r];
        }
      }
    }
    // ================= C2: LSTM 1 on [hq | h] + the context term; the attention LSTM's context term of the NEXT step
    {
      float acc[NB][8], aa[NB][8];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[b][j] = 0.f; aa[b][j] = 0.f; }
        if (b < B) {
          if constexpr (tres) tab_mul_lds(TL, e1, e2, Ti, tid, acc[b], aa[b]);
          else {
            if (b > 0) tab_load(tb, p.ctab, b, Ti, wg, tid);
            tab_mul(tb, e1 + b * M2TI, e2 + b * M2TI, Ti, tid, acc[b], aa[b]);
          }
        }
      }
      slice_acc<NB, 4>(s1, X1, acc, tid);
      slice_store<NB>(acc, rs, lane, wave);
      slice_store<NB>(aa, ra, lane, wave);
    }
    lds_barrier();
    if (wave == PUTW) {
      const float tot = slice_total<NB>(rs, lane);
      const float hn = lstm_unit(tot, bt + 72, 0.f, lane, c1, h1, zc, zh);
      if (lane < 8 * B) {
        const int b = lane >> 3, eu = 8 * wg + (lane & 7);
        gput(gr + b * gbs + G.h1 + eu, tag, hn, sx);
        if (last) { const int64_t oo = ((int64_t)(par ^ 1) * B + b) * M2N + eu; p.c1[oo] = c1; p.h1[oo] = h1; }
      }
    } else if (wave == AUXW) {
      const float tot = slice_total<NB>(ra, lane);
      if (lane < NB * 32) zca[lane] = tot;
    }
    MPROF(11);
    // ================= D: LSTM 2 on [h1_new | h]
    gather_vec<NB>(gr + G.h1, gbs, M2N, tag, B, tid, err, dead, [&](int b, int i, float v) {
      X2[b * 512 + i] = v;
      float* hs = X1 + b * 512 + M2N + i;
      *hs = (1.f - zh) * v + zh * *hs;
    });
    MPROF(12);
    {
      float acc[NB][8];
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[b][j] = 0.f;
      slice_acc<NB, 4>(s2, X2, acc, tid);
      slice_store<NB>(acc, rs, lane, wave);
    }
    lds_barrier();
    if (wave == PUTW) {
      const float tot = slice_total<NB>(rs, lane);
      const float hn = lstm_unit(tot, bt + 104, 0.f, lane, c2, h2, zc, zh);
      if (lane < 8 * B) {
        const int b = lane >> 3, eu = 8 * wg + (lane & 7);
        gput(gr + b * gbs + G.dout + eu, tag, hn, sx);
        if (last) { const int64_t oo = ((int64_t)(par ^ 1) * B + b) * M2N + eu; p.c2[oo] = c2; p.h2[oo] = h2; }
      }
    }
    MPROF(13);
    // ================= E: K | V | Q row (own 32 columns)
    gather_vec<NB>(gr + G.dout, gbs, M2N, tag, B, tid, err, dead, [&](int b, int i, float v) {
      XK[b * M2N + i] = v;
      float* hs = X2 + b * 512 + M2N + i;
      *hs = (1.f - zh) * v + zh * *hs;
    });
    MPROF(14);
    if (32 * wg < 3 * M2N) {
      float acc[NB][8];
      const int kl = tid >> 2;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float x0 = XK[b * M2N + kl], x1 = XK[b * M2N + kl + 128];
        float w0[8], w1[8];
        unpack8q(sk.v[0], w0); unpack8q(sk.v[1], w1);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[b][j] = x0 * w0[j] + x1 * w1[j];
      }
      slice_store<NB>(acc, rs, lane, wave);
    }
    lds_barrier();
    if (32 * wg < 3 * M2N && wave == PUTW) {
      const float v = slice_total<NB>(rs, lane) + bt[136 + (lane & 31)];
      const int b = lane >> 5, n = 32 * wg + (lane & 31);
      if (lane < 32 * B) {
        gput(gr + b * gbs + G.kvq + n, tag, v, sx);
        ast2(p.kvq + ((int64_t)b * p.Td + t) * 3 * M2N + n, v);        // the cache row (write-through): later steps read it with plain loads
      }
    }
    MPROF(15);
    // ================= F: cached self-attention, own (head, key chunk)
    {
      const int h = wg % heads, ch = wg / heads;
      const int nk = t + 1, per = max(M2T / 16, (nk + NCH - 1) / NCH), j0 = ch * per, j1 = min(j0 + per, nk), nkc = max(j1 - j0, 0);
      const float scale = rsqrtf((float)hd);
      const int kg = tid >> 4, dl = tid & 15, dpl = hd / 16;
      const bool has_t = j1 == nk && nkc > 0;           // the chunk that holds this step's row
      float* krow = fq + M2HD;                          // [hd] K row | [hd] V row of step t (from the granules)
      const bool resident = per == M2T / 16;            // the chunk's rows are in LDS: no global load on the step's chain
      for (int b = 0; b < ((nkc > 0 && resident) ? B : 0); ++b) {
        float* Kb = Kc + b * 32 * M2HD;
        float* Vb = Vc + b * 32 * M2HD;
        u64* grow = gr + b * gbs + G.kvq + h * hd;
        // the new row: query of this head (every active chunk), key / value (the chunk that holds row t: appended to its LDS rows)
        if (wave < 2) gather_span(grow + 2 * M2N, hd, tag, wave, 2, lane, [&](int i, float x) { fq[i] = x; }, err, dead);
        else if (has_t && wave < 4) gather_span(grow, hd, tag, wave - 2, 2, lane, [&](int i, float x) { Kb[(t - j0) * M2HD + i] = x; }, err, dead);
        else if (has_t && wave < 6) gather_span(grow + M2N, hd, tag, wave - 4, 2, lane, [&](int i, float x) { Vb[(t - j0) * M2HD + i] = x; }, err, dead);
        MPROF(26);
        lds_barrier();
        MPROF(27);
        {      // scores: 16 threads per key, 32 keys
          const int kc = min(kg, nkc - 1);
          float acc = 0.f;
          if (dpl == 8) {
            const float4 k0 = *reinterpret_cast<const float4*>(Kb + kc * M2HD + dl * 8), k1 = *reinterpret_cast<const float4*>(Kb + kc * M2HD + dl * 8 + 4);
            const float4 q0 = *reinterpret_cast<const float4*>(fq + dl * 8), q1 = *reinterpret_cast<const float4*>(fq + dl * 8 + 4);
            acc = (q0.x * k0.x + q0.y * k0.y + q0.z * k0.z + q0.w * k0.w) + (q1.x * k1.x + q1.y * k1.y + q1.z * k1.z + q1.w * k1.w);
          } else {
            const float4 k0 = *reinterpret_cast<const float4*>(Kb + kc * M2HD + dl * 4), q0 = *reinterpret_cast<const float4*>(fq + dl * 4);
            acc = q0.x * k0.x + q0.y * k0.y + q0.z * k0.z + q0.w * k0.w;
          }
          SATT_DPP_ADD(acc, 0xB1); SATT_DPP_ADD(acc, 0x4E); SATT_DPP_ADD(acc, 0x141); SATT_DPP_ADD(acc, 0x140);
          if (dl == 0 && kg < nkc) fsc[kg] = acc * scale;
        }
        lds_barrier();
        MPROF(28);
        const int nc4 = hd / 4, ng = M2T / nc4, c4 = tid % nc4, g = tid / nc4;
        float cm, cz;
        {      // chunk statistics (every wave for itself) and P V: thread = (4 columns, key group g of ng): keys g, g + ng
          const float sv = lane < nkc ? fsc[lane] : -INFINITY;
          const int ja = min(g, nkc - 1), jb2 = min(g + ng, nkc - 1);
          const float sa_ = fsc[ja], sb_ = fsc[jb2];
          const float4 xa = *reinterpret_cast<const float4*>(Vb + ja * M2HD + 4 * c4), xb = *reinterpret_cast<const float4*>(Vb + jb2 * M2HD + 4 * c4);
          cm = wave_max(sv);
          cz = wave_sum(lane < nkc ? __expf(sv - cm) : 0.f);
          const float wa_ = g < nkc ? __expf(sa_ - cm) : 0.f, wb_ = g + ng < nkc ? __expf(sb_ - cm) : 0.f;
          *reinterpret_cast<float4*>(fpt + g * hd + 4 * c4) = make_float4(wa_ * xa.x + wb_ * xb.x, wa_ * xa.y + wb_ * xb.y, wa_ * xa.z + wb_ * xb.z, wa_ * xa.w + wb_ * xb.w);
        }
        lds_barrier();
        MPROF(29);
        if (wave >= AUXW) {      // published by waves 6, 7 (the polling waves go on to the partials of the other chunks)
          const int d = tid - 64 * AUXW;
          u64* dst = gr + b * gbs + G.part + (h * NCH + ch) * (hd + 2);
          if (d < hd) {
            float o[16];
#pragma unroll
            for (int gg = 0; gg < 16; ++gg) o[gg] = fpt[gg * hd + d];          // (ng >= 16)
            float osum = ((o[0] + o[1]) + (o[2] + o[3])) + ((o[4] + o[5]) + (o[6] + o[7])) + (((o[8] + o[9]) + (o[10] + o[11])) + ((o[12] + o[13]) + (o[14] + o[15])));
            for (int gg = 16; gg < ng; ++gg) osum += fpt[gg * hd + d];
            gput(dst + 2 + d, tag, osum, sx);
          }
          if (d == 0) { gput(dst, tag, cm, sx); gput(dst + 1, tag, cz, sx); }
        }
        if (B > 1) lds_barrier();
      }
      for (int b = 0; b < ((nkc > 0 && !resident) ? B : 0); ++b) {      // (a chunk beyond the filled ones has nothing to publish)
        const float* base = p.kvq + (int64_t)b * p.Td * 3 * M2N + h * hd;
        // old rows: plain loads, requested before the poll for the new row
        const int jsafe = min(j0, max(t - 1, 0));
        float4 kv[2];
        // (at step 0 there is no old row: the clamped requests go to an immutable buffer - a load of row 0 before its write-through
        //  has landed would leave a stale line in this XCD's L2 for the later steps to hit)
        const float* old0 = t > 0 ? base + (int64_t)jsafe * 3 * M2N : p.ctab;
        const int jk = j0 + kg;
        {
          const float* kp = ((jk < j1 && jk < t) ? base + (int64_t)jk * 3 * M2N : old0) + dl * dpl;
          kv[0] = *reinterpret_cast<const float4*>(kp); kv[1] = *reinterpret_cast<const float4*>(kp + (dpl > 4 ? 4 : 0));
        }
        const int nc4 = hd / 4, ng = M2T / nc4, c4 = tid % nc4, g = tid / nc4;
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int jv = j0 + g + ng * u;
          v[u] = *reinterpret_cast<const float4*>(((jv < j1 && jv < t) ? base + (int64_t)jv * 3 * M2N : old0) + M2N + 4 * c4);
        }
        // the new row: query of this head (every workgroup), key / value of this head (the chunk that holds row t)
        u64* grow = gr + b * gbs + G.kvq + h * hd;
        if (wave < 2) gather_span(grow + 2 * M2N, hd, tag, wave, 2, lane, [&](int i, float x) { fq[i] = x; }, err, dead);
        else if (has_t && wave < 4) gather_span(grow, hd, tag, wave - 2, 2, lane, [&](int i, float x) { krow[i] = x; }, err, dead);
        else if (has_t && wave < 6) gather_span(grow + M2N, hd, tag, wave - 4, 2, lane, [&](int i, float x) { krow[hd + i] = x; }, err, dead);
        lds_barrier();
        for (int jb = 0; jb < nkc; jb += M2T / 16) {
          const int j = j0 + jb + kg;
          float acc = 0.f;
          for (int i = 0; i < dpl; i += 4) {
            float4 kk;
            if (j == t) kk = *reinterpret_cast<const float4*>(krow + dl * dpl + i);
            else if (jb == 0 && i < 8) kk = kv[i >> 2];
            else kk = *reinterpret_cast<const float4*>(base + (int64_t)((j < j1) ? j : jsafe) * 3 * M2N + dl * dpl + i);
            const float4 qq = *reinterpret_cast<const float4*>(fq + dl * dpl + i);
            acc += qq.x * kk.x + qq.y * kk.y + qq.z * kk.z + qq.w * kk.w;
          }
          SATT_DPP_ADD(acc, 0xB1); SATT_DPP_ADD(acc, 0x4E); SATT_DPP_ADD(acc, 0x141); SATT_DPP_ADD(acc, 0x140);
          if (dl == 0 && j < j1) fsc[jb + kg] = acc * scale;
        }
        lds_barrier();
        // chunk statistics: every wave computes them for itself (one workgroup barrier less than a single-wave softmax)
        float cm = -INFINITY, cz = 0.f;
        {
          for (int j = lane; j < nkc; j += 64) cm = fmaxf(cm, fsc[j]);
          cm = wave_max(cm);
          for (int j = lane; j < nkc; j += 64) cz += __expf(fsc[j] - cm);
          cz = wave_sum(cz);
        }
        {
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int jb = g; jb < nkc; jb += 4 * ng) {
            float pj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) pj[u] = fsc[min(jb + ng * u, nkc - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int jj = jb + ng * u, j = j0 + jj;
              float4 x = v[u];
              if (j == t) x = *reinterpret_cast<const float4*>(krow + hd + 4 * c4);
              else if (jb != g) x = *reinterpret_cast<const float4*>(base + (int64_t)((j < j1) ? j : jsafe) * 3 * M2N + M2N + 4 * c4);
              const float w = jj < nkc ? __expf(pj[u] - cm) : 0.f;
              acc.x += w * x.x; acc.y += w * x.y; acc.z += w * x.z; acc.w += w * x.w;
            }
          }
          *reinterpret_cast<float4*>(fpt + g * hd + 4 * c4) = acc;
        }
        lds_barrier();
        if (wave >= AUXW) {      // published by waves 6, 7 (the polling waves go on to the partials of the other chunks)
          const int d = tid - 64 * AUXW;
          u64* dst = gr + b * gbs + G.part + (h * NCH + ch) * (hd + 2);
          if (d < hd) {
            float o[16];
#pragma unroll
            for (int gg = 0; gg < 16; ++gg) o[gg] = fpt[min(gg, ng - 1) * hd + d];
            float osum = 0.f;
#pragma unroll
            for (int gg = 0; gg < 16; ++gg) osum += gg < ng ? o[gg] : 0.f;
            for (int gg = 16; gg < ng; ++gg) osum += fpt[gg * hd + d];
            gput(dst + 2 + d, tag, osum, sx);
          }
          if (d == 0) { gput(dst, tag, cm, sx); gput(dst + 1, tag, cz, sx); }
        }
        if (B > 1) lds_barrier();
      }
    }
    MPROF(16);
    // ================= G1: merge of the chunks that were filled (redundant), folded output transform (split)
    {
      const int nkg = t + 1, perg = max(M2T / 16, (nkg + NCH - 1) / NCH), nch = min((nkg + perg - 1) / perg, NCH);
      for (int b = 0; b < B; ++b) {
        const int wph = GATW / heads;          // waves per head when the heads are gathered side by side (one poll round trip)
        if (wave < GATW) {
          if (wph * heads == GATW && (nch * (hd + 2) + wph - 1) / wph <= 64 * GQ) {
            const int h = wave / wph;
            gather_span(gr + b * gbs + G.part + h * NCH * (hd + 2), nch * (hd + 2), tag, wave - h * wph, wph, lane,
                        [&](int i, float x) { pm[h * NCH * (hd + 2) + i] = x; }, err, dead);
          } else {
            for (int h = 0; h < heads; ++h)
              gather_span(gr + b * gbs + G.part + h * NCH * (hd + 2), nch * (hd + 2), tag, wave, GATW, lane,
                          [&](int i, float x) { pm[h * NCH * (hd + 2) + i] = x; }, err, dead);
          }
        }
        lds_barrier();
        MPROF(17);
        if (tid < M2N) {      // (one batch of LDS reads: chunks beyond the filled ones re-read the last one with weight zero)
          const int h = tid / hd, d = tid - h * hd;
          const float* q = pm + h * NCH * (hd + 2);
          float M = -INFINITY, zt = 0.f, o = 0.f;
          for (int c0 = 0; c0 < nch; c0 += 8) {
            float mm[8], zz[8], oo[8];
#pragma unroll
            for (int c2 = 0; c2 < 8; ++c2) {
              const float* qc = q + min(c0 + c2, nch - 1) * (hd + 2);
              mm[c2] = qc[0]; zz[c2] = qc[1]; oo[c2] = qc[2 + d];
            }
            float Mn = M;
#pragma unroll
            for (int c2 = 0; c2 < 8; ++c2) Mn = fmaxf(Mn, mm[c2]);
            const float fo = __expf(M - Mn);           // (exp(-inf) = 0 on the first batch)
            zt *= fo; o *= fo;
#pragma unroll
            for (int c2 = 0; c2 < 8; ++c2) {
              const float f = (c0 + c2 < nch && zz[c2] > 0.f) ? __expf(mm[c2] - Mn) : 0.f;
              zt += f * zz[c2]; o += f * oo[c2];
            }
            M = Mn;
          }
          va[b * M2N + tid] = o / zt;
        }
        lds_barrier();
      }
    }
    MPROF(18);
    split_mul<NB>(wot, va, M2N, M2N, bt + 24, SATT_ACT_TANH, XK, M2N, gr + G.tr, gbs, tag, wg, B, rs, tid, sx);
    MPROF(19);
    gather_vec<NB>(gr + G.tr, gbs, M2N, tag, B, tid, err, dead, [&](int b, int i, float v) { vc[b * M2N + i] = v; });
    MPROF(20);
    // ================= G2: mel | stop projection (split) -> y, the next step's fed frame
    have_p0 = fold && !last;      // (the last step of a launch hands over through yout: the next launch starts unfolded)
    if (have_p0) {
      split_mul_fb<NB>(wou, wfh, wfl, vc, NO, P0, bt + 32, bt + 168, gr + G.y, gr + G.p0, gbs, tag, wg, rs, tid, sx);
      MPROF(21);
      // y (waves 0..2: NO <= 192) and the next step's p0 (waves 3..6) in ONE gather phase
      if (wave < 3) {
        const int beg = 64 * wave, cnt = min(64, NO - beg);
#pragma unroll
        for (int b = 0; b < NB; ++b) gather_poll<1>(gr + b * gbs + G.y + beg, cnt, tag, lane, [&](int i, float v) { yv[b * M2NO + beg + i] = v; }, err, dead);
      } else if (wave < 7) {
        const int beg = 64 * (wave - 3), cnt = min(64, P0 - beg);
#pragma unroll
        for (int b = 0; b < NB; ++b) gather_poll<1>(gr + b * gbs + G.p0 + beg, cnt, tag + 1u, lane, [&](int i, float v) { vb[b * M2N + beg + i] = v; }, err, dead);
      }
      lds_barrier();
    } else {
      split_mul<NB>(wou, vc, M2N, NO, bt + 32, SATT_ACT_NONE, nullptr, 0, gr + G.y, gbs, tag, wg, B, rs, tid, sx);
      MPROF(21);
      gather_vec<NB>(gr + G.y, gbs, NO, tag, B, tid, err, dead, [&](int b, int i, float v) { yv[b * M2NO + i] = v; });
    }
    MPROF(22);
    if (wg == 3 % M2G) {
      for (int i = tid; i < NB * NO; i += M2T) { const int b = i / NO, c = i - b * NO; if (b < B) p.yout[((int64_t)b * (p.Td + 1) + t + 1) * NO + c] = yv[b * M2NO + c]; }
    }
    // stop rule (StopTokenBasedInferenceHelper): every workgroup holds the same frame bits, so every workgroup takes the same
    // decision and leaves the step loop at the stop token - no step runs past it (the state hand-over below is skipped: the
    // utterance is over)
    bool fire = false;
    if (p.flag && !p.tin && t > p.min_steps) {
      fire = true;
      for (int b = 0; b < B; ++b) fire = fire && (1.f / (1.f + __expf(-yv[b * M2NO + NO - 1])) > p.stop_threshold);
    }
    if (wg == 0 && tid == 0) {
      if (fire) *p.flag = t + 1;
      *p.step = t + 1; p.step[1] = t + 1;
    }
    if (fire) break;
    if (last) {
      // hand-over to the launch-per-layer path: the contexts of the last step (buffer t & 1), columns c = tid >> 4 of the own share
      const int cw = (CT + M2G - 1) / M2G, c = wg * cw + (tid >> 4), rp = tid & 15;
      if ((tid >> 4) < cw && c < CT) {
        const bool m1 = c < V1;
        for (int b = 0; b < B; ++b) {
          const float* vs = m1 ? p.values1 + (int64_t)b * Ti * V1 + c : p.values2 + (int64_t)b * Ti * V2 + (c - V1);
          const float* al = (m1 ? e1 : e2) + b * M2TI;
          float acc = 0.f;
          for (int r = rp; r < Ti; r += 16) acc += al[r] * vs[(int64_t)r * (m1 ? V1 : V2)];
          SATT_DPP_ADD(acc, 0xB1); SATT_DPP_ADD(acc, 0x4E); SATT_DPP_ADD(acc, 0x141); SATT_DPP_ADD(acc, 0x140);
          if (rp == 0) p.ctx[((int64_t)par * B + b) * CT + c] = acc;
        }
      }
    }
    MPROF(23);
#ifdef SATT_MEGA_PROF
    if (wg == satt_mega2_prof_wg && threadIdx.x == 0) { const unsigned long long c_ = clock64(); satt_mega2_prof[30] += c_ - mp_clk; mp_clk = c_; }
#endif
  }
}


}  // namespace

#ifdef SATT_MEGA_PROF
extern "C" int satt_dec_mega2_prof_select(int wg) {
  return hipMemcpyToSymbol(HIP_SYMBOL(satt_mega2_prof_wg), &wg, sizeof(int)) == hipSuccess ? 0 : -3;
}
extern "C" int satt_dec_mega2_prof_read(unsigned long long* host32, int reset) {
  if (hipMemcpyFromSymbol(host32, HIP_SYMBOL(satt_mega2_prof), sizeof(unsigned long long) * 32) != hipSuccess) return -3;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(satt_mega2_prof), z, sizeof(z)) != hipSuccess) return -3; }
  return 0;
}
#endif

// floats of the exchange buffer `part` of satt_dec_mega_params (two floats per granule)
extern "C" int64_t satt_dec_mega_scratch_floats(int B, int heads, int hd) {
  if (heads < 1 || B < 1 || heads * hd != M2N) return 0;
  return 2 * (int64_t)B * gl_of(hd).total;
}

// shapes the kernel takes (pointers are checked by satt_dec_mega)
extern "C" int satt_dec_mega_supported(const satt_dec_mega_params* pp) {
  if (!pp) return 0;
  const satt_dec_mega_params& p = *pp;
  const int hd = p.heads > 0 ? M2N / p.heads : 0, UQ = p.U1 + p.U2, CT = p.V1 + p.V2;
  return p.B >= 1 && p.B <= 2 && p.Td >= 1 && p.Ti >= 1 && p.Ti <= M2TI && (p.Ti + M2G - 1) / M2G <= 8 &&
         p.A == M2N && p.D == M2N && p.Ds == M2N && p.heads >= 2 && M2G % p.heads == 0 && p.heads * hd == M2N &&
         hd >= 16 && hd <= M2HD && hd % 16 == 0 && M2T % (hd / 4) == 0 && M2T / (hd / 4) <= 32 && (hd / 16 == 4 || hd / 16 == 8) &&
         p.U1 >= 1 && p.U1 % 4 == 0 && p.U2 >= 1 && p.U2 <= 64 && UQ <= M2N && UQ % 8 == 0 && p.V1 >= 1 && p.V2 >= 1 && CT <= 32 * M2G &&
         p.P0 >= 8 && p.P0 <= M2N && p.P0 % 8 == 0 && p.P1 >= 8 && p.P1 <= M2N && p.P1 % 8 == 0 &&
         p.feed >= 1 && p.feed <= M2N && p.feed + 1 <= p.NO && p.NO <= M2NO && p.ldout % 8 == 0 && p.ldout >= p.NO &&
         p.kernel >= 1 && p.kernel <= 16 && p.filters >= 1 && p.filters <= 8 &&
         mega2_lds_bytes(p.B <= 1 ? 1 : 2, p.Ti) <= 160 * 1024;
}

extern "C" int satt_dec_mega(const satt_dec_mega_params* pp, void* stream) {
  if (!pp || !satt_dec_mega_supported(pp) || pp->nsteps < 1) return SATT_E_UNSUPPORTED;
  const satt_dec_mega_params& p = *pp;
  if (!p.Wp0 || !p.Wp1 || !p.Wa || !p.Wq || !p.W1 || !p.W2 || !p.Wkvq || !p.Wot || !p.Wout || !p.bp0 || !p.bp1 || !p.ba || !p.b1l ||
      !p.b2l || !p.bkvq || !p.bot || !p.bout || !p.locF || !p.locFb || !p.locU || !p.v1 || !p.b1 || !p.v2 || !p.lengths || !p.keys1 ||
      !p.values1 || !p.keys2 || !p.values2 || !p.ca || !p.ha || !p.c1 || !p.h1 || !p.c2 || !p.h2 || !p.a_state || !p.alpha_state ||
      !p.ctx || !p.yout || !p.align1 || !p.align2 || !p.kvq || !p.part || !p.ctab || !p.step || !p.err) return SATT_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const int NB = p.B <= 1 ? 1 : 2;
  const size_t smem = mega2_lds_bytes(NB, p.Ti);
  // one XCD (grid 8 x 32, every eighth workgroup works: see the kernel) unless SATT_DECODE_ONE_XCD=0
  static const int spread = [] { const char* e = getenv("SATT_DECODE_ONE_XCD"); return (e && atoi(e) == 0) ? 1 : 8; }();
  const bool lj = p.U1 == 224 && p.U2 == 32 && p.V1 == 256 && p.V2 == 32 && p.heads == 2 && p.NO == 161 && p.feed == 80 && p.P0 == 256 &&
                  p.P1 == 128 && p.kernel == 10 && p.filters == 5 && getenv("SATT_DECODE_GENERIC") == nullptr;
#define SATT_MEGA2_(NBV, TR, LJV)                                                                                          \
  do {                                                                                                                       \
    if (hipFuncSetAttribute((const void*)dec_mega2_k<NBV, TR, LJV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) { \
      (void)hipGetLastError();                                                                                               \
      return SATT_E_LAUNCH;                                                                                                  \
    }                                                                                                                        \
    hipLaunchKernelGGL((dec_mega2_k<NBV, TR, LJV>), dim3(M2G * spread), dim3(M2T), smem, s, p, spread);                      \
  } while (0)
#define SATT_MEGA2(NBV, TR) do { if (lj) SATT_MEGA2_(NBV, TR, true); else SATT_MEGA2_(NBV, TR, false); } while (0)
  if (NB == 1 && p.Ti <= M2TR) SATT_MEGA2(1, true);
  else if (NB == 1) SATT_MEGA2(1, false);
  else SATT_MEGA2(2, false);
#undef SATT_MEGA2_
#undef SATT_MEGA2
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}
