// Small fused / memory-bound kernels of the training path (HBM- or latency-bound: coalesced row-major access,
// one pass per tensor wherever the math allows it).
#include "common.h"
#include <algorithm>

namespace {

constexpr int EW_NT = 256;
inline int ew_blocks(int64_t n, int per = EW_NT) { return (int)std::min<int64_t>((n + per - 1) / per, 256 * 8 * 4); }

// ---------------------------------------------------------------- embedding
__global__ void embedding_fwd_k(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                float* __restrict__ out, int n, int dim, int offset) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)n * dim;
       e += (int64_t)gridDim.x * blockDim.x) {
    int i = (int)(e / dim), c = (int)(e - (int64_t)i * dim);
    out[e] = table[(ids[i] - offset) * dim + c];
  }
}
// A thread owns one column of EMB_RUN CONSECUTIVE tokens and merges runs of equal ids before it touches the table: the padding
// symbol fills the tail of every sequence (a third of a padded batch) and its row used to collect one atomic per padded token
// and column - ~1700 serialised atomics per address, 37 us at the very end of the step; merged, the hot row sees an eighth.
// All loads are requested before the first use (ids and gradients of the whole run).
constexpr int EMB_RUN = 8;
__global__ void embedding_bwd_k(const int64_t* __restrict__ ids, const float* __restrict__ dout,
                                float* __restrict__ dtable, int n, int dim, int offset) {
  const int64_t groups = (n + EMB_RUN - 1) / EMB_RUN;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < groups * dim; e += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(e / dim), c = (int)(e - (int64_t)g * dim), i0 = g * EMB_RUN;
    int64_t id[EMB_RUN]; float v[EMB_RUN];
#pragma unroll
    for (int r = 0; r < EMB_RUN; ++r) {
      const int i = min(i0 + r, n - 1);
      id[r] = ids[i]; v[r] = dout[(int64_t)i * dim + c];
    }
    int64_t cur = id[0]; float acc = v[0];
#pragma unroll
    for (int r = 1; r < EMB_RUN; ++r) {
      if (i0 + r >= n) break;
      if (id[r] == cur) acc += v[r];
      else { atomicAdd(&dtable[(cur - offset) * dim + c], acc); cur = id[r]; acc = v[r]; }
    }
    atomicAdd(&dtable[(cur - offset) * dim + c], acc);
  }
}

// Deterministic form for small tables (the 256-symbol text embedding: 5120 tokens hit 256 rows, so the atomic form above
// serialises ~20 atomics per address - 37 us at the very end of the step - and sums them in a run-dependent order): one
// workgroup per TABLE ROW collects the tokens of that row in ascending order (all ids of a thread requested at once, ballot +
// prefix compaction into LDS) and adds their gradient rows in that order.  n <= 256 * EB_MAXI tokens.
constexpr int EB_NT = 256, EB_MAXI = 32;
__global__ __launch_bounds__(EB_NT) void embedding_bwd_rows_k(const int64_t* __restrict__ ids, const float* __restrict__ dout,
                                                             float* __restrict__ dtable, int n, int dim, int offset) {
  extern __shared__ int hits[];                 // [n] token indices of this row, ascending
  __shared__ int wbase[EB_NT / 64 + 1];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // thread t owns the CONTIGUOUS tokens [t * per, (t + 1) * per): thread-major order is ascending token order, so one block-wide
  // prefix sum of the per-thread hit counts places every hit (no pass per chunk)
  const int per = (n + EB_NT - 1) / EB_NT;      // <= EB_MAXI
  unsigned mask = 0u;
#pragma unroll
  for (int q = 0; q < EB_MAXI; ++q) {           // branch-free, clamped; all loads in flight
    const int i = tid * per + q;
    const int idv = (int)(ids[min(i, n - 1)] - offset);
    mask |= (q < per && i < n && idv == row) ? (1u << q) : 0u;
  }
  const int cnt = __popc(mask);
  int incl = cnt;                               // inclusive prefix over the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
  if (lane == 63) wbase[wave + 1] = incl;
  __syncthreads();
  if (tid == 0) { wbase[0] = 0; for (int w = 1; w <= EB_NT / 64; ++w) wbase[w] += wbase[w - 1]; }
  __syncthreads();
  int pos = wbase[wave] + incl - cnt;
  const int nh = wbase[EB_NT / 64];
  if (nh == 0) return;
  for (unsigned m = mask; m; m &= m - 1) hits[pos++] = tid * per + (__ffs(m) - 1);
  __syncthreads();
  for (int c = tid; c < dim; c += EB_NT) {
    // eight independent partial sums (a fixed pattern: the result does not depend on timing), combined in a fixed order
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int h = 0;
    for (; h + 8 <= nh; h += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] += dout[(int64_t)hits[h + u] * dim + c];
    }
    for (int u = 0; h < nh; ++h, ++u) p[u] += dout[(int64_t)hits[h] * dim + c];
    dtable[(int64_t)row * dim + c] += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  }
}

// ---------------------------------------------------------------- activation backward
__device__ __forceinline__ float act_grad(int act, float y) {
  if (act == SATT_ACT_RELU) return y != 0.f ? 1.f : 0.f;   // y is post-relu(-dropout): y==0 <=> no gradient
  if (act == SATT_ACT_TANH) return 1.f - y * y;
  if (act == SATT_ACT_SIGMOID) return y * (1.f - y);
  if (act == SATT_ACT_SOFTSIGN) { const float u = 1.f - fabsf(y); return u * u; }   // y = x/(1+|x|)
  return 1.f;
}
__global__ void act_bwd_k(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ y, int64_t ldy,
                          float* __restrict__ dx, int64_t lddx, int rows, int cols, int act, float scale) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)rows * cols;
       e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e / cols), c = (int)(e - (int64_t)r * cols);
    dx[r * lddx + c] = dy[r * lddy + c] * act_grad(act, y[r * ldy + c]) * scale;
  }
}

// the same with the activation output given as (z - res): the producer wrote z = act(u) + res in one pass (GEMM epilogue with a
// residual) and did not keep act(u) - recovered here to within one rounding of z
__global__ void act_bwd_res_k(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ z, int64_t ldz,
                              const float* __restrict__ res, int64_t ldres, float* __restrict__ dx, int64_t lddx, int rows,
                              int cols, int act, float scale) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)rows * cols;
       e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e / cols), c = (int)(e - (int64_t)r * cols);
    dx[r * lddx + c] = dy[r * lddy + c] * act_grad(act, z[r * ldz + c] - res[r * ldres + c]) * scale;
  }
}

// ---------------------------------------------------------------- batch norm
constexpr int BN_ROWS = 128;  // rows per chunk
inline int bn_chunks(int rows) { return (rows + BN_ROWS - 1) / BN_ROWS; }

// per (chunk, column): chunk mean and M2 in ONE pass over the chunk (sum and sum of squares of the values SHIFTED by the chunk's
// first row of that column: with the shift the 128-row fp32 sums keep M2 = ss - s^2 / n accurate whatever the column's mean -
// the two-pass form read the 42 MB bank twice)
__global__ __launch_bounds__(256) void bn_partial_k(const float* __restrict__ x, int64_t ldx, float* __restrict__ ws,
                                                    int rows, int C) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * BN_ROWS, r1 = min(rows, r0 + BN_ROWS);
  float s = 0.f, ss = 0.f, shift = 0.f;
  if (c < C) {
    shift = x[(int64_t)r0 * ldx + c];
#pragma unroll 8
    for (int r = r0 + rl; r < r1; r += 4) { const float d = x[(int64_t)r * ldx + c] - shift; s += d; ss += d * d; }
  }
  red[0][rl][cl] = s; red[1][rl][cl] = ss;
  __syncthreads();
  if (rl == 0 && c < C) {
    const float n = (float)(r1 - r0);
    const float st = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
    const float sst = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    ws[((int64_t)blockIdx.y * 2 + 0) * C + c] = shift + st / n;
    ws[((int64_t)blockIdx.y * 2 + 1) * C + c] = fmaxf(sst - st * st / n, 0.f);
  }
}
// merge of the per-chunk (mean, M2) pairs (Chan et al.), 64 channels x 4 chunk groups per workgroup: the 2 x nchunk
// dependent loads of a channel are split four ways and meet in LDS (a single thread per channel made this tiny kernel
// one of the longest of the encoder: 20 us)
__global__ __launch_bounds__(256) void bn_finalize_k(const float* __restrict__ ws, int nchunk, int rows, int C, float eps,
                                                     float momentum, float* __restrict__ mean_o,
                                                     float* __restrict__ rstd_o, float* __restrict__ mmean,
                                                     float* __restrict__ mvar) {
  __shared__ double red[4][64];
  const int cl = threadIdx.x & 63, kg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const bool ok = c < C;
  double tot = 0.0;
  if (ok) for (int k = kg; k < nchunk; k += 4) {
    const int n = min(BN_ROWS, rows - k * BN_ROWS);
    tot += (double)n * ws[((int64_t)k * 2) * C + c];
  }
  red[kg][cl] = tot;
  __syncthreads();
  const double mean = (red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]) / rows;
  __syncthreads();
  double m2 = 0.0;
  if (ok) for (int k = kg; k < nchunk; k += 4) {
    const int n = min(BN_ROWS, rows - k * BN_ROWS);
    const double d = (double)ws[((int64_t)k * 2) * C + c] - mean;
    m2 += (double)ws[((int64_t)k * 2 + 1) * C + c] + n * d * d;
  }
  red[kg][cl] = m2;
  __syncthreads();
  if (kg == 0 && ok) {
    m2 = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    const double var = m2 / rows;
    mean_o[c] = (float)mean;
    rstd_o[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (mmean) mmean[c] = momentum * mmean[c] + (1.f - momentum) * (float)mean;
    if (mvar) mvar[c] = momentum * mvar[c] + (1.f - momentum) * (float)(rows > 1 ? m2 / (rows - 1) : var);
  }
}
__device__ __forceinline__ float apply_act(int act, float v) {
  if (act == SATT_ACT_RELU) return fmaxf(v, 0.f);
  if (act == SATT_ACT_TANH) return tanhf(v);
  if (act == SATT_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  if (act == SATT_ACT_SOFTSIGN) return v / (1.f + fabsf(v));
  return v;
}
__global__ void bn_apply_k(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                           const float* __restrict__ beta, const float* __restrict__ mean,
                           const float* __restrict__ rstd, float* __restrict__ y, int64_t ldy, int rows, int C,
                           int act) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)rows * C;
       e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e / C), c = (int)(e - (int64_t)r * C);
    float v = (x[(int64_t)r * ldx + c] - mean[c]) * rstd[c] * gamma[c] + beta[c];
    y[(int64_t)r * ldy + c] = apply_act(act, v);
  }
}
__global__ void bn_infer_k(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                           const float* __restrict__ beta, const float* __restrict__ mmean,
                           const float* __restrict__ mvar, float* __restrict__ y, int64_t ldy, int rows, int C,
                           float eps, int act) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)rows * C;
       e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e / C), c = (int)(e - (int64_t)r * C);
    float v = (x[(int64_t)r * ldx + c] - mmean[c]) * rsqrtf(mvar[c] + eps) * gamma[c] + beta[c];
    y[(int64_t)r * ldy + c] = apply_act(act, v);
  }
}
__device__ __forceinline__ float bn_dyp(int act, float dy, float ybn) {
  if (act == SATT_ACT_RELU) return ybn > 0.f ? dy : 0.f;
  if (act == SATT_ACT_TANH) { float t = tanhf(ybn); return dy * (1.f - t * t); }
  if (act == SATT_ACT_SIGMOID) { float s = 1.f / (1.f + expf(-ybn)); return dy * s * (1.f - s); }
  return dy;
}
__global__ __launch_bounds__(256) void bn_bwd_partial_k(const float* __restrict__ dy, int64_t lddy,
                                                        const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, float* __restrict__ ws,
                                                        int rows, int C, int act) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * BN_ROWS, r1 = min(rows, r0 + BN_ROWS);
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
    const float mu = mean[c], rs = rstd[c], g = gamma[c], bt = beta[c];
#pragma unroll 8
    for (int r = r0 + rl; r < r1; r += 4) {
      float xh = (x[(int64_t)r * ldx + c] - mu) * rs;
      float d = bn_dyp(act, dy[(int64_t)r * lddy + c], g * xh + bt);
      s1 += d; s2 += d * xh;
    }
  }
  red[0][rl][cl] = s1; red[1][rl][cl] = s2;
  __syncthreads();
  if (rl == 0 && c < C) {
    ws[((int64_t)blockIdx.y * 2 + 0) * C + c] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    ws[((int64_t)blockIdx.y * 2 + 1) * C + c] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
  }
}
__global__ __launch_bounds__(256) void bn_bwd_finalize_k(float* __restrict__ ws, int nchunk, int C,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, kg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s1 = 0.f, s2 = 0.f;
  if (c < C) for (int k = kg; k < nchunk; k += 4) { s1 += ws[((int64_t)k * 2) * C + c]; s2 += ws[((int64_t)k * 2 + 1) * C + c]; }
  red[0][kg][cl] = s1; red[1][kg][cl] = s2;
  __syncthreads();
  if (kg == 0 && c < C) {
    s1 = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
    s2 = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    ws[((int64_t)nchunk * 2) * C + c] = s1;
    ws[((int64_t)nchunk * 2 + 1) * C + c] = s2;
    if (dbeta) dbeta[c] += s1;
    if (dgamma) dgamma[c] += s2;
  }
}
__global__ void bn_bwd_apply_k(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ldx,
                               const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ rstd,
                               const float* __restrict__ tot, float* __restrict__ dx, int64_t lddx, int rows, int C,
                               int act) {
  const float inv = 1.f / (float)rows;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)rows * C;
       e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e / C), c = (int)(e - (int64_t)r * C);
    float xh = (x[(int64_t)r * ldx + c] - mean[c]) * rstd[c];
    float d = bn_dyp(act, dy[(int64_t)r * lddy + c], gamma[c] * xh + beta[c]);
    dx[(int64_t)r * lddx + c] = gamma[c] * rstd[c] * (d - tot[c] * inv - xh * tot[C + c] * inv);
  }
}

// ---------------------------------------------------------------- batch norm, one launch per direction
// The three-launch forms above cost 18 us (forward) / 35 us (backward) plus two launch gaps for a 2.6 MB activation: the
// kernels are launch- and latency-bound, not bandwidth-bound.  Here every workgroup (64 channels x FBN_ROWS rows) writes its
// partial statistics, the LAST one to arrive of a channel group merges them and raises a flag, and every workgroup then
// normalises its own rows (its second read of them comes from L2).  All workgroups of a channel group must be resident while
// they wait: the host only takes this path for small grids (<= FBN_MAX_WGS).  sync: 2 words per channel group {arrivals,
// flag + departures}, zero before the first launch and zero again when the launch has finished.
constexpr int FBN_ROWS = 64;
constexpr int FBN_MAX_WGS = 256;
constexpr int FBN_KMAX = 24;     // partial results per merging thread (4 threads per channel): at most 96 chunks = 6144 rows
inline int fbn_chunks(int rows) { return (rows + FBN_ROWS - 1) / FBN_ROWS; }

// No fences anywhere: an agent-scope release / acquire fence writes back / invalidates the WHOLE L2 of the XCD (measured: the
// fenced version of these kernels cost the step +0.22 ms - every later kernel of the encoder started on a cold L2).  Instead
// everything another workgroup reads is written with agent-scope (write-through) stores and read with agent-scope loads, and
// a wave waits for its own stores (s_waitcnt vmcnt(0)) before it signals - the protocol of csrc/cluster_xchg.h.
__device__ __forceinline__ float fbn_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fbn_st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fbn_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// arrival at the group's barrier (wave 0 has written the workgroup's partial results with fbn_st); true for the last workgroup
__device__ __forceinline__ bool fbn_arrive(uint32_t* sync, int nchunk) {
  __shared__ int s_last;
  if (threadIdx.x < 64) fbn_drain();
  if (threadIdx.x == 0)
    s_last = __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)(nchunk - 1);
  __syncthreads();
  return s_last != 0;
}
// the last workgroup publishes (wave 0 has written the merged results with fbn_st); everyone waits (bounded), leaves, and the
// last to leave clears the words.  false on a timeout.
__device__ __forceinline__ bool fbn_release_and_wait(uint32_t* sync, int nchunk, bool last) {
  __shared__ int s_ok;
  if (threadIdx.x < 64 && last) fbn_drain();
  if (threadIdx.x == 0) {
    if (last) {
      __hip_atomic_store(&sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int ok = 0;
    for (unsigned i = 0; i < (1u << 22); ++i) {
      if (__hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 1; break; }
      __builtin_amdgcn_s_sleep(4);
    }
    s_ok = ok;
    if (__hip_atomic_fetch_add(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)nchunk)
      __hip_atomic_store(&sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return s_ok != 0;
}

__global__ __launch_bounds__(256) void bn_fwd_fused_k(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ y, int64_t ldy,
                                                      float* mean_o, float* rstd_o, float* __restrict__ mmean,
                                                      float* __restrict__ mvar, float* ws, uint32_t* sync_all, int rows, int C,
                                                      float eps, float momentum, int act, int nchunk) {
  __shared__ float red[2][4][64];
  __shared__ double dred[4][64];
  const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * FBN_ROWS, r1 = min(rows, r0 + FBN_ROWS);
  const bool ok = c < C;
  const int cc = ok ? c : C - 1;
  uint32_t* sync = sync_all + 2 * blockIdx.x;
  constexpr int NR = FBN_ROWS / 4;
  float xv[NR];
  {   // own rows into registers (kept for the normalisation); statistics shifted by the chunk's first row (see bn_partial_k)
    const float shift = x[(int64_t)r0 * ldx + cc];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int r = r0 + rl + 4 * i;
      xv[i] = x[(int64_t)min(r, r1 - 1) * ldx + cc];
      const float d = r < r1 ? xv[i] - shift : 0.f;
      s += d; ss += d * d;
    }
    red[0][rl][cl] = s; red[1][rl][cl] = ss;
    __syncthreads();
    if (rl == 0 && ok) {
      const float n = (float)(r1 - r0);
      const float st = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
      const float sst = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
      fbn_st(ws + ((int64_t)blockIdx.y * 2 + 0) * C + c, shift + st / n);
      fbn_st(ws + ((int64_t)blockIdx.y * 2 + 1) * C + c, fmaxf(sst - st * st / n, 0.f));
    }
  }
  const bool last = fbn_arrive(sync, nchunk);
  if (last) {     // merge of the per-chunk (mean, M2) pairs (Chan et al.), as bn_finalize_k; every partial of a thread is
    //               requested before the first is used (agent-scope loads: one memory round trip, not one per chunk)
    float pm[FBN_KMAX], pv[FBN_KMAX];
#pragma unroll
    for (int i = 0; i < FBN_KMAX; ++i) {
      const int k = min(rl + 4 * i, nchunk - 1);
      pm[i] = fbn_ld(ws + ((int64_t)k * 2) * C + cc); pv[i] = fbn_ld(ws + ((int64_t)k * 2 + 1) * C + cc);
    }
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < FBN_KMAX; ++i) {
      const int k = rl + 4 * i;
      if (k < nchunk) tot += (double)min(FBN_ROWS, rows - k * FBN_ROWS) * pm[i];
    }
    dred[rl][cl] = tot;
    __syncthreads();
    const double mean = (dred[0][cl] + dred[1][cl] + dred[2][cl] + dred[3][cl]) / rows;
    __syncthreads();
    double m2 = 0.0;
#pragma unroll
    for (int i = 0; i < FBN_KMAX; ++i) {
      const int k = rl + 4 * i;
      if (k < nchunk) {
        const int n = min(FBN_ROWS, rows - k * FBN_ROWS);
        const double d = (double)pm[i] - mean;
        m2 += (double)pv[i] + n * d * d;
      }
    }
    dred[rl][cl] = m2;
    __syncthreads();
    if (rl == 0 && ok) {
      m2 = dred[0][cl] + dred[1][cl] + dred[2][cl] + dred[3][cl];
      const double var = m2 / rows;
      fbn_st(mean_o + c, (float)mean);
      fbn_st(rstd_o + c, (float)(1.0 / sqrt(var + (double)eps)));
      if (mmean) mmean[c] = momentum * mmean[c] + (1.f - momentum) * (float)mean;
      if (mvar) mvar[c] = momentum * mvar[c] + (1.f - momentum) * (float)(rows > 1 ? m2 / (rows - 1) : var);
    }
  }
  const bool fine = fbn_release_and_wait(sync, nchunk, last);
  const float mu = fbn_ld(mean_o + cc), rs = fine ? fbn_ld(rstd_o + cc) : __builtin_nanf("");
  const float g = gamma[cc] * rs, bt = beta[cc];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int r = r0 + rl + 4 * i;
    if (r < r1 && ok) y[(int64_t)r * ldy + c] = apply_act(act, (xv[i] - mu) * g + bt);
  }
}

__global__ __launch_bounds__(256) void bn_bwd_fused_k(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x,
                                                      int64_t ldx, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, float* __restrict__ dx, int64_t lddx,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, float* ws,
                                                      uint32_t* sync_all, int rows, int C, int act, int nchunk) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * FBN_ROWS, r1 = min(rows, r0 + FBN_ROWS);
  const bool ok = c < C;
  const int cc = ok ? c : C - 1;
  uint32_t* sync = sync_all + 2 * blockIdx.x;
  constexpr int NR = FBN_ROWS / 4;
  const float mu = mean[cc], rs = rstd[cc], g = gamma[cc], bt = beta[cc];
  float xh[NR], dv[NR];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int r = r0 + rl + 4 * i, rc = min(r, r1 - 1);
    xh[i] = (x[(int64_t)rc * ldx + cc] - mu) * rs;
    dv[i] = dy[(int64_t)rc * lddy + cc];
  }
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int r = r0 + rl + 4 * i;
    dv[i] = r < r1 ? bn_dyp(act, dv[i], g * xh[i] + bt) : 0.f;
    s1 += dv[i]; s2 += dv[i] * xh[i];
  }
  red[0][rl][cl] = s1; red[1][rl][cl] = s2;
  __syncthreads();
  if (rl == 0 && ok) {
    fbn_st(ws + ((int64_t)blockIdx.y * 2 + 0) * C + c, red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl]);
    fbn_st(ws + ((int64_t)blockIdx.y * 2 + 1) * C + c, red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl]);
  }
  const bool last = fbn_arrive(sync, nchunk);
  float* tot = ws + (int64_t)nchunk * 2 * C;
  if (last) {
    float pa[FBN_KMAX], pb[FBN_KMAX];
#pragma unroll
    for (int i = 0; i < FBN_KMAX; ++i) {
      const int k = min(rl + 4 * i, nchunk - 1);
      pa[i] = fbn_ld(ws + ((int64_t)k * 2) * C + cc); pb[i] = fbn_ld(ws + ((int64_t)k * 2 + 1) * C + cc);
    }
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int i = 0; i < FBN_KMAX; ++i)
      if (rl + 4 * i < nchunk) { t1 += pa[i]; t2 += pb[i]; }
    red[0][rl][cl] = t1; red[1][rl][cl] = t2;
    __syncthreads();
    if (rl == 0 && ok) {
      t1 = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
      t2 = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
      fbn_st(tot + c, t1); fbn_st(tot + C + c, t2);
      if (dbeta) dbeta[c] += t1;
      if (dgamma) dgamma[c] += t2;
    }
  }
  const bool fine = fbn_release_and_wait(sync, nchunk, last);
  const float inv = fine ? 1.f / (float)rows : __builtin_nanf("");
  const float t1 = fbn_ld(tot + cc) * inv, t2 = fbn_ld(tot + C + cc) * inv, gr = g * rs;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int r = r0 + rl + 4 * i;
    if (r < r1 && ok) dx[(int64_t)r * lddx + c] = gr * (dv[i] - t1 - xh[i] * t2);
  }
}

// ---------------------------------------------------------------- maxpool (2, stride 1, SAME) over time
__global__ void maxpool_fwd_k(const float* __restrict__ x, float* __restrict__ y, int B, int T, int C) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)B * T * C;
       e += (int64_t)gridDim.x * blockDim.x) {
    int t = (int)((e / C) % T);
    float a = x[e];
    y[e] = (t + 1 < T) ? fmaxf(a, x[e + C]) : a;
  }
}
// dx[t] = dy[t]*[x[t] >= x[t+1]] + dy[t-1]*[x[t] > x[t-1]]  (ties go to the first element of the window)
__global__ void maxpool_bwd_k(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
                              int B, int T, int C) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)B * T * C;
       e += (int64_t)gridDim.x * blockDim.x) {
    int t = (int)((e / C) % T);
    float a = x[e], g = 0.f;
    if (t + 1 >= T || a >= x[e + C]) g += dy[e];
    if (t > 0 && a > x[e - C]) g += dy[e - C];
    dx[e] = g;
  }
}

// four channels per thread, 32-bit index arithmetic (the scalar form spends two 64-bit divisions per element and is VALU bound:
// 46 us for the 5120 x 2048 bank against ~25 us of memory time); neighbour rows are read at clamped addresses and masked
__global__ void maxpool_bwd4_k(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, int rows,
                               int T, int C4) {
  const int n = rows * C4;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const int row = e / C4, t = row % T;
    const bool has_next = t + 1 < T, has_prev = t > 0;
    const float4* xp = reinterpret_cast<const float4*>(x) + e;
    const float4* dp = reinterpret_cast<const float4*>(dy) + e;
    const float4 a = xp[0], an = xp[has_next ? C4 : 0], ap = xp[has_prev ? -C4 : 0];
    const float4 g0 = dp[0], gp = dp[has_prev ? -C4 : 0];
    float4 g;
    g.x = ((!has_next || a.x >= an.x) ? g0.x : 0.f) + ((has_prev && a.x > ap.x) ? gp.x : 0.f);
    g.y = ((!has_next || a.y >= an.y) ? g0.y : 0.f) + ((has_prev && a.y > ap.y) ? gp.y : 0.f);
    g.z = ((!has_next || a.z >= an.z) ? g0.z : 0.f) + ((has_prev && a.z > ap.z) ? gp.z : 0.f);
    g.w = ((!has_next || a.w >= an.w) ? g0.w : 0.f) + ((has_prev && a.w > ap.w) ? gp.w : 0.f);
    reinterpret_cast<float4*>(dx)[e] = g;
  }
}

// ---------------------------------------------------------------- batch norm + activation + max-pool in one pass (conv bank)
// y = act(bn(x)) is never stored: the forward kernel writes max(y[t], y[t+1]) only, the backward kernels recompute y from x with the
// SAME expression (bn_act: the tie decisions of the pooling compare bit-identical values in both directions).
__device__ __forceinline__ float bn_pre(float x, float mean, float rstd, float g, float b) { return (x - mean) * rstd * g + b; }
// A thread owns 4 channels of BNP_ROWS consecutive rows and carries the activated next row along: BNP_ROWS + 1 row reads for BNP_ROWS
// outputs (r6; one output per thread read every row twice).
constexpr int BNP_ROWS = 8;
__global__ void bn_apply_maxpool4_k(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ mp,
                                    int rows, int T, int C4, int act) {
  const int nrb = (rows + BNP_ROWS - 1) / BNP_ROWS, n = nrb * C4;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const int rb = e / C4, c4 = e - rb * C4, ra = rb * BNP_ROWS, rbnd = min(rows, ra + BNP_ROWS);
    const float4 m = reinterpret_cast<const float4*>(mean)[c4], r = reinterpret_cast<const float4*>(rstd)[c4];
    const float4 g = reinterpret_cast<const float4*>(gamma)[c4], b = reinterpret_cast<const float4*>(beta)[c4];
    const float4* xp = reinterpret_cast<const float4*>(x) + (int64_t)ra * C4 + c4;
    float4* op = reinterpret_cast<float4*>(mp) + (int64_t)ra * C4 + c4;
    auto act4 = [&](const float4 a) {
      float4 y;
      y.x = apply_act(act, bn_pre(a.x, m.x, r.x, g.x, b.x)); y.y = apply_act(act, bn_pre(a.y, m.y, r.y, g.y, b.y));
      y.z = apply_act(act, bn_pre(a.z, m.z, r.z, g.z, b.z)); y.w = apply_act(act, bn_pre(a.w, m.w, r.w, g.w, b.w));
      return y;
    };
    float4 xv[BNP_ROWS + 1];
#pragma unroll
    for (int k = 0; k <= BNP_ROWS; ++k) xv[k] = xp[(int64_t)min(k, max(rows - 1 - ra, 0)) * C4];      // (clamped: all loads in flight at once)
    float4 y = act4(xv[0]);
#pragma unroll
    for (int k = 0; k < BNP_ROWS; ++k) {
      const float4 yn = act4(xv[k + 1]);
      if (ra + k < rbnd) {
        const bool has_next = (ra + k) % T + 1 < T;
        float4 o;
        o.x = has_next ? fmaxf(y.x, yn.x) : y.x; o.y = has_next ? fmaxf(y.y, yn.y) : y.y;
        o.z = has_next ? fmaxf(y.z, yn.z) : y.z; o.w = has_next ? fmaxf(y.w, yn.w) : y.w;
        op[(int64_t)k * C4] = o;
      }
      y = yn;
    }
  }
}
// backward, pass 1: d y[t] from d mp (ties go to the first element of the window, as maxpool_bwd_k), through the activation, the
// per-chunk sums of the BatchNorm backward, and d (the gradient behind the activation) written for pass 2
__global__ __launch_bounds__(256) void maxpool_bn_bwd_partial_k(const float* __restrict__ dmp, const float* __restrict__ x,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               float* __restrict__ dbuf, float* __restrict__ ws, int rows, int T,
                                                               int C, int act) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * BN_ROWS, r1 = min(rows, r0 + BN_ROWS);
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
    const float mu = mean[c], rs = rstd[c], g = gamma[c], bt = beta[c];
    // a lane walks a quarter of the chunk's rows IN ORDER and carries the window with it: per row one new x (the next row's) and one
    // new d mp - the interleaved walk (rows rl, rl + 4, ...) fetched x three times and d mp twice per element (r6: five loads -> two)
    const int q = (BN_ROWS + 3) / 4, ra = r0 + rl * q, rb = min(r1, ra + q);
    if (ra < rb) {
      const float* xc = x + c;
      const float* gc = dmp + c;
      float xq = x[(int64_t)max(ra - 1, 0) * C + c], xt = xc[(int64_t)ra * C], gp = dmp[(int64_t)max(ra - 1, 0) * C + c];
      float yq = apply_act(act, bn_pre(xq, mu, rs, g, bt)), vt = bn_pre(xt, mu, rs, g, bt), yt = apply_act(act, vt);
#pragma unroll 4
      for (int r = ra; r < rb; ++r) {
        const int t = r % T;
        const bool has_next = t + 1 < T, has_prev = t > 0;
        const float xn = xc[(int64_t)min(r + 1, rows - 1) * C], g0 = gc[(int64_t)r * C];
        const float vn = bn_pre(xn, mu, rs, g, bt), yn = apply_act(act, vn);
        const float dy = ((!has_next || yt >= yn) ? g0 : 0.f) + ((has_prev && yt > yq) ? gp : 0.f);
        const float d = bn_dyp(act, dy, vt);
        const float xh = (xt - mu) * rs;
        s1 += d; s2 += d * xh;
        dbuf[(int64_t)r * C + c] = d;
        xt = xn; yq = yt; vt = vn; yt = yn; gp = g0;
      }
    }
  }
  red[0][rl][cl] = s1; red[1][rl][cl] = s2;
  __syncthreads();
  if (rl == 0 && c < C) {
    ws[((int64_t)blockIdx.y * 2 + 0) * C + c] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    ws[((int64_t)blockIdx.y * 2 + 1) * C + c] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
  }
}

// ---------------------------------------------------------------- highway
__global__ void highway_fwd_k(const float* __restrict__ z, const float* __restrict__ x, float* __restrict__ y,
                              int rows, int H) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)rows * H;
       e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e / H), c = (int)(e - (int64_t)r * H);
    float hp = z[(int64_t)r * 2 * H + c], tp = z[(int64_t)r * 2 * H + H + c];
    float t = 1.f / (1.f + expf(-tp));
    y[e] = fmaxf(hp, 0.f) * t + x[e] * (1.f - t);
  }
}
__global__ void highway_bwd_k(const float* __restrict__ dy, const float* __restrict__ z, const float* __restrict__ x,
                              float* __restrict__ dz, float* __restrict__ dx, int rows, int H) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)rows * H;
       e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e / H), c = (int)(e - (int64_t)r * H);
    float hp = z[(int64_t)r * 2 * H + c], tp = z[(int64_t)r * 2 * H + H + c];
    float t = 1.f / (1.f + expf(-tp));
    float h = fmaxf(hp, 0.f), g = dy[e];
    dz[(int64_t)r * 2 * H + c] = hp > 0.f ? g * t : 0.f;
    dz[(int64_t)r * 2 * H + H + c] = g * (h - x[e]) * t * (1.f - t);
    dx[e] = g * (1.f - t);
  }
}

// ---------------------------------------------------------------- column sums
__global__ __launch_bounds__(256) void colsum_k(const float* __restrict__ x, int64_t ldx, float* __restrict__ out,
                                                int rows, int cols, int rows_per_block) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s = 0.f;
  if (c < cols) for (int r = r0 + rl; r < r1; r += 4) s += x[(int64_t)r * ldx + c];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < cols) atomicAdd(&out[c], red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

__global__ void axpby_k(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy, int rows,
                        int cols, float a, float b) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)rows * cols;
       e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e / cols), c = (int)(e - (int64_t)r * cols);
    float* d = y + (int64_t)r * ldy + c;
    float v = a * x[(int64_t)r * ldx + c];
    *d = (b == 0.f) ? v : v + b * *d;
  }
}
__global__ void seq_mask_k(const float* __restrict__ x, const int64_t* __restrict__ len, float* __restrict__ y,
                           int B, int T, int C, int round_bf16) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)B * T * C;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t bt = e / C;
    int b = (int)(bt / T), t = (int)(bt - (int64_t)b * T);
    const float v = (t < len[b]) ? x[e] : 0.f;
    y[e] = round_bf16 ? bf2f(f2bf(v)) : v;
  }
}
__global__ void bcast_add_k(const float* __restrict__ sv, float* __restrict__ y, int B, int T, int C) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)B * T * C;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C); const int b = (int)(e / ((int64_t)T * C));
    y[e] += sv[(int64_t)b * C + c];
  }
}
__global__ __launch_bounds__(256) void segment_colsum_k(const float* __restrict__ x, float* __restrict__ ds, int T,
                                                        int C, int accumulate) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, rl = threadIdx.x >> 6, b = blockIdx.y;
  float s = 0.f;
  if (c < C) for (int t = rl; t < T; t += 4) s += x[((int64_t)b * T + t) * C + c];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    const float v = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    float* d = ds + (int64_t)b * C + c;
    *d = accumulate ? *d + v : v;
  }
}
__global__ void to_bf16_k(const float* __restrict__ src, int64_t ld, uint16_t* __restrict__ dst, int rows, int cols,
                          int transpose) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)rows * cols;
       e += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(e / cols), c = (int)(e - (int64_t)r * cols);
    uint16_t v = f2bf(src[(int64_t)r * ld + c]);
    if (transpose) dst[(int64_t)c * rows + r] = v; else dst[e] = v;
  }
}

// ---------------------------------------------------------------- softmax over score rows (one wave per row)
constexpr int SM_MAXPER = 16;  // T <= 1024
__global__ __launch_bounds__(256) void softmax_fwd_k(const float* __restrict__ s, float* __restrict__ p,
                                                     float* __restrict__ pd, int64_t nrows, int T, float scale,
                                                     int causal, uint32_t thresh, float dscale, uint32_t stream,
                                                     const uint32_t* __restrict__ seedp) {
  const int lane = threadIdx.x & 63;
  const int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int i = (int)(row % T);
  const int lim = causal ? i + 1 : T;
  const float* sr = s + row * T;
  float v[SM_MAXPER];
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < SM_MAXPER; ++k) {
    int j = lane + k * 64;
    v[k] = (j < lim) ? sr[j] * scale : -INFINITY;
    m = fmaxf(m, v[k]);
  }
  m = wave_max(m);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < SM_MAXPER; ++k) { v[k] = (lane + k * 64 < lim) ? expf(v[k] - m) : 0.f; sum += v[k]; }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  const uint32_t seed = (thresh && seedp) ? *seedp : 0u;
#pragma unroll
  for (int k = 0; k < SM_MAXPER; ++k) {
    int j = lane + k * 64;
    if (j < T) {
      float pr = v[k] * inv;
      p[row * T + j] = pr;
      if (pd) {
        float q = pr;
        if (thresh) q = satt_keep(seed, stream, (uint32_t)(row * T + j), thresh) ? pr * dscale : 0.f;
        pd[row * T + j] = q;
      }
    }
  }
}
// y = keep(seed, stream, r*cols + c) ? x * scale : 0 : tf.layers.dropout forward and (applied to dy) backward
__global__ void dropout_k(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy, int64_t n, int cols,
                          uint32_t thresh, float scale, uint32_t stream, const uint32_t* __restrict__ seedp) {
  const uint32_t seed = seedp ? *seedp : 0u;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / cols; const int c = (int)(e - r * cols);
    const float v = x[r * ldx + c];
    y[r * ldy + c] = (thresh == 0 || satt_keep(seed, stream, (uint32_t)e, thresh)) ? v * scale : 0.f;
  }
}
// one wave per row of an arbitrary strided [rows, cols] matrix (the KV-cached incremental attention row)
__global__ __launch_bounds__(256) void softmax_rows_k(const float* __restrict__ s, int64_t lds_, float* __restrict__ p,
                                                      int64_t ldp, int rows, int cols, float scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* sr = s + (int64_t)row * lds_;
  float m = -INFINITY;
  for (int j = lane; j < cols; j += 64) m = fmaxf(m, sr[j] * scale);
  m = wave_max(m);
  float sum = 0.f;
  for (int j = lane; j < cols; j += 64) sum += expf(sr[j] * scale - m);
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int j = lane; j < cols; j += 64) p[(int64_t)row * ldp + j] = expf(sr[j] * scale - m) * inv;
}
__global__ __launch_bounds__(256) void softmax_bwd_k(const float* __restrict__ dpd, const float* __restrict__ p,
                                                     float* __restrict__ ds, int64_t nrows, int T, float scale,
                                                     uint32_t thresh, float dscale, uint32_t stream,
                                                     const uint32_t* __restrict__ seedp) {
  const int lane = threadIdx.x & 63;
  const int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const uint32_t seed = (thresh && seedp) ? *seedp : 0u;
  float dp[SM_MAXPER], pr[SM_MAXPER];
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < SM_MAXPER; ++k) {
    int j = lane + k * 64;
    dp[k] = 0.f; pr[k] = 0.f;
    if (j < T) {
      pr[k] = p[row * T + j];
      float g = dpd[row * T + j];
      if (thresh) g = satt_keep(seed, stream, (uint32_t)(row * T + j), thresh) ? g * dscale : 0.f;
      dp[k] = g;
      dot += g * pr[k];
    }
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int k = 0; k < SM_MAXPER; ++k) {
    int j = lane + k * 64;
    if (j < T) ds[row * T + j] = pr[k] * (dp[k] - dot) * scale;
  }
}

// ---------------------------------------------------------------- losses
// mel element (b, tm, c) lives at mel[(b*Td + tm/r)*mel_ld + (tm%r)*nm + c]  (r = Tm/Td), stop (b,td) at stop[(b*Td+td)*stop_ld]
__device__ __forceinline__ int64_t mel_addr(int64_t e, int nm, int rn, int64_t mel_ld) {
  const int64_t step = e / rn;              // b*Td + td
  return step * mel_ld + (e - step * rn);
}
__global__ void loss_sums_k(const float* __restrict__ mel, int64_t mel_ld, const float* __restrict__ tgt,
                            const float* __restrict__ smask, const float* __restrict__ stop, int64_t stop_ld,
                            const float* __restrict__ done, const float* __restrict__ bmask, int64_t nmel, int nm,
                            int rn, int64_t nstop, int l2, float* __restrict__ ws) {
  float s_abs = 0.f, s_m = 0.f, s_b = 0.f, s_bm = 0.f;
  const int64_t gt = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = gt; e < nmel; e += gs) {
    float w = smask[e / nm], d = mel[mel_addr(e, nm, rn, mel_ld)] - tgt[e];
    s_abs += (l2 ? d * d : fabsf(d)) * w;
  }
  for (int64_t e = gt; e < nmel / nm; e += gs) s_m += smask[e];
  for (int64_t e = gt; e < nstop; e += gs) {
    float x = stop[e * stop_ld], z = done[e], w = bmask[e];
    s_b += (fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)))) * w;
    s_bm += w;
  }
  s_abs = wave_sum(s_abs); s_m = wave_sum(s_m); s_b = wave_sum(s_b); s_bm = wave_sum(s_bm);
  __shared__ float red[4][EW_NT / 64];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][w] = s_abs; red[1][w] = s_m; red[2][w] = s_b; red[3][w] = s_bm; }
  __syncthreads();
  if (threadIdx.x < 4) {       // one atomic per block per accumulator
    float t = 0.f;
    for (int i = 0; i < EW_NT / 64; ++i) t += red[threadIdx.x][i];
    atomicAdd(&ws[threadIdx.x], t);
  }
}
__global__ void loss_grad_k(const float* __restrict__ mel, int64_t mel_ld, const float* __restrict__ tgt,
                            const float* __restrict__ smask, const float* __restrict__ stop, int64_t stop_ld,
                            const float* __restrict__ done, const float* __restrict__ bmask, int64_t nmel, int nm,
                            int rn, int64_t nstop, int l2, const float* __restrict__ ws, float* __restrict__ losses,
                            float* __restrict__ dmel, int64_t dmel_ld, float* __restrict__ dstop, int64_t dstop_ld) {
  const float inv_m = 1.f / ((float)nm * ws[1]), inv_b = 1.f / ws[3];
  const int64_t gt = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, gs = (int64_t)gridDim.x * blockDim.x;
  if (gt == 0) {
    float ml = ws[0] * inv_m, dl = ws[2] * inv_b;
    losses[0] = ml; losses[1] = dl; losses[2] = ml + dl;
  }
  if (dmel)
    for (int64_t e = gt; e < nmel; e += gs) {
      float w = smask[e / nm], d = mel[mel_addr(e, nm, rn, mel_ld)] - tgt[e];
      float g = l2 ? 2.f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
      dmel[mel_addr(e, nm, rn, dmel_ld)] = g * w * inv_m;
    }
  if (dstop)
    for (int64_t e = gt; e < nstop; e += gs) {
      float x = stop[e * stop_ld];
      dstop[e * dstop_ld] = (1.f / (1.f + expf(-x)) - done[e]) * bmask[e] * inv_b;
    }
}

// Split form for the training step: the mask sums depend on the batch only, so they are computed EARLY (satt_loss_mask_sums,
// off the critical path) and the loss needs ONE launch between the forward and the backward pass instead of a memset and two
// kernels: gradients from the known denominators, loss numerators by block reduction + one atomic per block, and the last
// block to arrive (a counter behind the completed atomics) writes the three loss values.
// ws: [0] sum |d| w, [1] sum w, [2] sum bce w_b, [3] sum w_b, [4] blocks done (as an int)
__global__ __launch_bounds__(256) void loss_mask_sums_k(const float* __restrict__ smask, const float* __restrict__ bmask,
                                                       int64_t nsm, int64_t nbm, float* __restrict__ ws) {
  float s_m = 0.f, s_b = 0.f;
  for (int64_t e = threadIdx.x; e < nsm; e += 256) s_m += smask[e];
  for (int64_t e = threadIdx.x; e < nbm; e += 256) s_b += bmask[e];
  s_m = wave_sum(s_m); s_b = wave_sum(s_b);
  __shared__ float red[2][4];
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s_m; red[1][threadIdx.x >> 6] = s_b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ws[0] = 0.f; ws[2] = 0.f; reinterpret_cast<int*>(ws)[4] = 0;
    ws[1] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    ws[3] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}
constexpr int LOSS_NT = 1024;
__global__ __launch_bounds__(LOSS_NT) void loss_fused_k(const float* __restrict__ mel, int64_t mel_ld, const float* __restrict__ tgt,
                                                       const float* __restrict__ smask, const float* __restrict__ stop,
                                                       int64_t stop_ld, const float* __restrict__ done,
                                                       const float* __restrict__ bmask, int64_t nmel, int nm, int rn,
                                                       int64_t nstop, int l2, float* __restrict__ ws, float* __restrict__ losses,
                                                       float* __restrict__ dmel, int64_t dmel_ld, float* __restrict__ dstop,
                                                       int64_t dstop_ld, int zero_pad) {
  const float inv_m = 1.f / ((float)nm * ws[1]), inv_b = 1.f / ws[3];
  // one WAVE per decoder step (a row of rn = r * num_mels values): the column -> frame division is made once per lane and
  // 64-column block, not per element (nmel < 2^31 is checked by the host: 32-bit index arithmetic), and a wave's accesses are
  // contiguous runs of a row
  const unsigned gtu = blockIdx.x * blockDim.x + threadIdx.x, gsu = gridDim.x * blockDim.x;
  const unsigned lane = threadIdx.x & 63, gw = gtu >> 6, nw = gsu >> 6;
  const unsigned unm = (unsigned)nm, urn = (unsigned)rn, nrows = (unsigned)(nmel / rn), r = urn / unm;
  float s_abs = 0.f, s_b = 0.f;
  // every load of a pass (RU rows x CU column blocks) is in flight before the first use: the kernel is latency bound (a wave
  // sees 3 - 4 rows), not bandwidth bound
  constexpr int RU = 4, CU = 3;
  for (unsigned cb = 0; cb < urn; cb += 64 * CU) {
    unsigned cc[CU], fr[CU]; bool cok[CU];
#pragma unroll
    for (int j = 0; j < CU; ++j) {
      const unsigned col = cb + 64 * j + lane;
      cok[j] = col < urn; cc[j] = cok[j] ? col : 0u; fr[j] = cc[j] / unm;
    }
    for (unsigned row0 = gw; row0 < nrows; row0 += RU * nw) {
      float w[RU][CU], d[RU][CU];
#pragma unroll
      for (int i = 0; i < RU; ++i) {
        const unsigned row = min(row0 + i * nw, nrows - 1);
#pragma unroll
        for (int j = 0; j < CU; ++j) {
          w[i][j] = smask[row * r + fr[j]];
          d[i][j] = mel[(int64_t)row * mel_ld + cc[j]] - tgt[row * urn + cc[j]];
        }
      }
#pragma unroll
      for (int i = 0; i < RU; ++i) {
        const unsigned row = row0 + i * nw;
#pragma unroll
        for (int j = 0; j < CU; ++j)
          if (row < nrows && cok[j]) {
            s_abs += (l2 ? d[i][j] * d[i][j] : fabsf(d[i][j])) * w[i][j];
            const float g = l2 ? 2.f * d[i][j] : (d[i][j] > 0.f ? 1.f : (d[i][j] < 0.f ? -1.f : 0.f));
            dmel[(int64_t)row * dmel_ld + cc[j]] = g * w[i][j] * inv_m;
          }
      }
    }
  }
  // padded gradient rows (the input-gradient GEMM reads whole 8-column groups): zeros behind the stop-token column
  if (zero_pad > 0 && (int)lane < zero_pad)
    for (unsigned row = gw; row < nrows; row += nw) dmel[(int64_t)row * dmel_ld + urn + 1 + lane] = 0.f;
  for (unsigned e = gtu; e < (unsigned)nstop; e += gsu) {
    const float x = stop[(int64_t)e * stop_ld], z = done[e], w = bmask[e];
    s_b += (fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)))) * w;
    dstop[(int64_t)e * dstop_ld] = (1.f / (1.f + expf(-x)) - z) * w * inv_b;
  }
  s_abs = wave_sum(s_abs); s_b = wave_sum(s_b);
  __shared__ float red[2][LOSS_NT / 64];
  __shared__ int last;
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][w] = s_abs; red[1][w] = s_b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t0 = 0.f, t1 = 0.f;
    for (int i = 0; i < LOSS_NT / 64; ++i) { t0 += red[0][i]; t1 += red[1][i]; }
    // RETURNING atomics: both sums have been performed at the memory side before the counter is touched
    const float o0 = atomicAdd(&ws[0], t0), o1 = atomicAdd(&ws[2], t1);
    asm volatile("" :: "v"(o0), "v"(o1));
    const int c = atomicAdd(reinterpret_cast<int*>(ws) + 4, 1);
    last = (c == (int)gridDim.x - 1);
    if (last) {
      const float a0 = __hip_atomic_load(&ws[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float a2 = __hip_atomic_load(&ws[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float ml = a0 * inv_m, dl = a2 * inv_b;
      losses[0] = ml; losses[1] = dl; losses[2] = ml + dl;
    }
  }
}

// ---------------------------------------------------------------- optimiser
// Sum of squares in a FIXED summation order (no atomics): block b writes its partial to state[4 + b], sumsq_final_k adds
// the partials in index order.  Data-parallel replicas compute the clip factor from bit-identical all-reduced gradients:
// with an atomic accumulation the factor - and from then on the replicas - would differ in the last bit from run to run.
constexpr int SUMSQ_PARTS = 2048;
__global__ __launch_bounds__(256) void sumsq_k(const float* __restrict__ g, int64_t n, float* __restrict__ state) {
  float s = 0.f;
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;    // the flat gradient buffer is 16 B aligned
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n4; e += stride) {
    const float4 v = g4[e];
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  for (int64_t e = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += stride) s += g[e] * g[e];
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) state[4 + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_k(float* __restrict__ state, int nparts) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) s += state[4 + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) state[0] = red[0];
}
// state[0]=sumsq  -> state[1]=global norm (of scaled grads), state[2]=lr_t, state[3]=clip*grad scale
__global__ void adam_prepare_k(float* __restrict__ state, int32_t* __restrict__ step_dev,
                               uint32_t* __restrict__ seed_dev, float lr0, int decay, float step_factor, float b1,
                               float b2, float clip, float grad_scale, const uint32_t* __restrict__ err0,
                               const uint32_t* __restrict__ err1, const uint32_t* __restrict__ err2) {
  // sticky hand-off-timeout words of the step's cluster kernels: a step whose recurrent kernels gave up waiting produced
  // garbage gradients - the update is SKIPPED on the device (state[4] = 1) and the host raises at its next status check
  // ... or whose gradient is not finite: under data parallelism a rank with a set error word poisons its gradient before the
  // all-reduce (poison_on_error_k), so the sum - and with it this test - is the same on EVERY rank: replicas skip together
  const bool timed_out = (err0 && *err0) || (err1 && *err1) || (err2 && *err2), bad = timed_out || !isfinite(state[0]);
  state[4] = bad ? 1.f : 0.f;            // (state[4..] are the sum-of-squares partials: consumed by sumsq_final_k already)
  // STICKY counts behind the partials (state[4] is rewritten every step; the host looks only at log / checkpoint steps): skipped
  // updates since the host last cleared them, and those of them whose only reason was a non-finite gradient - an isolated NaN
  // step between two log steps must stop the run as the reference's NanTensorHook does, not become a silently dropped update
  if (bad) state[4 + SUMSQ_PARTS] += 1.f;
  if (bad && !timed_out) state[5 + SUMSQ_PARTS] += 1.f;
  const int step = step_dev[0];  // 0-based global_step before this update
  const float norm = sqrtf(state[0]) * grad_scale;
  float lr = lr0;
  if (decay) {
    const float warm = 4000.f, s = (float)step * step_factor + 1.f;
    lr = lr0 * sqrtf(warm) * fminf(s * powf(warm, -1.5f), rsqrtf(s));
  }
  const float t = (float)(step + 1);
  state[1] = norm;
  state[2] = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
  state[3] = grad_scale * (clip > 0.f ? 1.f / fmaxf(1.f, norm / clip) : 1.f);
  step_dev[0] = step + 1;
  if (seed_dev) seed_dev[0] += 1u;
}
__global__ void poison_on_error_k(float* __restrict__ g, const uint32_t* __restrict__ err0, const uint32_t* __restrict__ err1,
                                  const uint32_t* __restrict__ err2) {
  if ((err0 && *err0) || (err1 && *err1) || (err2 && *err2)) g[0] = __builtin_nanf("");
}
__global__ void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, int64_t n, const float* __restrict__ state, float b1, float b2,
                       float eps) {
  const float lr_t = state[2], gs = state[3];
  if (state[4] != 0.f) return;           // cluster hand-off timeout in this step (adam_prepare_k): parameters and moments stay
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    float gr = g[e] * gs;
    float mm = b1 * m[e] + (1.f - b1) * gr;
    float vv = b2 * v[e] + (1.f - b2) * gr * gr;
    m[e] = mm; v[e] = vv;
    p[e] -= lr_t * mm / (sqrtf(vv) + eps);
  }
}

}  // namespace

// ---------------------------------------------------------------- location-filter weight gradient (forward attention)
// dF[j][k] += sum_{b, t >= 1, t'} a[b, t-1, t' + j - PL] * dfl[b, t, t', k]   (a_{-1} = 0: no term for t = 0)
// dbF[k]   += sum_{b, t, t'} dfl[b, t, t', k]
// 50 + 5 outputs over 2 M rows: as a GEMM this is one 64 x 64 tile with K = B*Td*Ti (0.39 ms of a split-K launch);
// here every workgroup walks a few (b, t) pairs with the previous alignment row in LDS, a thread per memory row t'
// keeps the KW x 5 partial sums in registers, and one reduction + 55 atomics per workgroup finish it.
template <int KW>
__global__ __launch_bounds__(256) void loc_filter_dw_k(const float* __restrict__ a1, const float* __restrict__ dfl,
                                                       float* __restrict__ dF, float* __restrict__ dbF, int nbt, int Td,
                                                       int Ti, int per) {
  constexpr int F = 5, PL = (KW - 1) / 2;
  __shared__ float arow[1024 + KW];
  __shared__ float red[KW * F + F];
  const int tid = threadIdx.x;
  float acc[KW * F], accb[F];
#pragma unroll
  for (int i = 0; i < KW * F; ++i) acc[i] = 0.f;
#pragma unroll
  for (int k = 0; k < F; ++k) accb[k] = 0.f;
  if (tid < KW * F + F) red[tid] = 0.f;
  const int bt0 = blockIdx.x * per, bt1 = min(nbt, bt0 + per);
  for (int bt = bt0; bt < bt1; ++bt) {
    const bool has_prev = bt % Td != 0;
    __syncthreads();
    for (int i = tid; i < Ti + KW; i += 256) {
      const int tt = i - PL;
      arow[i] = (has_prev && tt >= 0 && tt < Ti) ? a1[(int64_t)(bt - 1) * Ti + tt] : 0.f;
    }
    __syncthreads();
    for (int tt = tid; tt < Ti; tt += 256) {
      const float* d = dfl + ((int64_t)bt * Ti + tt) * F;
      float dv[F];
#pragma unroll
      for (int k = 0; k < F; ++k) { dv[k] = d[k]; accb[k] += dv[k]; }
#pragma unroll
      for (int j = 0; j < KW; ++j) {
        const float av = arow[tt + j];            // = a[t' + j - PL]
#pragma unroll
        for (int k = 0; k < F; ++k) acc[j * F + k] += av * dv[k];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < KW * F; ++i) { const float v = wave_sum(acc[i]); if ((tid & 63) == 0) atomicAdd(red + i, v); }
#pragma unroll
  for (int k = 0; k < F; ++k) { const float v = wave_sum(accb[k]); if ((tid & 63) == 0) atomicAdd(red + KW * F + k, v); }
  __syncthreads();
  if (tid < KW * F) atomicAdd(dF + tid, red[tid]);
  else if (tid < KW * F + F) atomicAdd(dbF + (tid - KW * F), red[tid]);
}

#define S_ ((hipStream_t)stream)

extern "C" int satt_embedding_fwd(const int64_t* ids, const float* table, float* out, int n, int dim, int offset,
                                  void* stream) {
  if (n <= 0) return SATT_OK;
  hipLaunchKernelGGL(embedding_fwd_k, dim3(ew_blocks((int64_t)n * dim)), dim3(EW_NT), 0, S_, ids, table, out, n, dim,
                     offset);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_embedding_bwd(const int64_t* ids, const float* dout, float* dtable, int n, int dim, int offset,
                                  void* stream) {
  if (n <= 0) return SATT_OK;
  hipLaunchKernelGGL(embedding_bwd_k, dim3(ew_blocks((int64_t)((n + EMB_RUN - 1) / EMB_RUN) * dim)), dim3(EW_NT), 0, S_, ids, dout, dtable, n,
                     dim, offset);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_embedding_bwd_rows(const int64_t* ids, const float* dout, float* dtable, int n, int dim, int offset,
                                       int nrows, void* stream) {
  if (n <= 0) return SATT_OK;
  if (nrows <= 0 || dim <= 0) return SATT_E_BADARG;
  if (n > EB_NT * EB_MAXI || nrows > 4096) return satt_embedding_bwd(ids, dout, dtable, n, dim, offset, stream);
  const size_t smem = sizeof(int) * (size_t)n;
  hipLaunchKernelGGL(embedding_bwd_rows_k, dim3(nrows), dim3(EB_NT), smem, S_, ids, dout, dtable, n, dim, offset);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_act_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, float* dx, int64_t lddx,
                            int rows, int cols, int act, float scale, void* stream) {
  if (rows <= 0 || cols <= 0) return SATT_OK;
  hipLaunchKernelGGL(act_bwd_k, dim3(ew_blocks((int64_t)rows * cols)), dim3(EW_NT), 0, S_, dy, lddy, y, ldy, dx, lddx,
                     rows, cols, act, scale);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_act_bwd_res(const float* dy, int64_t lddy, const float* z, int64_t ldz, const float* res, int64_t ldres,
                                float* dx, int64_t lddx, int rows, int cols, int act, float scale, void* stream) {
  if (rows <= 0 || cols <= 0) return SATT_OK;
  if (!res) return SATT_E_BADARG;
  hipLaunchKernelGGL(act_bwd_res_k, dim3(ew_blocks((int64_t)rows * cols)), dim3(EW_NT), 0, S_, dy, lddy, z, ldz, res, ldres, dx,
                     lddx, rows, cols, act, scale);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int64_t satt_bn_ws_floats(int rows, int C) { return (int64_t)2 * C * (bn_chunks(rows) + 1); }
extern "C" int satt_bn_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy,
                           float* mean, float* rstd, float* moving_mean, float* moving_var, float* ws, int rows,
                           int C, float eps, float momentum, int act, void* stream) {
  if (rows <= 0 || C <= 0) return SATT_E_BADARG;
  const int nchunk = bn_chunks(rows);
  hipLaunchKernelGGL(bn_partial_k, dim3((C + 63) / 64, nchunk), dim3(256), 0, S_, x, ldx, ws, rows, C);
  hipLaunchKernelGGL(bn_finalize_k, dim3((C + 63) / 64), dim3(256), 0, S_, ws, nchunk, rows, C, eps, momentum, mean,
                     rstd, moving_mean, moving_var);
  hipLaunchKernelGGL(bn_apply_k, dim3(ew_blocks((int64_t)rows * C)), dim3(EW_NT), 0, S_, x, ldx, gamma, beta, mean,
                     rstd, y, ldy, rows, C, act);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int64_t satt_bn_fused_ws_floats(int rows, int C) { return (int64_t)2 * C * (fbn_chunks(rows) + 1); }
extern "C" int satt_bn_fused_sync_words(int C) { return 2 * ((C + 63) / 64); }
static bool fbn_fits(int rows, int C) {
  return (int64_t)((C + 63) / 64) * fbn_chunks(rows) <= FBN_MAX_WGS && fbn_chunks(rows) <= 4 * FBN_KMAX;
}
extern "C" int satt_bn_fwd_fused(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy,
                                 float* mean, float* rstd, float* moving_mean, float* moving_var, float* ws, uint32_t* sync,
                                 int rows, int C, float eps, float momentum, int act, void* stream) {
  if (rows <= 0 || C <= 0 || !sync || !ws) return SATT_E_BADARG;
  if (!fbn_fits(rows, C)) return SATT_E_UNSUPPORTED;
  const int nchunk = fbn_chunks(rows);
  hipLaunchKernelGGL(bn_fwd_fused_k, dim3((C + 63) / 64, nchunk), dim3(256), 0, S_, x, ldx, gamma, beta, y, ldy, mean, rstd,
                     moving_mean, moving_var, ws, sync, rows, C, eps, momentum, act, nchunk);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_bn_bwd_fused(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                                 const float* beta, const float* mean, const float* rstd, float* dx, int64_t lddx,
                                 float* dgamma, float* dbeta, float* ws, uint32_t* sync, int rows, int C, int act, void* stream) {
  if (rows <= 0 || C <= 0 || !sync || !ws) return SATT_E_BADARG;
  if (!fbn_fits(rows, C)) return SATT_E_UNSUPPORTED;
  const int nchunk = fbn_chunks(rows);
  hipLaunchKernelGGL(bn_bwd_fused_k, dim3((C + 63) / 64, nchunk), dim3(256), 0, S_, dy, lddy, x, ldx, gamma, beta, mean, rstd, dx,
                     lddx, dgamma, dbeta, ws, sync, rows, C, act, nchunk);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_bn_infer(const float* x, int64_t ldx, const float* gamma, const float* beta,
                             const float* moving_mean, const float* moving_var, float* y, int64_t ldy, int rows,
                             int C, float eps, int act, void* stream) {
  if (rows <= 0 || C <= 0) return SATT_E_BADARG;
  hipLaunchKernelGGL(bn_infer_k, dim3(ew_blocks((int64_t)rows * C)), dim3(EW_NT), 0, S_, x, ldx, gamma, beta,
                     moving_mean, moving_var, y, ldy, rows, C, eps, act);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_bn_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                           const float* beta, const float* mean, const float* rstd, float* dx, int64_t lddx,
                           float* dgamma, float* dbeta, float* ws, int rows, int C, int act, void* stream) {
  if (rows <= 0 || C <= 0) return SATT_E_BADARG;
  const int nchunk = bn_chunks(rows);
  hipLaunchKernelGGL(bn_bwd_partial_k, dim3((C + 63) / 64, nchunk), dim3(256), 0, S_, dy, lddy, x, ldx, gamma, beta,
                     mean, rstd, ws, rows, C, act);
  hipLaunchKernelGGL(bn_bwd_finalize_k, dim3((C + 63) / 64), dim3(256), 0, S_, ws, nchunk, C, dgamma, dbeta);
  hipLaunchKernelGGL(bn_bwd_apply_k, dim3(ew_blocks((int64_t)rows * C)), dim3(EW_NT), 0, S_, dy, lddy, x, ldx, gamma,
                     beta, mean, rstd, ws + (int64_t)nchunk * 2 * C, dx, lddx, rows, C, act);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_bn_maxpool_fwd(const float* x, const float* gamma, const float* beta, float* mp, float* mean, float* rstd,
                                   float* moving_mean, float* moving_var, float* ws, int B, int T, int C, float eps,
                                   float momentum, int act, void* stream) {
  const int rows = B * T;
  if (rows <= 0 || C <= 0 || T <= 0) return SATT_E_BADARG;
  if (C % 4 || (int64_t)rows * (C / 4) >= (1ll << 31) ||
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(mp) | reinterpret_cast<uintptr_t>(gamma) |
        reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(rstd)) & 15))
    return SATT_E_UNSUPPORTED;
  const int nchunk = bn_chunks(rows);
  hipLaunchKernelGGL(bn_partial_k, dim3((C + 63) / 64, nchunk), dim3(256), 0, S_, x, (int64_t)C, ws, rows, C);
  hipLaunchKernelGGL(bn_finalize_k, dim3((C + 63) / 64), dim3(256), 0, S_, ws, nchunk, rows, C, eps, momentum, mean, rstd,
                     moving_mean, moving_var);
  hipLaunchKernelGGL(bn_apply_maxpool4_k, dim3(ew_blocks((int64_t)((rows + BNP_ROWS - 1) / BNP_ROWS) * (C / 4))), dim3(EW_NT), 0, S_, x, gamma, beta, mean, rstd,
                     mp, rows, T, C / 4, act);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_maxpool_bn_bwd(const float* dmp, const float* x, const float* gamma, const float* beta, const float* mean,
                                   const float* rstd, float* dx, float* dgamma, float* dbeta, float* ws, float* dbuf, int B,
                                   int T, int C, int act, void* stream) {
  const int rows = B * T;
  if (rows <= 0 || C <= 0 || T <= 0 || !dbuf) return SATT_E_BADARG;
  const int nchunk = bn_chunks(rows);
  hipLaunchKernelGGL(maxpool_bn_bwd_partial_k, dim3((C + 63) / 64, nchunk), dim3(256), 0, S_, dmp, x, gamma, beta, mean, rstd, dbuf,
                     ws, rows, T, C, act);
  hipLaunchKernelGGL(bn_bwd_finalize_k, dim3((C + 63) / 64), dim3(256), 0, S_, ws, nchunk, C, dgamma, dbeta);
  // dbuf already carries the activation's derivative: the apply pass runs with the identity activation
  hipLaunchKernelGGL(bn_bwd_apply_k, dim3(ew_blocks((int64_t)rows * C)), dim3(EW_NT), 0, S_, dbuf, (int64_t)C, x, (int64_t)C, gamma,
                     beta, mean, rstd, ws + (int64_t)nchunk * 2 * C, dx, (int64_t)C, rows, C, (int)SATT_ACT_NONE);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_maxpool_fwd(const float* x, float* y, int B, int T, int C, void* stream) {
  hipLaunchKernelGGL(maxpool_fwd_k, dim3(ew_blocks((int64_t)B * T * C)), dim3(EW_NT), 0, S_, x, y, B, T, C);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_maxpool_bwd(const float* dy, const float* x, float* dx, int B, int T, int C, void* stream) {
  const bool vec = C % 4 == 0 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0 &&
                   (int64_t)B * T * (C / 4) < (1ll << 31);
  if (vec) hipLaunchKernelGGL(maxpool_bwd4_k, dim3(ew_blocks((int64_t)B * T * (C / 4))), dim3(EW_NT), 0, S_, dy, x, dx, B * T, T, C / 4);
  else hipLaunchKernelGGL(maxpool_bwd_k, dim3(ew_blocks((int64_t)B * T * C)), dim3(EW_NT), 0, S_, dy, x, dx, B, T, C);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_highway_fwd(const float* z, const float* x, float* y, int rows, int H, void* stream) {
  hipLaunchKernelGGL(highway_fwd_k, dim3(ew_blocks((int64_t)rows * H)), dim3(EW_NT), 0, S_, z, x, y, rows, H);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_highway_bwd(const float* dy, const float* z, const float* x, float* dz, float* dx, int rows,
                                int H, void* stream) {
  hipLaunchKernelGGL(highway_bwd_k, dim3(ew_blocks((int64_t)rows * H)), dim3(EW_NT), 0, S_, dy, z, x, dz, dx, rows, H);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_colsum(const float* x, int64_t ldx, float* out, int rows, int cols, int accumulate,
                           void* stream) {
  if (rows <= 0 || cols <= 0) return SATT_OK;
  if (!accumulate) { if (hipMemsetAsync(out, 0, sizeof(float) * cols, S_) != hipSuccess) return SATT_E_LAUNCH; }
  const int rpb = 256;
  hipLaunchKernelGGL(colsum_k, dim3((cols + 63) / 64, (rows + rpb - 1) / rpb), dim3(256), 0, S_, x, ldx, out, rows,
                     cols, rpb);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
/* dF [kernel,1,filters], dbF [filters] += the location-filter gradients from the saved softmax alignments a1 [B,Td,Ti] and
 * d fl [B,Td,Ti,filters]; SATT_E_UNSUPPORTED unless kernel == 10, filters == 5, Ti <= 1024 (callers then use satt_gemm) */
extern "C" int satt_loc_filter_dw(const float* a1, const float* dfl, float* dF, float* dbF, int B, int Td, int Ti, int kernel,
                                  int filters, void* stream) {
  if (!a1 || !dfl || !dF || !dbF || B <= 0 || Td <= 0 || Ti <= 0) return SATT_E_BADARG;
  if (kernel != 10 || filters != 5 || Ti > 1024) return SATT_E_UNSUPPORTED;
  const int nbt = B * Td, per = (nbt + 1023) / 1024;
  hipLaunchKernelGGL(loc_filter_dw_k<10>, dim3((nbt + per - 1) / per), dim3(256), 0, S_, a1, dfl, dF, dbF, nbt, Td, Ti, per);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
// L2 regularisation of the baseline model (reference modules/regularizers.py:11-18, models/models.py:109-114):
// loss += scale * sum over the selected tensors of sum(w^2) / 2 (tf.nn.l2_loss), i.e. d w += scale * w.
// table [nseg][2] = (offset, count) into the flat parameter / gradient buffers; grid (blocks, nseg).
namespace {
__global__ void l2_reg_k(const float* __restrict__ w, float* __restrict__ g, const int64_t* __restrict__ tab, float scale,
                         float* __restrict__ reg, float* __restrict__ total) {
  const int64_t off = tab[2 * blockIdx.y], n = tab[2 * blockIdx.y + 1];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = w[off + i];
    g[off + i] += scale * v;
    s += v * v;
  }
  s = wave_sum(s);
  __shared__ float sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float r = 0.5f * scale * ((sm[0] + sm[1]) + (sm[2] + sm[3]));
    atomicAdd(reg, r);
    if (total) atomicAdd(total, r);
  }
}
}  // namespace
extern "C" int satt_l2_reg(const float* w, float* g, const int64_t* table, int nseg, float scale, float* reg, float* total,
                           void* stream) {
  if (!w || !g || !table || !reg || nseg <= 0) return SATT_E_BADARG;
  hipLaunchKernelGGL(l2_reg_k, dim3(32, nseg), dim3(256), 0, (hipStream_t)stream, w, g, table, scale, reg, total);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

extern "C" int satt_axpby(const float* x, int64_t ldx, float* y, int64_t ldy, int rows, int cols, float a, float b,
                          void* stream) {
  if (rows <= 0 || cols <= 0) return SATT_OK;
  hipLaunchKernelGGL(axpby_k, dim3(ew_blocks((int64_t)rows * cols)), dim3(EW_NT), 0, S_, x, ldx, y, ldy, rows, cols, a,
                     b);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_seq_mask(const float* x, const int64_t* lengths, float* y, int B, int T, int C, int round_bf16,
                             void* stream) {
  hipLaunchKernelGGL(seq_mask_k, dim3(ew_blocks((int64_t)B * T * C)), dim3(EW_NT), 0, S_, x, lengths, y, B, T, C, round_bf16);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_bcast_add(const float* sv, float* y, int B, int T, int C, void* stream) {
  hipLaunchKernelGGL(bcast_add_k, dim3(ew_blocks((int64_t)B * T * C)), dim3(EW_NT), 0, S_, sv, y, B, T, C);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_segment_colsum(const float* x, float* ds, int B, int T, int C, int accumulate, void* stream) {
  hipLaunchKernelGGL(segment_colsum_k, dim3((C + 63) / 64, B), dim3(256), 0, S_, x, ds, T, C, accumulate);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_to_bf16(const float* src, int64_t ld, uint16_t* dst, int rows, int cols, int transpose,
                            void* stream) {
  if (rows <= 0 || cols <= 0) return SATT_OK;
  hipLaunchKernelGGL(to_bf16_k, dim3(ew_blocks((int64_t)rows * cols)), dim3(EW_NT), 0, S_, src, ld, dst, rows, cols,
                     transpose);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_softmax_fwd(const float* s, float* p, float* pd, int nbh, int T, float scale, int causal,
                                uint32_t drop_thresh, float drop_scale, uint32_t drop_stream, const uint32_t* seed,
                                void* stream) {
  if (T > SM_MAXPER * 64) return SATT_E_UNSUPPORTED;
  const int64_t nrows = (int64_t)nbh * T;
  hipLaunchKernelGGL(softmax_fwd_k, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, S_, s, p, pd, nrows, T, scale,
                     causal, drop_thresh, drop_scale, drop_stream, seed);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_dropout(const float* x, int64_t ldx, float* y, int64_t ldy, int rows, int cols, uint32_t drop_thresh,
                            float drop_scale, uint32_t drop_stream, const uint32_t* seed, void* stream) {
  if (rows <= 0 || cols <= 0) return SATT_E_BADARG;
  const int64_t n = (int64_t)rows * cols;
  hipLaunchKernelGGL(dropout_k, dim3(ew_blocks(n)), dim3(EW_NT), 0, S_, x, ldx, y, ldy, n, cols, drop_thresh,
                     drop_thresh ? drop_scale : 1.f, drop_stream, seed);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_softmax_rows(const float* s, int64_t lds_, float* p, int64_t ldp, int rows, int cols, float scale,
                                 void* stream) {
  if (rows <= 0 || cols <= 0) return SATT_E_BADARG;
  hipLaunchKernelGGL(softmax_rows_k, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S_, s, lds_, p, ldp, rows, cols, scale);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_softmax_bwd(const float* dpd, const float* p, float* ds, int nbh, int T, float scale, int causal,
                                uint32_t drop_thresh, float drop_scale, uint32_t drop_stream, const uint32_t* seed,
                                void* stream) {
  (void)causal;  // masked positions have p == 0 -> ds == 0
  if (T > SM_MAXPER * 64) return SATT_E_UNSUPPORTED;
  const int64_t nrows = (int64_t)nbh * T;
  hipLaunchKernelGGL(softmax_bwd_k, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, S_, dpd, p, ds, nrows, T, scale,
                     drop_thresh, drop_scale, drop_stream, seed);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_loss_fwd_bwd(const float* mel, int64_t mel_ld, const float* target, const float* spec_mask,
                                 const float* stop, int64_t stop_ld, const float* done, const float* bin_mask, int B,
                                 int Tm, int nm, int Td, int l2, float* losses, float* dmel, int64_t dmel_ld,
                                 float* dstop, int64_t dstop_ld, float* ws, void* stream) {
  if (B <= 0 || Td <= 0 || Tm % Td != 0) return SATT_E_BADARG;
  const int64_t nmel = (int64_t)B * Tm * nm, nstop = (int64_t)B * Td;
  const int rn = (Tm / Td) * nm;
  if (hipMemsetAsync(ws, 0, 4 * sizeof(float), S_) != hipSuccess) return SATT_E_LAUNCH;
  hipLaunchKernelGGL(loss_sums_k, dim3(std::min(ew_blocks(nmel), 512)), dim3(EW_NT), 0, S_, mel, mel_ld, target, spec_mask, stop,
                     stop_ld, done, bin_mask, nmel, nm, rn, nstop, l2, ws);
  hipLaunchKernelGGL(loss_grad_k, dim3(ew_blocks(nmel)), dim3(EW_NT), 0, S_, mel, mel_ld, target, spec_mask, stop,
                     stop_ld, done, bin_mask, nmel, nm, rn, nstop, l2, ws, losses, dmel, dmel_ld, dstop, dstop_ld);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_loss_mask_sums(const float* spec_mask, const float* bin_mask, int B, int Tm, int Td, float* ws, void* stream) {
  if (B <= 0 || Td <= 0 || Tm <= 0 || !ws) return SATT_E_BADARG;
  hipLaunchKernelGGL(loss_mask_sums_k, dim3(1), dim3(256), 0, S_, spec_mask, bin_mask, (int64_t)B * Tm, (int64_t)B * Td, ws);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_loss_fwd_bwd_presummed(const float* mel, int64_t mel_ld, const float* target, const float* spec_mask,
                                           const float* stop, int64_t stop_ld, const float* done, const float* bin_mask, int B,
                                           int Tm, int nm, int Td, int l2, float* losses, float* dmel, int64_t dmel_ld,
                                           float* dstop, int64_t dstop_ld, float* ws, void* stream) {
  if (B <= 0 || Td <= 0 || Tm % Td != 0 || !dmel || !dstop) return SATT_E_BADARG;
  const int64_t nmel = (int64_t)B * Tm * nm, nstop = (int64_t)B * Td;
  const int rn = (Tm / Td) * nm;
  if (nmel >= (1ll << 31)) return SATT_E_UNSUPPORTED;
  // the gradient buffer may carry zero-filled pad columns behind [mel | stop] (dstop == dmel + rn, dmel_ld > rn + 1)
  const int zero_pad = (dstop == dmel + rn && dstop_ld == dmel_ld && dmel_ld > rn + 1) ? (int)std::min<int64_t>(dmel_ld - rn - 1, 64) : 0;
  const int nblk = (int)std::min<int64_t>((nstop + LOSS_NT / 64 - 1) / (LOSS_NT / 64), 256);
  hipLaunchKernelGGL(loss_fused_k, dim3(nblk), dim3(LOSS_NT), 0, S_, mel, mel_ld, target, spec_mask, stop,
                     stop_ld, done, bin_mask, nmel, nm, rn, nstop, l2, ws, losses, dmel, dmel_ld, dstop, dstop_ld, zero_pad);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
// Probe for the single-launch attention backward: does a KERNEL of another stream run while a kernel of `stream` is
// running?  The probe kernel spins (bounded) until *flag != 0 and reports 1 in *out if it saw it; a second one-thread
// kernel launched on `set_stream` right after it sets the flag.  If both streams share a hardware queue, or kernels
// are serialised (counter-collecting profilers do that), the setter is stuck behind the probe: out = 0.
__global__ void stream_probe_k(const uint32_t* __restrict__ flag, uint32_t* __restrict__ out, unsigned max_spins) {
  unsigned seen = 0;
  for (unsigned i = 0; i < max_spins; ++i) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) { seen = 1; break; }
    __builtin_amdgcn_s_sleep(50);
  }
  *out = seen;
}
__global__ void stream_probe_set_k(uint32_t* __restrict__ flag) {
  __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
extern "C" int satt_stream_probe(uint32_t* flag, uint32_t* out, unsigned max_spins, void* stream, void* set_stream) {
  if (!flag || !out || stream == set_stream) return SATT_E_BADARG;
  hipLaunchKernelGGL(stream_probe_k, dim3(1), dim3(1), 0, S_, flag, out, max_spins);
  hipLaunchKernelGGL(stream_probe_set_k, dim3(1), dim3(1), 0, static_cast<hipStream_t>(set_stream), flag);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_sumsq(const float* g, int64_t n, float* state, void* stream) {
  if (reinterpret_cast<uintptr_t>(g) & 15) return SATT_E_BADARG;
  const int nb = std::min(ew_blocks(n, 256 * 16), SUMSQ_PARTS);
  hipLaunchKernelGGL(sumsq_k, dim3(nb), dim3(256), 0, S_, g, n, state);
  hipLaunchKernelGGL(sumsq_final_k, dim3(1), dim3(256), 0, S_, state, nb);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_sumsq_state_floats(void) { return 4 + SUMSQ_PARTS + 2; }
extern "C" int satt_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float* state,
                              int32_t* step_dev, uint32_t* seed_dev, float lr0, int decay, float step_factor,
                              float b1, float b2, float eps, float clip, float grad_scale, const uint32_t* err0,
                              const uint32_t* err1, const uint32_t* err2, void* stream) {
  hipLaunchKernelGGL(adam_prepare_k, dim3(1), dim3(1), 0, S_, state, step_dev, seed_dev, lr0, decay, step_factor, b1,
                     b2, clip, grad_scale, err0, err1, err2);
  hipLaunchKernelGGL(adam_k, dim3(ew_blocks(n, 256 * 4)), dim3(256), 0, S_, p, g, m, v, n, state, b1, b2, eps);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
extern "C" int satt_poison_on_error(float* g, const uint32_t* err0, const uint32_t* err1, const uint32_t* err2, void* stream) {
  if (!g) return SATT_E_BADARG;
  hipLaunchKernelGGL(poison_on_error_k, dim3(1), dim3(1), 0, S_, g, err0, err1, err2);
  SATT_LAUNCH_CHECK(); return SATT_OK;
}
