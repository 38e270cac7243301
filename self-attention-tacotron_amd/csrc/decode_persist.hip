// Persistent cooperative decode loop (BASELINE config 5): ALL decoder steps of an utterance in ONE launch.
// The captured-graph form of the step (csrc/decode.hip) is bound by launch boundaries: 11 dependent launches x >= 4.7 us
// each, whatever they compute (profiles/r02_decode_timeline.txt).  Here the same phases run inside one kernel on G
// workgroups that all sit on ONE XCD (grid = 8 G workgroups, workgroup i runs on XCD i % 8: the members are i % 8 == 0, the
// others exit at once; the placement is verified through HW_REG_XCC_ID before anything depends on it), separated by grid
// barriers through that XCD's L2: every member publishes an 8-byte {sequence} granule with a plain store and polls the
// G granules with sc1 loads (bounded; a timeout sets the error word and the kernel runs out without waiting again).
// A barrier costs ~1 us instead of a launch boundary.  Data handed from phase to phase is written with plain stores
// (complete in L2 after s_waitcnt vmcnt(0)) and read after an acquire fence (buffer_inv: no stale L1 lines).
// Phases of a step (12 for the dual-source model): pre-net layers | attention LSTM (gates + cell) | query layer |
// energies per slice of memory rows | masked softmax + forward recursion + context slice | LSTM1 | LSTM2 | K|V|Q row |
// self-attention partials per (sample, head, chunk of cache rows) | combine + output transform | mel / stop projection.
#include "cluster_xchg.h"

namespace {

constexpr int PNT = 256;                 // threads per member workgroup
constexpr int PSMEM = 6144;              // floats of LDS scratch shared by the phases
constexpr int PL_COLS = 32, PL_KMAX = 1024, PL_KI = PL_KMAX / 32;

__device__ __forceinline__ float bsum4(float v, float* sm, int tid) {      // sum over the 4 waves of a member
  v = wave_sum(v);
  lds_barrier();
  if ((tid & 63) == 0) sm[tid >> 6] = v;
  lds_barrier();
  return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
__device__ __forceinline__ float bmax4(float v, float* sm, int tid) {
  v = wave_max(v);
  lds_barrier();
  if ((tid & 63) == 0) sm[tid >> 6] = v;
  lds_barrier();
  return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// Workgroup barriers inside the phases are lds_barrier() (s_waitcnt lgkmcnt(0); s_barrier): they order LDS traffic only.
// __syncthreads() would also drain every outstanding global load of the wave, i.e. turn "weights, inputs and epilogue
// operands in flight together" into one L2 round trip each (a phase is ~5 dependent round trips of ~0.9 us otherwise).
// grid barrier over the G members (all on one XCD): see the file header
__device__ __forceinline__ void grid_barrier(u64* bar, int G, int me, uint32_t seq, unsigned int* err, int* dead, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  if (tid == 0) gput(bar + me, seq, 0.f, true);
  if (tid < 64 && !*dead) {
    const gu64* g = (const gu64*)(bar + min(tid, G - 1));
    for (unsigned spins = 0;; ++spins) {
      const u64 x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all((uint32_t)(x >> 32) >= seq)) break;
      if (spins > (1u << 20)) {
        if (tid == 0) __hip_atomic_store((gu32*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *dead = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  lds_barrier();
  asm volatile("" ::: "memory");
}

// Data that another member wrote in an earlier phase is read with sc1 loads, which bypass this CU's L1 and are served from
// the XCD's L2 (the members share it).  The alternative - an agent-scope acquire fence (buffer_inv sc1) after every barrier -
// also drops the L2's lines: every phase then fetched its weights from HBM again (15 us per phase); buffer_inv sc0 does not
// drop the L1 in this mode (stale activations).  Weights, biases and the memories are read-only: ordinary cached loads.
__device__ __forceinline__ float ldc(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// BRANCH-FREE: every load is issued unconditionally at a clamped (always valid) address and masked afterwards.  A load
// inside `if (k < K)` makes the compiler wait for it at the end of the branch: 5-25 loads of one L2 (or LDS) latency EACH,
// 2 us per phase, instead of all of them in flight together.
template <bool BF, bool VEC>
__device__ __forceinline__ void lin_weights(const satt_dec_linear_params& p, float (&w)[PL_KI][4], int kl, int n, int K) {
  const int nc = VEC ? min(n, (int)p.ldw - 4) : n;          // rows are ldw >= 4 columns wide in memory
#pragma unroll
  for (int i = 0; i < PL_KI; ++i) {
    const int k = kl + 32 * i, kc = min(k, K - 1);
    if (VEC) {
      float v0, v1, v2, v3;
      if (BF) {
        const uint2 v = *reinterpret_cast<const uint2*>(p.Wb + (int64_t)kc * p.ldw + nc);
        v0 = __uint_as_float(v.x << 16); v1 = __uint_as_float(v.x & 0xFFFF0000u);
        v2 = __uint_as_float(v.y << 16); v3 = __uint_as_float(v.y & 0xFFFF0000u);
      } else {
        const float4 v = *reinterpret_cast<const float4*>(p.W + (int64_t)kc * p.ldw + nc);
        v0 = v.x; v1 = v.y; v2 = v.z; v3 = v.w;
      }
      const bool ok = k < K && n < p.N;
      w[i][0] = ok ? v0 : 0.f; w[i][1] = ok ? v1 : 0.f; w[i][2] = ok ? v2 : 0.f; w[i][3] = ok ? v3 : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nj = min(n + j, p.N - 1);
        const float v = BF ? bf2f(p.Wb[(int64_t)kc * p.ldw + nj]) : p.W[(int64_t)kc * p.ldw + nj];
        w[i][j] = (k < K && n + j < p.N) ? v : 0.f;
      }
    }
  }
}

// ---- y = act([x0 | x1 | x2] W + b) (+ res) for sample b, columns [32 vbx, 32 vbx + 32): the body of dec_linear_k<1>
// (csrc/decode.hip) with the step as a value.  combine != NULL: segment 0 is the self-attention output assembled from the
// per-chunk partials {max, sum, unnormalised o[hd]} of the previous phase.
__device__ __forceinline__ void lin_body(const satt_dec_linear_params& p, int64_t step, int vbx, int b, float* smem, int tid,
                         const satt_dec_persist_params* combine, const uint16_t* wlds, unsigned long long* dbg = nullptr) {
#define LSTAMP(i) do { if (dbg && tid == 0) dbg[i] = wall_clock64(); } while (0)
  LSTAMP(0);
  float* xs = smem;
  float* red = smem + PL_KMAX;
  const int cg = tid & 7, kl = tid >> 3;
  const int n0 = vbx * PL_COLS, n = n0 + 4 * cg, H = p.lstm_H;
  const bool bf = p.Wb != nullptr;
  int K = p.k[0];
  if (p.nseg > 1) K += p.k[1];
  if (p.nseg > 2) K += p.k[2];
  float w[PL_KI][4];
  // the weight slice of this thread goes to registers with ALL loads in flight: the kind of load (bf16 / fp32, vector / scalar) is
  // decided once, outside the unrolled loop (a branch per element serialises the loads: one L2 round trip each)
  if (wlds) {            // this member's 32 columns of W, resident in LDS as bf16 [K][32] since the start of the launch
#pragma unroll
    for (int i = 0; i < PL_KI; ++i) {
      const int k = kl + 32 * i, kc = min(k, K - 1);
      const uint2 v = *reinterpret_cast<const uint2*>(wlds + kc * PL_COLS + 4 * cg);
      const bool ok = k < K;
      w[i][0] = ok ? __uint_as_float(v.x << 16) : 0.f; w[i][1] = ok ? __uint_as_float(v.x & 0xFFFF0000u) : 0.f;
      w[i][2] = ok ? __uint_as_float(v.y << 16) : 0.f; w[i][3] = ok ? __uint_as_float(v.y & 0xFFFF0000u) : 0.f;
    }
  } else if (bf) lin_weights<true, true>(p, w, kl, n, K);
  else lin_weights<false, true>(p, w, kl, n, K);        // (rows of ldw % 4 == 0 columns only: checked on the host - code size)
  const int64_t par = step & 1;
  // epilogue operands of this thread, requested with the weights (their latency is then hidden behind the staging)
  float pf_bias = 0.f, pf_res = 0.f;
  if (!H && tid < PL_COLS && n0 + tid < p.N) {
    if (p.bias) pf_bias = p.bias[n0 + tid];
    if (p.res) pf_res = ldc(p.res + (int64_t)b * p.res_bs + step * p.res_ss + n0 + tid);
  }
  float pf_b4[4] = {0.f, 0.f, 0.f, 0.f};
  if (H && tid < 8 && p.bias) {
#pragma unroll
    for (int g = 0; g < 4; ++g) pf_b4[g] = p.bias[g * H + 8 * vbx + tid];
  }
  LSTAMP(1);
  lds_barrier();                       // the previous user of smem is done
  LSTAMP(2);
  int K0 = 0;
  for (int s = 0; s < p.nseg; ++s) {
    const int ks = p.k[s];
    if (s == 0 && combine) {
      // self-attention output of (b, head h) from the chunk partials: o = sum_c e^{m_c - M} o_c / sum_c e^{m_c - M} l_c
      const int hd = combine->D / combine->heads, nc = combine->nchunk;
      for (int d = tid; d < ks; d += PNT) {
        const int h = d / hd, dd = d - h * hd;
        const float* pp = combine->sa_part + ((int64_t)(b * combine->heads + h) * nc) * (hd + 2);
        constexpr int NCM = 16;             // nchunk <= 16 (checked on the host): every load of the row in flight at once
        float mc[NCM], lc[NCM], oc[NCM];
#pragma unroll
        for (int c = 0; c < NCM; ++c) {
          const int cc = min(c, nc - 1);
          mc[c] = ldc(pp + cc * (hd + 2)); lc[c] = ldc(pp + cc * (hd + 2) + 1); oc[c] = ldc(pp + cc * (hd + 2) + 2 + dd);
        }
        float M = -INFINITY;
#pragma unroll
        for (int c = 0; c < NCM; ++c) if (c < nc) M = fmaxf(M, mc[c]);
        float L = 0.f, o = 0.f;
#pragma unroll
        for (int c = 0; c < NCM; ++c)
          if (c < nc && mc[c] > -INFINITY) { const float e = __expf(mc[c] - M); L += e * lc[c]; o += e * oc[c]; }
        xs[K0 + d] = o / L;
      }
    } else {
      for (int k = tid; k < ks; k += PNT)
        xs[K0 + k] = ldc(p.x[s] + (int64_t)b * p.x_bs[s] + step * p.x_ss[s] + par * p.x_ps[s] + k);
    }
    K0 += ks;
  }
  float c_old = 0.f, h_old = 0.f;
  const int eu = 8 * vbx + (tid & 7);
  const bool cell = H && tid < 8;
  if (cell) {
    c_old = ldc(p.c_state + par * p.B * H + (int64_t)b * H + eu);
    h_old = ldc(p.h_state + par * p.B * H + (int64_t)b * H + eu);
  }
  lds_barrier();
  LSTAMP(3);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < PL_KI; ++i) {
    const int k = kl + 32 * i;
    const float xv = k < K ? xs[min(k, PL_KMAX - 1)] : 0.f;       // (w is zero there as well; xs may hold anything)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += xv * w[i][j];
  }
  *reinterpret_cast<float4*>(red + kl * PL_COLS + 4 * cg) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  LSTAMP(4);
  lds_barrier();
  LSTAMP(5);
  if (H) {      // ZoneoutLSTMCell, inference mode (columns regrouped by the caller: gate * 8 + unit within the block)
    if (cell) {
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float s = 0.f;
#pragma unroll 8
        for (int q = 0; q < 32; ++q) s += red[q * PL_COLS + g * 8 + tid];
        z[g] = s + pf_b4[g];
      }
      const float cn = sigmoidf_(z[2] + 1.f) * c_old + sigmoidf_(z[0]) * tanhf_(z[1]);
      const float hn = sigmoidf_(z[3]) * tanhf_(cn);
      const int64_t o = (par ^ 1) * p.B * H + (int64_t)b * H + eu;
      p.c_state[o] = (1.f - p.zc) * cn + p.zc * c_old;
      p.h_state[o] = (1.f - p.zh) * hn + p.zh * h_old;
      p.y[(int64_t)b * p.y_bs + step * p.y_ss + eu] = hn;
    }
    return;
  }
  if (tid < PL_COLS && n0 + tid < p.N) {
    float s = 0.f;
#pragma unroll 8
    for (int q = 0; q < 32; ++q) s += red[q * PL_COLS + tid];
    s += pf_bias;
    if (p.act == SATT_ACT_RELU) s = fmaxf(s, 0.f);
    else if (p.act == SATT_ACT_TANH) s = tanhf_(s);
    else if (p.act == SATT_ACT_SOFTSIGN) s = s / (1.f + fabsf(s));
    s += pf_res;
    p.y[(int64_t)b * p.y_bs + step * p.y_ss + n0 + tid] = s;
  }
}

// ---- energies of memory rows [sl R, sl R + R) of sample b (R <= 8: 4 waves x 2 row passes); pq from the query-layer phase
template <int F>
__device__ __forceinline__ void energy_body(const satt_dec_attention_params& p, const float* __restrict__ pqg, int t, int b, int sl, int R,
                            float* smem, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int Ti = p.Ti, U1 = p.U1, U2 = p.U2, UQ = U1 + U2, KW = p.kernel, PL = (KW - 1) / 2, r0 = sl * R;
  float* pq = smem; float* ftab = pq + UQ; float* aw = ftab + KW * F + F; float* fl = aw + R + KW;
  const int len = (int)p.lengths[b];
  const float* ga = p.a_state + ((int64_t)(t & 1) * p.B + b) * Ti;
  lds_barrier();
  for (int i = tid; i < R + KW; i += PNT) {
    const int tt = r0 + i - PL;
    aw[i] = (tt >= 0 && tt < Ti) ? ldc(ga + tt) : 0.f;
  }
  for (int i = tid; i < UQ; i += PNT) pq[i] = ldc(pqg + (int64_t)b * UQ + i);
  for (int i = tid; i < KW * F + F; i += PNT) ftab[i] = i < KW * F ? p.locF[i] : p.locFb[i - KW * F];
  const int d0 = lane * 4;
  float v1r[4], b1r[4], Ur[F][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool ok = d0 + q < U1;
    v1r[q] = ok ? p.v1[d0 + q] : 0.f;
    b1r[q] = ok ? p.b1[d0 + q] : 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f) Ur[f][q] = ok ? p.locU[f * U1 + d0 + q] : 0.f;
  }
  const float v2r = lane < U2 ? p.v2[lane] : 0.f;
  const float* k1 = p.keys1 + (int64_t)b * Ti * U1;
  const float* k2 = U2 ? p.keys2 + (int64_t)b * Ti * U2 : nullptr;
  constexpr int RP = 2;
  float4 kk[RP]; float kk2[RP];
#pragma unroll
  for (int u = 0; u < RP; ++u) {
    const int i = wave + 4 * u, tt = r0 + i;
    kk[u] = make_float4(0.f, 0.f, 0.f, 0.f); kk2[u] = 0.f;
    if (i < R && tt < len) {
      if (d0 < U1) kk[u] = *reinterpret_cast<const float4*>(k1 + (int64_t)tt * U1 + d0);
      if (lane < U2) kk2[u] = k2[(int64_t)tt * U2 + lane];
    }
  }
  lds_barrier();
  for (int i = tid; i < R * F; i += PNT) {
    const int rr = i / F, f = i - rr * F;
    float s = ftab[KW * F + f];
    for (int j = 0; j < KW; ++j) s += aw[rr + j] * ftab[j * F + f];
    fl[i] = s;
  }
  lds_barrier();
  float c1r[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) c1r[q] = d0 + q < U1 ? b1r[q] + pq[d0 + q] : 0.f;
  const float pq2 = lane < U2 ? pq[U1 + lane] : 0.f;
  float acc[RP], acc2[RP];
#pragma unroll
  for (int u = 0; u < RP; ++u) {
    const int i = min(wave + 4 * u, R - 1);
    const float kq[4] = {kk[u].x, kk[u].y, kk[u].z, kk[u].w};
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float x = kq[q] + c1r[q];
#pragma unroll
      for (int f = 0; f < F; ++f) x += fl[i * F + f] * Ur[f][q];
      a += v1r[q] * tanhf_(x);
    }
    acc[u] = a;
    acc2[u] = v2r * tanhf_(kk2[u] + pq2);
  }
  wave_sum_multi<RP>(acc);
  wave_sum_multi<RP>(acc2);
  if (lane == 0) {
#pragma unroll
    for (int u = 0; u < RP; ++u) {
      const int i = wave + 4 * u, tt = r0 + i;
      if (i < R && tt < len) { p.e1[(int64_t)b * Ti + tt] = acc[u]; if (U2) p.e2[(int64_t)b * Ti + tt] = acc2[u]; }
    }
  }
}

// ---- masked softmax + forward recursion (recomputed per member) and context columns [32 cs, 32 cs + 32) of sample b
__device__ __forceinline__ void context_body(const satt_dec_attention_params& p, int t, int b, int cs, float* smem, int tid) {
  const int Ti = p.Ti, V1 = p.V1, V2 = p.V2, CT = V1 + V2;
  float* a1 = smem; float* a2 = a1 + Ti; float* alphap = a2 + Ti; float* aold = alphap + Ti;
  float* part = aold + Ti; float* sm = part + 32 * 32;
  const int par = t & 1, len = (int)p.lengths[b];
  const bool forced = p.teach1 != nullptr, dual = V2 > 0;
  const int64_t row = ((int64_t)b * p.Td + t) * Ti;
  const int cg = cs * 8 + (tid & 7), rg = tid >> 3, col = 4 * cg;
  const bool s1c = col < V1, cok = col < CT;
  const float* vv = s1c ? p.values1 + (int64_t)b * Ti * V1 + col : (cok ? p.values2 + (int64_t)b * Ti * V2 + (col - V1) : nullptr);
  const int ld = s1c ? V1 : V2;
  constexpr int NR = 8;
  float4 x[NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int tt = rg + 32 * u;
    x[u] = (cok && tt < len) ? *reinterpret_cast<const float4*>(vv + (int64_t)tt * ld) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  lds_barrier();
  if (!forced) {
    for (int i = tid; i < Ti; i += PNT) {
      a1[i] = i < len ? ldc(p.e1 + (int64_t)b * Ti + i) : -INFINITY;
      a2[i] = (dual && i < len) ? ldc(p.e2 + (int64_t)b * Ti + i) : -INFINITY;
      alphap[i] = ldc(p.alpha_state + ((int64_t)par * p.B + b) * Ti + i);
      aold[i] = ldc(p.a_state + ((int64_t)par * p.B + b) * Ti + i);
    }
  } else {
    for (int i = tid; i < Ti; i += PNT) { a1[i] = p.teach1[row + i]; a2[i] = p.teach2 ? p.teach2[row + i] : 0.f; }
  }
  lds_barrier();
  float* ga_n = p.a_state + ((int64_t)(par ^ 1) * p.B + b) * Ti;
  float* gal_n = p.alpha_state + ((int64_t)(par ^ 1) * p.B + b) * Ti;
  if (!forced) {
    float m1 = -INFINITY, m2 = -INFINITY;
    for (int i = tid; i < len; i += PNT) { m1 = fmaxf(m1, a1[i]); m2 = fmaxf(m2, a2[i]); }
    m1 = bmax4(m1, sm, tid); m2 = bmax4(m2, sm, tid);
    float s1 = 0.f, s2 = 0.f;
    for (int i = tid; i < Ti; i += PNT) {
      const float x1 = i < len ? __expf(a1[i] - m1) : 0.f, x2 = (dual && i < len) ? __expf(a2[i] - m2) : 0.f;
      a1[i] = x1; a2[i] = x2; s1 += x1; s2 += x2;
    }
    s1 = bsum4(s1, sm, tid); s2 = bsum4(s2, sm, tid);
    const float r1 = 1.f / s1, r2 = dual ? 1.f / s2 : 0.f;
    float sa = 0.f;
    for (int i = tid; i < Ti; i += PNT) {
      const float a = a1[i] * r1;
      a2[i] *= r2;
      if (cs == 0) ga_n[i] = p.cumulative ? a + aold[i] : a;
      float al = a;
      if (p.att1_mode == 0) {
        al = (0.5f * alphap[i] + 0.5f * (i > 0 ? alphap[i - 1] : 0.f) + 1e-7f) * a;
        sa += al;
      }
      a1[i] = al;
    }
    if (p.att1_mode == 0) {
      sa = bsum4(sa, sm, tid);
      const float rs = 1.f / sa;
      for (int i = tid; i < Ti; i += PNT) a1[i] *= rs;
    }
    lds_barrier();
  }
  if (cs == 0) {
    for (int i = tid; i < Ti; i += PNT) {
      gal_n[i] = a1[i];
      if (forced) ga_n[i] = a1[i];
      p.align1[row + i] = a1[i];
      if (p.align2) p.align2[row + i] = a2[i];
    }
  }
  const float* al = s1c ? a1 : a2;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int tt = rg + 32 * u;
    const float w = tt < len ? al[tt] : 0.f;
    acc.x += w * x[u].x; acc.y += w * x[u].y; acc.z += w * x[u].z; acc.w += w * x[u].w;
  }
  for (int t0 = 32 * NR; t0 < len; t0 += 32 * NR) {
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int tt = t0 + rg + 32 * u;
      x[u] = (cok && tt < len) ? *reinterpret_cast<const float4*>(vv + (int64_t)tt * ld) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int tt = t0 + rg + 32 * u;
      const float w = tt < len ? al[tt] : 0.f;
      acc.x += w * x[u].x; acc.y += w * x[u].y; acc.z += w * x[u].z; acc.w += w * x[u].w;
    }
  }
  *reinterpret_cast<float4*>(part + (rg * 8 + (tid & 7)) * 4) = acc;
  lds_barrier();
  if (tid < 32) {
    const int c = cs * 32 + tid;
    if (c < CT) {
      float s = 0.f;
#pragma unroll 8
      for (int g = 0; g < 32; ++g) s += part[g * 32 + tid];
      p.ctx[((int64_t)par * p.B + b) * CT + c] = s;       // double-buffered by step parity, as in the graph form
    }
  }
}

// ---- self-attention partials of (sample b, head h) over cache rows [c CH, c CH + CH) & [0, t]: {max, sum e^{s - max},
// sum e^{s - max} V row}; a wave per row (any head depth), then a thread per (dim, row group)
__device__ __forceinline__ void satt_body(const satt_dec_persist_params& P, int t, int b, int h, int c, float* smem, int tid) {
  const int hd = P.D / P.heads, CH = P.chunk, D = P.D, lane = tid & 63, wave = tid >> 6;
  float* q = smem; float* s = q + hd; float* part = s + CH; float* sm = part + PNT;
  const float* base = P.kvq + (int64_t)b * P.Td * 3 * D + h * hd;
  float* out = P.sa_part + ((int64_t)(b * P.heads + h) * P.nchunk + c) * (hd + 2);
  const int j0 = c * CH, j1 = min(j0 + CH, t + 1);        // rows [j0, j1)
  lds_barrier();
  if (j1 <= j0) {
    if (tid == 0) { out[0] = -INFINITY; out[1] = 0.f; }
    return;
  }
  for (int i = tid; i < hd; i += PNT) q[i] = ldc(base + (int64_t)t * 3 * D + 2 * D + i);
  lds_barrier();
  for (int jr = wave; jr < j1 - j0; jr += 4 * 4) {
    float acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + jr + 4 * u;
      float a = 0.f;
      if (j < j1)
        for (int d = lane; d < hd; d += 64) a += q[d] * ldc(base + (int64_t)j * 3 * D + d);
      acc[u] = a;
    }
    wave_sum_multi<4>(acc);
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) if (jr + 4 * u < j1 - j0) s[jr + 4 * u] = acc[u] * P.scale;
    }
  }
  lds_barrier();
  const int n = j1 - j0;
  float m = -INFINITY;
  for (int j = tid; j < n; j += PNT) m = fmaxf(m, s[j]);
  m = bmax4(m, sm, tid);
  float z = 0.f;
  for (int j = tid; j < n; j += PNT) { const float e = __expf(s[j] - m); s[j] = e; z += e; }
  z = bsum4(z, sm, tid);
  const int ng = PNT / hd > 0 ? PNT / hd : 1, d = tid % hd, g = tid / hd;
  float acc = 0.f;
  if (g < ng)
    for (int j = g; j < n; j += ng) acc += s[j] * ldc(base + (int64_t)(j0 + j) * 3 * D + D + d);
  part[tid] = acc;
  lds_barrier();
  if (tid < hd) {
    float o = 0.f;
    for (int gg = 0; gg < ng; ++gg) o += part[gg * hd + tid];
    out[2 + tid] = o;
  }
  if (tid == 0) { out[0] = m; out[1] = z; }
}

__global__ __launch_bounds__(PNT) void dec_persist_k(const satt_dec_persist_params Pv) {
  if (blockIdx.x & 7) return;                       // members are the workgroups the dispatcher places on XCD 0
  __shared__ __attribute__((aligned(16))) float smem[PSMEM];
  __shared__ int dead_s;
  // The step program (3.6 KB of descriptors) is indexed with run-time phase numbers: left in the kernel-argument struct that
  // makes the compiler copy it to scratch (3600 bytes per lane), and the kernarg segment itself is host memory.  One copy
  // into LDS at the start instead: every later access is an LDS read.
  __shared__ __attribute__((aligned(16))) satt_dec_persist_params P;
  // LDS-resident weight slices (dynamic LDS): the three LSTM matrices are 4.0 of the 4.85 MB a step reads, more than the
  // 4 MB L2 of the one XCD the members share - streamed from L2 every step they evict each other and every weight load goes
  // to the memory-side cache (lin phases of 10-13 us).  Each member keeps its 32 columns of each of them in LDS instead
  // (bf16 [K][32]: 124 KB for the LJSpeech model); the small matrices (0.9 MB) then stay in L2.
  extern __shared__ __attribute__((aligned(16))) uint16_t wres[];
  __shared__ int wres_off[SATT_DEC_MAX_LIN];
  {
    typedef const __attribute__((address_space(4))) uint32_t* kptr;
    const kptr src = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t* dst = reinterpret_cast<uint32_t*>(&P);
    for (int i = threadIdx.x; i < (int)(sizeof(satt_dec_persist_params) / 4); i += PNT) dst[i] = src[i];
  }
  __syncthreads();
  const int me = blockIdx.x >> 3, G = P.G, tid = threadIdx.x;
  u64* bar = reinterpret_cast<u64*>(P.ws);
  unsigned int* err = reinterpret_cast<unsigned int*>(bar + 2 * G);
  if (tid == 0) dead_s = 0;
  __syncthreads();
  // placement handshake: every member publishes its XCC id; all must agree
  if (tid == 0) gput(bar + G + me, 1u, __int_as_float(xcc_id()), true);
  if (tid < 64) {
    const gu64* g = (const gu64*)(bar + G + min(tid, G - 1));
    u64 x = 0;
    for (unsigned spins = 0;; ++spins) {
      x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all((uint32_t)(x >> 32) == 1u)) break;
      if (spins > (1u << 20)) { dead_s = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    const int mine = xcc_id();
    if (!__all((int)(uint32_t)x == mine)) dead_s = 1;
    if (dead_s && tid == 0) __hip_atomic_store((gu32*)err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (dead_s) return;          // not co-resident / not on one XCD: nothing was computed, the host falls back to the graph
  {   // fill the resident slices (plain cached loads: the weights are read-only)
    int off = 0;
    for (int a = 0; a < SATT_DEC_MAX_LIN; ++a) {
      const satt_dec_linear_params& L = P.lin[a];
      int K = 0;
      for (int q = 0; q < L.nseg && q < 3; ++q) K += L.k[q];
      const bool res = a < P.nlin_used && L.lstm_H > 0 && L.Wb != nullptr && (L.N / PL_COLS) == G && P.wres_elems >= off + K * PL_COLS;
      if (tid == 0) wres_off[a] = res ? off : -1;
      if (res) {
        for (int e = tid; e < K * (PL_COLS / 4); e += PNT) {
          const int k = e / (PL_COLS / 4), c4 = e - k * (PL_COLS / 4);
          *reinterpret_cast<uint2*>(wres + off + k * PL_COLS + 4 * c4) =
              *reinterpret_cast<const uint2*>(L.Wb + (int64_t)k * L.ldw + me * PL_COLS + 4 * c4);
        }
        off += K * PL_COLS;
      }
    }
  }
  __syncthreads();
  uint32_t seq = 0;
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(bar + 2 * G) + 64);
  const satt_dec_attention_params& A = P.att;
  const int ncs = ((A.V1 + A.V2) / 4 + 7) / 8;
  for (int t = P.t0; t < P.t1; ++t) {
    if (P.flag && t > P.t0) {
      const int f = (int)__hip_atomic_load((const gu32*)P.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (f) break;
    }
    const bool stamp = me == 0 && tid == 0 && t == P.t0 + 2;
    for (int ph = 0; ph < P.nphase; ++ph) {
      const int kind = P.phase_kind[ph], arg = P.phase_arg[ph];
      if (stamp) stamps[2 * ph] = wall_clock64();
      if (kind == 0) {
        const satt_dec_linear_params& L = P.lin[arg];
        const int ncol = (L.N + PL_COLS - 1) / PL_COLS, nvb = ncol * L.B;
        const uint16_t* wl = wres_off[arg] >= 0 ? wres + wres_off[arg] : nullptr;      // resident: column block == me
        for (int vb = me; vb < nvb; vb += G)
          lin_body(L, (int64_t)t, vb % ncol, vb / ncol, smem, tid, arg == P.combine_lin ? &P : nullptr, wl,
                   (stamp && (ph == 0 || ph == 2)) ? stamps + 32 + 8 * (ph / 2) : nullptr);
        // step bookkeeping of the graph form, carried by member 0: the stop rule of the PREVIOUS step (helpers.py:103-107)
        if (L.stop && me == 0 && tid < 64 && t >= 1) {
          bool ok = true;
          for (int b = tid; b < L.B; b += 64) {
            const float sgm = 1.f / (1.f + __expf(-ldc(L.stop + (int64_t)b * L.stop_bs + (int64_t)(t - 1) * L.stop_ss)));
            ok = ok && (sgm > L.stop_threshold);
          }
          const bool all = __ballot(!ok) == 0ull;
          if (tid == 0 && all && (t - 1) > L.min_steps && *L.flag == 0) *L.flag = t;
        }
      } else if (kind == 1) {
        if (!A.teach1) {
          const int NS = P.nslice, R = (A.Ti + NS - 1) / NS;
          for (int vb = me; vb < NS * A.B; vb += G) energy_body<5>(A, P.pq, t, vb / NS, vb % NS, R, smem, tid);
        }
      } else if (kind == 2) {
        for (int vb = me; vb < ncs * A.B; vb += G) context_body(A, t, vb / ncs, vb % ncs, smem, tid);
      } else {
        const int nv = A.B * P.heads * P.nchunk;
        for (int vb = me; vb < nv; vb += G) {
          const int c = vb % P.nchunk, bh = vb / P.nchunk;
          satt_body(P, t, bh / P.heads, bh % P.heads, c, smem, tid);
        }
      }
      if (stamp) stamps[2 * ph + 1] = wall_clock64();
      grid_barrier(bar, G, me, ++seq, err, &dead_s, tid);
      if (stamp && ph + 1 == P.nphase) stamps[2 * ph + 2] = wall_clock64();
    }
  }
}

}  // namespace

// [2 G granules] [64 B: error word] [64 x 8 B: time stamps of member 0 around the phases of the third step (diagnostics)]
extern "C" int64_t satt_dec_persist_ws_bytes(int G) { return (int64_t)sizeof(u64) * 2 * G + 64 + 64 * 8; }

/* host-synchronous: 0 = the last launch on `ws` ran to its end, 1 = a grid barrier timed out, 2 = the members were not
 * co-resident on one XCD (nothing was computed) */
extern "C" int satt_dec_persist_status(const void* ws, int G, void* stream, int* status) {
  if (!ws || !status) return SATT_E_BADARG;
  unsigned int v = 0;
  if (hipMemcpyAsync(&v, (const char*)ws + sizeof(u64) * 2 * G, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess)
    return SATT_E_LAUNCH;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return SATT_E_LAUNCH;
  *status = (int)v;
  return SATT_OK;
}

extern "C" int satt_dec_persist(const satt_dec_persist_params* pp, void* stream) {
  if (!pp) return SATT_E_BADARG;
  const satt_dec_persist_params& P = *pp;
  if (P.B <= 0 || P.G < 1 || P.G > 32 || P.nphase < 1 || P.nphase > SATT_DEC_MAX_PHASES || !P.ws || P.t0 < 0 || P.t1 <= P.t0)
    return SATT_E_BADARG;
  const satt_dec_attention_params& A = P.att;
  bool has_att = false, has_sa = false;
  for (int ph = 0; ph < P.nphase; ++ph) {
    const int kind = P.phase_kind[ph], arg = P.phase_arg[ph];
    if (kind < 0 || kind > 3) return SATT_E_BADARG;
    if (kind == 0) {
      if (arg < 0 || arg >= SATT_DEC_MAX_LIN) return SATT_E_BADARG;
      const satt_dec_linear_params& L = P.lin[arg];
      int K = 0;
      if (L.B != P.B || L.N <= 0 || L.nseg < 1 || L.nseg > 3 || !L.y || (!L.W && !L.Wb)) return SATT_E_BADARG;
      for (int s = 0; s < L.nseg; ++s) { if ((!L.x[s] && !(s == 0 && arg == P.combine_lin)) || L.k[s] <= 0) return SATT_E_BADARG; K += L.k[s]; }
      if (K > PL_KMAX) return SATT_E_UNSUPPORTED;
      // 16-byte / 8-byte weight loads only (the scalar variants would cost 16 KB of code: the kernel has to fit the instruction
      // cache); N may end inside a 4-column group when the caller pads the rows with zero columns
      if (L.ldw % 4 || ((uintptr_t)(L.Wb ? (const void*)L.Wb : (const void*)L.W)) % (L.Wb ? 8 : 16)) return SATT_E_UNSUPPORTED;
      if (L.lstm_H && (L.N != 4 * L.lstm_H || L.lstm_H % 8 || !L.c_state || !L.h_state || L.N % 4 || L.ldw % 4)) return SATT_E_BADARG;
      if (L.stop && !L.flag) return SATT_E_BADARG;
    } else if (kind == 3) has_sa = true; else has_att = true;
  }
  if (has_att) {
    if (A.B != P.B || A.Ti <= 0 || A.Td < P.t1 || A.kernel < 1 || !A.values1 || !A.ctx || !A.align1 || !A.a_state || !A.alpha_state ||
        !A.lengths || (!A.teach1 && (!A.keys1 || !P.pq || !A.e1 || (A.U2 > 0 && !A.e2))))
      return SATT_E_BADARG;
    if (A.agentW) return SATT_E_UNSUPPORTED;       // the transition agent is implemented in the graph form only
    if (A.filters != 5 || A.U1 > 256 || A.U2 > 64 || A.U1 % 4 || A.U2 % 4 || A.V1 % 4 || A.V2 % 4) return SATT_E_UNSUPPORTED;
    if (P.nslice < 1 || (A.Ti + P.nslice - 1) / P.nslice > 8) return SATT_E_BADARG;
    const int R = (A.Ti + P.nslice - 1) / P.nslice;
    if (A.U1 + A.U2 + A.kernel * 5 + 5 + R + A.kernel + R * 5 > PSMEM || 4 * A.Ti + 32 * 32 + 8 > PSMEM) return SATT_E_UNSUPPORTED;
  }
  if (has_sa) {
    if (!P.kvq || !P.sa_part || P.heads <= 0 || P.D % P.heads || P.nchunk < 1 || P.nchunk > 16 || P.chunk < 1 || P.nchunk * P.chunk < P.t1 || P.Td < P.t1)
      return SATT_E_BADARG;
    const int hd = P.D / P.heads;
    if (hd > PNT || PNT % hd || hd + P.chunk + PNT + 8 > PSMEM) return SATT_E_UNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(P.ws, 0, (size_t)satt_dec_persist_ws_bytes(P.G), s) != hipSuccess) return SATT_E_LAUNCH;
  const size_t dyn = sizeof(uint16_t) * (size_t)(P.wres_elems > 0 ? P.wres_elems : 0);
  if (dyn > 128 * 1024) return SATT_E_BADARG;
  if (dyn > 0)
    (void)hipFuncSetAttribute((const void*)dec_persist_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
  hipLaunchKernelGGL(dec_persist_k, dim3(8 * P.G), dim3(PNT), dyn, s, P);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}
