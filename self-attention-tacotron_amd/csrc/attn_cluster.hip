// Cluster form of the dual-source attention RNN loop: C workgroups per sample (grid (B, C): with B % 8 == 0 the
// cluster of a sample sits on one XCD).  The two dominant costs of the single-workgroup kernel are split C ways,
// everything cheap stays REDUNDANT and bitwise identical in every member of the cluster, so only few exchanges
// per step are needed (8-byte {tag,value} granules, parity double-buffered, bounded spins — see lstm_cluster.hip):
//   forward : [ctx|h] x Wrec split by gate columns  -> X1: all-gather (h', h_state)          (2A floats)
//             energies split by memory rows (t' mod C) -> X2: all-gather (e1, e2)             (2 len floats)
//   backward: d alpha split by memory rows          -> Xb: all-gather (d alpha, d a2)         (2 Ti floats)
//             energy backward split by memory rows  -> Xd: all-reduce d pq (C partials x UQ) + all-gather dfl rows
//             dz x Wrec^T split by output columns   -> Xh: all-gather d[ctx|h]                (CT+A floats)
// Weight slices are pre-packed per cluster member (satt_attn_cluster_pack) so every member streams a contiguous
// [K][NL] bf16 matrix from L2 (1.1 MB / C per step instead of 1.1 MB).
#include "attn_common.h"

namespace {

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

__device__ __forceinline__ void gput(u64* g, uint32_t tag, float v) {
  __hip_atomic_store((gu64*)g, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ONE wave gathers granules src[0..count) (count <= 256) carrying `tag`, calling store(i, value) for each.
template <class St>
__device__ __forceinline__ void gather_chunk(u64* src, int count, uint32_t tag, int lane, St store,
                                             unsigned int* err_word, int* dead) {
  float v[4]; bool ok[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { v[q] = 0.f; ok[q] = (lane + 64 * q) >= count; }
  if (!*dead) {
    for (unsigned spins = 0;; ++spins) {
      bool all_ok = true;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!ok[q]) {
          const u64 x = __hip_atomic_load((gu64*)(src + lane + 64 * q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((uint32_t)(x >> 32) == tag) { v[q] = __uint_as_float((uint32_t)x); ok[q] = true; }
          else all_ok = false;
        }
      }
      if (__all(all_ok)) break;
      if (spins > (1u << 21)) {
        if (lane == 0) __hip_atomic_store((gu32*)err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *dead = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) { const int i = lane + 64 * q; if (i < count) store(i, v[q]); }
}
// all AW waves cooperate: chunk k (256 granules) is gathered by wave k % AW
template <class St>
__device__ __forceinline__ void gather_all(u64* src, int n, uint32_t tag, int wave, int lane, St store,
                                           unsigned int* err_word, int* dead) {
  for (int c0 = wave * 256; c0 < n; c0 += AW * 256)
    gather_chunk(src + c0, min(256, n - c0), tag, lane, [&](int i, float v) { store(c0 + i, v); }, err_word, dead);
}

struct WsLayout {   // granule offsets (per sample, per parity) inside the workspace
  int x1, x2, xb, xd, xh, per_parity;
};
__host__ __device__ inline int nwp_of(int K, int C) { return (((K + C - 1) / C) + 7) & ~7; }
__host__ __device__ inline WsLayout ws_layout(int A, int Ti, int C, int UQ, int F, int K) {
  WsLayout w; int o = 0;
  w.x1 = o; o += 2 * A;
  w.x2 = o; o += 2 * Ti;
  w.xb = o; o += 2 * Ti;
  w.xd = o; o += C * UQ + Ti * F;
  w.xh = o; o += C * nwp_of(K, C);
  w.per_parity = o;
  return w;
}

struct SmemCF {
  int vec, z, q, pq, aprev, alA, alB, e1, e2, fl, Fs, bFs, partial, dead, kofs, vofs, total;
};
__host__ __device__ inline SmemCF carve_cf(int A, int CT, int UQ, int Ti, int F, int KW, int NL, int nown, bool klds) {
  auto u = [](int x) { return (x + 3) & ~3; };
  SmemCF s; int o = 0;
  s.vec = o; o += u(CT + A); s.z = o; o += u(NL); s.q = o; o += u(A); s.pq = o; o += u(UQ);
  s.aprev = o; o += u(Ti); s.alA = o; o += u(Ti); s.alB = o; o += u(Ti); s.e1 = o; o += u(Ti); s.e2 = o; o += u(Ti);
  s.fl = o; o += u(Ti * F); s.Fs = o; o += u(KW * F); s.bFs = o; o += u(F);
  s.partial = o; o += ANT * 8;
  s.dead = o; o += 4;
  s.kofs = o; if (klds) o += u((nown * UQ + 1) / 2);
  s.vofs = o; if (klds) o += u((Ti * CT + 1) / 2);          // bf16 values1 [Ti][V1] then values2 [Ti][V2]
  s.total = o;
  return s;
}

template <int F, bool KLDS>
__global__ __launch_bounds__(ANT) void attn_cluster_fwd_k(const satt_attn_cluster_params cp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const satt_attn_rnn_params& p = cp.f;
  const int C = cp.C;
  const int A = p.A, G = 4 * A, V1 = p.V1, V2 = p.V2, CT = V1 + V2, U1 = p.U1, U2 = p.U2, UQ = U1 + U2;
  const int Ti = p.Ti, Td = p.Td, KW = p.kernel, PL = (KW - 1) / 2;
  const int AU = A / C, NL = 4 * AU, KR = CT + A;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x, c = blockIdx.y;
  const int nown_max = (Ti + C - 1) / C;
  const SmemCF L = carve_cf(A, CT, UQ, Ti, F, KW, NL, nown_max, KLDS);
  float* vec = smem + L.vec;        // [CT + A]  ctx1 | ctx2 | h_state   (full, replicated)
  float* z = smem + L.z;            // [NL]      own gate pre-activations
  float* q = smem + L.q;            // [A]       query = h' (full after X1)
  float* pq = smem + L.pq;          // [UQ]
  float* aprev = smem + L.aprev;    // [Ti]
  float* alA = smem + L.alA;
  float* alB = smem + L.alB;
  float* e1 = smem + L.e1;          // [Ti] full after X2
  float* e2 = smem + L.e2;
  float* fl = smem + L.fl;          // [Ti*F] (own rows only are valid)
  float* Fs = smem + L.Fs;
  float* bFs = smem + L.bFs;
  float* partial = smem + L.partial;
  int* dead = reinterpret_cast<int*>(smem + L.dead);
  uint16_t* K1s = reinterpret_cast<uint16_t*>(smem + L.kofs);   // bf16 [nown][U1]  (local row i <-> t' = c + C*i)
  uint16_t* K2s = K1s + nown_max * U1;
  uint16_t* V1s = reinterpret_cast<uint16_t*>(smem + L.vofs);   // bf16 [Ti][V1]
  uint16_t* V2s = V1s + Ti * V1;                                 // bf16 [Ti][V2]

  const int len = (int)p.lengths[b];
  const uint32_t seed = p.seed ? *p.seed : 0u;
  const float* xg = p.xg + (size_t)b * Td * G;
  const float* keys1 = p.keys1 + (size_t)b * Ti * U1;
  const float* values1 = p.values1 + (size_t)b * Ti * V1;
  const float* keys2 = p.keys2 + (size_t)b * Ti * U2;
  const float* values2 = p.values2 + (size_t)b * Ti * V2;
  const int OW = A + CT;
  float* out = p.out + (size_t)b * Td * OW;
  const uint16_t* Wslice = cp.WrecP + (size_t)c * KR * NL;
  const WsLayout WL = ws_layout(A, Ti, C, UQ, F, KR);
  u64* wsb = reinterpret_cast<u64*>(cp.ws);
  unsigned int* err_word = reinterpret_cast<unsigned int*>(wsb + (size_t)2 * p.B * WL.per_parity);
  const int nown = len > c ? (len - c + C - 1) / C : 0;         // own memory rows: t' = c + C*i < len

  const int d0 = lane * NQ;
  const bool actU = d0 < U1, actV = d0 < V1;
  float v1r[NQ], b1r[NQ], Ur[NQ][F];
#pragma unroll
  for (int qq = 0; qq < NQ; ++qq) {
    const int d = d0 + qq;
    v1r[qq] = d < U1 ? p.v1[d] : 0.f;
    b1r[qq] = d < U1 ? p.b1[d] : 0.f;
#pragma unroll
    for (int k = 0; k < F; ++k) Ur[qq][k] = d < U1 ? p.locU[k * U1 + d] : 0.f;
  }
  const float v2r = lane < U2 ? p.v2[lane] : 0.f;

  for (int i = tid; i < CT + A; i += ANT) vec[i] = 0.f;
  for (int i = tid; i < Ti; i += ANT) { aprev[i] = 0.f; alA[i] = (i == 0) ? 1.f : 0.f; e1[i] = 0.f; e2[i] = 0.f; }
  for (int i = tid; i < KW * F; i += ANT) Fs[i] = p.locF[i];
  if (tid < F) bFs[tid] = p.locFb[tid];
  if (tid == 0) *dead = 0;
  if (KLDS) {
    for (int e = tid; e < nown * U1; e += ANT) { const int i = e / U1, d = e - i * U1; K1s[e] = f2bf(keys1[(size_t)(c + C * i) * U1 + d]); }
    for (int e = tid; e < nown * U2; e += ANT) { const int i = e / U2, d = e - i * U2; K2s[e] = f2bf(keys2[(size_t)(c + C * i) * U2 + d]); }
    for (int e = tid; e < len * V1; e += ANT) V1s[e] = f2bf(values1[e]);
    for (int e = tid; e < len * V2; e += ANT) V2s[e] = f2bf(values2[e]);
  }
  float cst = 0.f, hst = 0.f;
  float* alp = alA;
  float* aln = alB;
  if (cp.t0 > 0) {        // chunked launch: restart from the tensors saved by the previous chunk at step t0-1
    const size_t bp = (size_t)b * Td + cp.t0 - 1;
    __syncthreads();
    for (int i = tid; i < CT; i += ANT) vec[i] = out[(size_t)(cp.t0 - 1) * OW + A + i];
    for (int i = tid; i < A; i += ANT) vec[CT + i] = p.hstate[bp * A + i];
    for (int i = tid; i < Ti; i += ANT) { aprev[i] = p.a1[bp * Ti + i]; alA[i] = p.align1[bp * Ti + i]; }
    if (tid < AU) { cst = p.cstate[bp * A + c * AU + tid]; hst = p.hstate[bp * A + c * AU + tid]; }
  }
  __syncthreads();

  PROF_DECL;
  for (int t = cp.t0; t < cp.t1; ++t) {
    PROF(0);
    const size_t bt = (size_t)b * Td + t;
    const uint32_t tag = (uint32_t)(t + 1);
    u64* wp = wsb + ((size_t)(t & 1) * p.B + b) * WL.per_parity;
    float xi = 0.f, xj = 0.f, xf = 0.f, xo = 0.f;
    if (tid < AU) {
      const float* xr = xg + (size_t)t * G + c * AU + tid;
      xi = xr[0]; xj = xr[A]; xf = xr[2 * A]; xo = xr[3 * A];
    }
    // (1) own gate columns: [ctx_{t-1} | h_{t-1}] x Wrec[:, own]
    matvec_bf16<ANT, MVU>(vec, Wslice, KR, NL, partial, z);
    PROF(1);
    // (2) LSTM cell for own units, publish (h', h_state)
    if (tid < AU) {
      const int j = c * AU + tid;
      const float gi = sigmoidf_(xi + z[tid]);
      const float gj = tanhf_(xj + z[AU + tid]);
      const float gf = sigmoidf_(xf + z[2 * AU + tid] + 1.0f);
      const float go = sigmoidf_(xo + z[3 * AU + tid]);
      const float cn = gf * cst + gi * gj;
      const float hn = go * tanhf_(cn);
      const uint32_t idx = (uint32_t)bt * (uint32_t)A + (uint32_t)j;
      if (p.training) {
        if (p.zc_thresh == 0 || satt_keep(seed, p.stream_c, idx, p.zc_thresh)) cst = cn;
        if (p.zh_thresh == 0 || satt_keep(seed, p.stream_h, idx, p.zh_thresh)) hst = hn;
      } else {
        cst = (1.f - p.zc) * cn + p.zc * cst;
        hst = (1.f - p.zh) * hn + p.zh * hst;
      }
      gput(wp + WL.x1 + j, tag, hn);
      gput(wp + WL.x1 + A + j, tag, hst);
      float* gr = p.gates + bt * G;
      gr[j] = gi; gr[A + j] = gj; gr[2 * A + j] = gf; gr[3 * A + j] = go;
      p.cnew[bt * A + j] = cn;
      p.cstate[bt * A + j] = cst;
      p.hstate[bt * A + j] = hst;
      out[(size_t)t * OW + j] = hn;
    }
    // X1: gather h' -> q, h_state -> vec[CT..]
    gather_all(wp + WL.x1, 2 * A, tag, wave, lane,
               [&](int i, float v) { if (i < A) q[i] = v; else vec[CT + (i - A)] = v; }, err_word, dead);
    __syncthreads();
    PROF(2);
    // (3) processed queries (redundant in every member: identical inputs, identical arithmetic)
    matvec_bf16<ANT, MVU>(q, p.Wq, A, UQ, partial, pq);
    PROF(3);
    if (c == 0 && tid < UQ) p.pq[bt * UQ + tid] = pq[tid];
    // (4) location features for own rows
    {
      float* flg = p.fl + bt * Ti * F;
      for (int e = tid; e < nown * F; e += ANT) {
        const int i = e / F, k = e - i * F, tt = c + C * i;
        float s = bFs[k];
        for (int jj = 0; jj < KW; ++jj) {
          const int src = tt + jj - PL;
          if (src >= 0 && src < Ti) s += aprev[src] * Fs[jj * F + k];
        }
        fl[tt * F + k] = s; flg[tt * F + k] = s;
      }
      // rows beyond the sequence length are never read back, but keep the saved tensor defined
      if (c == 0) for (int e = tid + len * F; e < Ti * F; e += ANT) flg[e] = 0.f;
    }
    __syncthreads();
    PROF(4);
    // (5) energies of own rows, publish
    {
      float pqb[NQ];
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) pqb[qq] = (d0 + qq) < U1 ? pq[d0 + qq] + b1r[qq] : 0.f;
      const float pq2 = lane < U2 ? pq[U1 + lane] : 0.f;
      for (int i0 = wave; i0 < nown; i0 += RB * AW) {
        float red[2 * RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const int i = i0 + u * AW, tt = c + C * i;
          float acc = 0.f, acc2 = 0.f;
          if (i < nown) {
            float kk[NQ];
            load_key4<KLDS>(keys1 + (KLDS ? 0 : (size_t)tt * U1), K1s, KLDS ? i : 0, U1, d0, actU, kk);
            const float k2 = load_key1<KLDS>(keys2 + (KLDS ? 0 : (size_t)tt * U2), K2s, KLDS ? i : 0, U2, lane, lane < U2);
            float f[F];
#pragma unroll
            for (int k = 0; k < F; ++k) f[k] = fl[tt * F + k];
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq) {
              float lf = 0.f;
#pragma unroll
              for (int k = 0; k < F; ++k) lf += f[k] * Ur[qq][k];
              acc += v1r[qq] * tanhf_(kk[qq] + pqb[qq] + lf);
            }
            acc2 = lane < U2 ? v2r * tanhf_(k2 + pq2) : 0.f;
          }
          red[u] = acc; red[RB + u] = acc2;
        }
        wave_sum_multi<2 * RB>(red);
        if (lane < RB) {
          const int i = i0 + lane * AW, tt = c + C * i;
          float r1 = red[0], r2 = red[RB];
#pragma unroll
          for (int u = 1; u < RB; ++u) { r1 = (lane == u) ? red[u] : r1; r2 = (lane == u) ? red[RB + u] : r2; }
          if (i < nown) { gput(wp + WL.x2 + tt, tag, r1); gput(wp + WL.x2 + Ti + tt, tag, r2); }
        }
      }
    }
    // X2: gather e1[0..len), e2[0..len)
    gather_all(wp + WL.x2, len, tag, wave, lane, [&](int i, float v) { e1[i] = v; }, err_word, dead);
    gather_all(wp + WL.x2 + Ti, len, tag, (wave + AW / 2) % AW, lane, [&](int i, float v) { e2[i] = v; }, err_word, dead);
    __syncthreads();
    PROF(5);
    // (6) masked softmax + forward-attention recursion (redundant)
    if (wave == 0) {
      wave_softmax(e1, len, Ti, lane);
      float s = 0.f;
      for (int tt = lane; tt < Ti; tt += 64) {
        const float w = 0.5f * alp[tt] + 0.5f * (tt > 0 ? alp[tt - 1] : 0.f) + 1e-7f;
        const float v = w * e1[tt];
        aln[tt] = v; s += v;
      }
      s = wave_sum(s);
      const float inv = 1.f / s;
      float* o1 = p.align1 + bt * Ti;
      float* oa = p.a1 + bt * Ti;
      for (int tt = lane; tt < Ti; tt += 64) {
        const float v = aln[tt] * inv;
        aln[tt] = v;
        const float a = e1[tt];
        aprev[tt] = a;
        if (c == 0) { o1[tt] = v; oa[tt] = a; }
      }
    } else if (wave == 1) {
      wave_softmax(e2, len, Ti, lane);
      if (c == 0) {
        float* o2 = p.align2 + bt * Ti;
        for (int tt = lane; tt < Ti; tt += 64) o2[tt] = e2[tt];
      }
    }
    __syncthreads();
    PROF(6);
    // (7) contexts (redundant): values from LDS (bf16) in the fast mode, fp32 rows from L2 in the exact mode
    {
      float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (actV) {
        if (KLDS) {
#pragma unroll 4
          for (int tt = wave; tt < len; tt += AW) {
            const float a = aln[tt];
            const uint2 w = *reinterpret_cast<const uint2*>(V1s + tt * V1 + d0);
            c4.x += a * __uint_as_float(w.x << 16); c4.y += a * __uint_as_float(w.x & 0xFFFF0000u);
            c4.z += a * __uint_as_float(w.y << 16); c4.w += a * __uint_as_float(w.y & 0xFFFF0000u);
          }
        } else {
          const float* vb = values1 + d0;
          int tt = wave;
          for (; tt + 3 * AW < len; tt += 4 * AW) {
            const float4 r0 = *reinterpret_cast<const float4*>(vb + (size_t)tt * V1);
            const float4 r1 = *reinterpret_cast<const float4*>(vb + (size_t)(tt + AW) * V1);
            const float4 r2 = *reinterpret_cast<const float4*>(vb + (size_t)(tt + 2 * AW) * V1);
            const float4 r3 = *reinterpret_cast<const float4*>(vb + (size_t)(tt + 3 * AW) * V1);
            const float a0 = aln[tt], a1 = aln[tt + AW], a2 = aln[tt + 2 * AW], a3 = aln[tt + 3 * AW];
            c4.x += a0 * r0.x + a1 * r1.x + a2 * r2.x + a3 * r3.x;
            c4.y += a0 * r0.y + a1 * r1.y + a2 * r2.y + a3 * r3.y;
            c4.z += a0 * r0.z + a1 * r1.z + a2 * r2.z + a3 * r3.z;
            c4.w += a0 * r0.w + a1 * r1.w + a2 * r2.w + a3 * r3.w;
          }
          for (; tt < len; tt += AW) {
            const float a = aln[tt];
            const float4 v = *reinterpret_cast<const float4*>(vb + (size_t)tt * V1);
            c4.x += a * v.x; c4.y += a * v.y; c4.z += a * v.z; c4.w += a * v.w;
          }
        }
        *reinterpret_cast<float4*>(partial + wave * V1 + d0) = c4;
      }
      const int NS2 = ANT / V2, c2 = tid % V2, s2 = tid / V2;
      if (s2 < NS2) {
        float acc = 0.f;
        if (KLDS) { for (int tt = s2; tt < len; tt += NS2) acc += e2[tt] * bf2f(V2s[tt * V2 + c2]); }
        else { for (int tt = s2; tt < len; tt += NS2) acc += e2[tt] * values2[(size_t)tt * V2 + c2]; }
        partial[AW * V1 + s2 * V2 + c2] = acc;
      }
      __syncthreads();
      if (tid < V1) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < AW; ++k) s += partial[k * V1 + tid];
        vec[tid] = s;
        if (c == 0) out[(size_t)t * OW + A + tid] = s;
      } else if (tid < CT) {
        const int cc = tid - V1;
        float s = 0.f;
        for (int k = 0; k < NS2; ++k) s += partial[AW * V1 + k * V2 + cc];
        vec[V1 + cc] = s;
        if (c == 0) out[(size_t)t * OW + A + V1 + cc] = s;
      }
    }
    { float* tmp = alp; alp = aln; aln = tmp; }
    __syncthreads();
    PROF(7);
  }
  PROF_STORE(0);
}

struct SmemCB {
  int dz, dvec, dq, dpq, pqv, dctx, alprev, a, al, a2, dal, da2, de1, dac, dalc, fl, dfl, Fs, dpart, yown, partial, dead,
      kofs, total;
};
__host__ __device__ inline SmemCB carve_cb(int A, int CT, int UQ, int Ti, int F, int KW, int C, int nown, bool klds) {
  auto u = [](int x) { return (x + 3) & ~3; };
  SmemCB s; int o = 0;
  s.dz = o; o += 4 * A; s.dvec = o; o += u(C * nwp_of(CT + A, C)); s.dq = o; o += u(A); s.dpq = o; o += u(UQ);
  s.pqv = o; o += u(UQ); s.dctx = o; o += u(CT);
  const int T4 = u(Ti);
  s.alprev = o; o += T4; s.a = o; o += T4; s.al = o; o += T4; s.a2 = o; o += T4; s.dal = o; o += T4; s.da2 = o; o += T4;
  s.de1 = o; o += T4; s.dac = o; o += T4; s.dalc = o; o += T4;
  s.fl = o; o += u(Ti * F); s.dfl = o; o += u(Ti * F); s.Fs = o; o += u(KW * F);
  s.dpart = o; o += u(C * UQ);
  s.yown = o; o += u(nwp_of(CT + A, C));
  s.partial = o; o += ANT * 8;
  s.dead = o; o += 4;
  s.kofs = o; if (klds) o += u((nown * UQ + 1) / 2);
  s.total = o;
  return s;
}

template <int F, bool KLDS>
__global__ __launch_bounds__(ANT) void attn_cluster_bwd_k(const satt_attn_cluster_bwd_params cb) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const satt_attn_rnn_bwd_params& pb = cb.b;
  const satt_attn_rnn_params& p = pb.f;
  const int C = cb.C;
  const int A = p.A, G = 4 * A, V1 = p.V1, V2 = p.V2, CT = V1 + V2, U1 = p.U1, U2 = p.U2, UQ = U1 + U2;
  const int Ti = p.Ti, Td = p.Td, KW = p.kernel, PL = (KW - 1) / 2;
  const int KR = CT + A, NWP = nwp_of(KR, C);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x, c = blockIdx.y;
  const int nown_max = (Ti + C - 1) / C;
  const SmemCB L = carve_cb(A, CT, UQ, Ti, F, KW, C, nown_max, KLDS);
  float* dz = smem + L.dz;          // [G] full (cell backward is redundant)
  float* dvec = smem + L.dvec;      // [C*NWP] gathered d[ctx|h] in padded column layout
  float* dq = smem + L.dq;
  float* dpq = smem + L.dpq;
  float* pqv = smem + L.pqv;
  float* dctx = smem + L.dctx;
  float* alprev = smem + L.alprev;
  float* a = smem + L.a;
  float* al = smem + L.al;
  float* a2 = smem + L.a2;
  float* dal = smem + L.dal;
  float* da2 = smem + L.da2;
  float* de1 = smem + L.de1;
  float* dac = smem + L.dac;
  float* dalc = smem + L.dalc;
  float* fl = smem + L.fl;
  float* dfl = smem + L.dfl;
  float* Fs = smem + L.Fs;
  float* dpart = smem + L.dpart;    // [C][UQ] gathered d pq partials
  float* partial = smem + L.partial;
  int* dead = reinterpret_cast<int*>(smem + L.dead);
  uint16_t* K1s = reinterpret_cast<uint16_t*>(smem + L.kofs);
  uint16_t* K2s = K1s + nown_max * U1;

  const int len = (int)p.lengths[b];
  const uint32_t seed = p.seed ? *p.seed : 0u;
  const float* keys1 = p.keys1 + (size_t)b * Ti * U1;
  const float* values1 = p.values1 + (size_t)b * Ti * V1;
  const float* keys2 = p.keys2 + (size_t)b * Ti * U2;
  const float* values2 = p.values2 + (size_t)b * Ti * V2;
  const int OW = A + CT;
  const float* dout = pb.dout + (size_t)b * Td * OW;
  const uint16_t* WTslice = cb.WrecTP + (size_t)c * G * NWP;
  const WsLayout WL = ws_layout(A, Ti, C, UQ, F, KR);
  u64* wsb = reinterpret_cast<u64*>(cb.ws);
  unsigned int* err_word = reinterpret_cast<unsigned int*>(wsb + (size_t)2 * p.B * WL.per_parity);
  const int nown = len > c ? (len - c + C - 1) / C : 0;
  // column k of [ctx|h] lives at padded position (k / NWP) * NWP + (k % NWP) == k  (slices are consecutive blocks)

  const int d0 = lane * NQ;
  const bool actU = d0 < U1, actV = d0 < V1;
  float v1r[NQ], b1r[NQ], Ur[NQ][F];
#pragma unroll
  for (int qq = 0; qq < NQ; ++qq) {
    const int d = d0 + qq;
    v1r[qq] = d < U1 ? p.v1[d] : 0.f;
    b1r[qq] = d < U1 ? p.b1[d] : 0.f;
#pragma unroll
    for (int k = 0; k < F; ++k) Ur[qq][k] = d < U1 ? p.locU[k * U1 + d] : 0.f;
  }
  const float v2r = lane < U2 ? p.v2[lane] : 0.f;

  for (int i = tid; i < C * NWP; i += ANT) dvec[i] = 0.f;
  for (int i = tid; i < Ti; i += ANT) { dac[i] = 0.f; dalc[i] = 0.f; dal[i] = 0.f; da2[i] = 0.f; }
  for (int i = tid; i < Ti * F; i += ANT) dfl[i] = 0.f;
  for (int i = tid; i < KW * F; i += ANT) Fs[i] = p.locF[i];
  if (tid == 0) *dead = 0;
  if (KLDS) {
    for (int e = tid; e < nown * U1; e += ANT) { const int i = e / U1, d = e - i * U1; K1s[e] = f2bf(keys1[(size_t)(c + C * i) * U1 + d]); }
    for (int e = tid; e < nown * U2; e += ANT) { const int i = e / U2, d = e - i * U2; K2s[e] = f2bf(keys2[(size_t)(c + C * i) * U2 + d]); }
  }
  float dc_state = 0.f, dh_state = 0.f;
  constexpr int PFL = 2;                                   // fl elements prefetched per thread (PFL*ANT >= Ti*F typically)
  float pf_alprev = 0.f, pf_a = 0.f, pf_al = 0.f, pf_a2 = 0.f, pf_pq = 0.f, pf_fl[PFL];
  auto prefetch = [&](int tn) {                            // issue the loads of step tn (consumed one iteration later)
    const size_t bn = (size_t)b * Td + tn;
    if (tid < Ti) {
      pf_alprev = tn > 0 ? p.align1[(bn - 1) * Ti + tid] : (tid == 0 ? 1.f : 0.f);
      pf_a = p.a1[bn * Ti + tid]; pf_al = p.align1[bn * Ti + tid]; pf_a2 = p.align2[bn * Ti + tid];
    }
#pragma unroll
    for (int u = 0; u < PFL; ++u) { const int e = tid + u * ANT; pf_fl[u] = e < Ti * F ? p.fl[bn * Ti * F + e] : 0.f; }
    if (tid < UQ) pf_pq = p.pq[bn * UQ + tid];
  };
  prefetch(cb.t1 - 1);
  float* stb = cb.state ? cb.state + (size_t)b * (C * NWP + 2 * A + 2 * Ti) : nullptr;
  if (cb.t1 < Td) {        // continue from the chunk that processed steps >= t1
    __syncthreads();
    for (int i = tid; i < C * NWP; i += ANT) dvec[i] = stb[i];
    if (tid < A) { dc_state = stb[C * NWP + tid]; dh_state = stb[C * NWP + A + tid]; }
    for (int i = tid; i < Ti; i += ANT) { dac[i] = stb[C * NWP + 2 * A + i]; dalc[i] = stb[C * NWP + 2 * A + Ti + i]; }
  }
  __syncthreads();

  PROF_DECL;
  for (int t = cb.t1 - 1; t >= cb.t0; --t) {
    PROF(0);
    const size_t bt = (size_t)b * Td + t;
    const uint32_t tag = (uint32_t)(t + 1);
    u64* wp = wsb + ((size_t)(t & 1) * p.B + b) * WL.per_parity;
    // (a) forward state of this step: prefetched into registers one step ahead (see the end of the loop body)
    if (tid < Ti) { alprev[tid] = pf_alprev; a[tid] = pf_a; al[tid] = pf_al; a2[tid] = pf_a2; }
    for (int i = tid + ANT; i < Ti; i += ANT) {        // Ti > ANT only
      alprev[i] = t > 0 ? p.align1[(bt - 1) * Ti + i] : (i == 0 ? 1.f : 0.f);
      a[i] = p.a1[bt * Ti + i]; al[i] = p.align1[bt * Ti + i]; a2[i] = p.align2[bt * Ti + i];
    }
#pragma unroll
    for (int u = 0; u < PFL; ++u) { const int e = tid + u * ANT; if (e < Ti * F) fl[e] = pf_fl[u]; }
    for (int e = tid + PFL * ANT; e < Ti * F; e += ANT) fl[e] = p.fl[bt * Ti * F + e];
    if (tid < UQ) pqv[tid] = pf_pq;
    if (tid < CT) {
      const float g = dout[(size_t)t * OW + A + tid] + dvec[tid];
      dctx[tid] = g;
      if (c == 0) pb.dctx[bt * CT + tid] = g;
    }
    if (t > cb.t0) prefetch(t - 1);                        // loads fly while the rest of this step executes
    __syncthreads();
    PROF(1);
    // (b) d alpha / d a2 for own rows, publish
    {
      float dcr[NQ];
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) dcr[qq] = (d0 + qq) < V1 ? dctx[d0 + qq] : 0.f;
      const float dc2 = lane < V2 ? dctx[V1 + lane] : 0.f;
      const float* vb = values1 + d0;
      for (int i0 = wave; i0 < nown; i0 += 4 * AW) {
        float4 r[4]; float w2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * AW, tt = c + C * i;
          r[u] = make_float4(0.f, 0.f, 0.f, 0.f); w2[u] = 0.f;
          if (i < nown) {
            if (actV) r[u] = *reinterpret_cast<const float4*>(vb + (size_t)tt * V1);
            if (lane < V2) w2[u] = values2[(size_t)tt * V2 + lane];
          }
        }
        float red8[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          red8[u] = r[u].x * dcr[0] + r[u].y * dcr[1] + r[u].z * dcr[2] + r[u].w * dcr[3];
          red8[4 + u] = w2[u] * dc2;
        }
        wave_sum_multi<8>(red8);
        if (lane < 4) {
          const int i = i0 + lane * AW, tt = c + C * i;
          float s1 = red8[0], s2 = red8[4];
#pragma unroll
          for (int u = 1; u < 4; ++u) { s1 = (lane == u) ? red8[u] : s1; s2 = (lane == u) ? red8[4 + u] : s2; }
          if (i < nown) { gput(wp + WL.xb + tt, tag, s1); gput(wp + WL.xb + Ti + tt, tag, s2); }
        }
      }
    }
    gather_all(wp + WL.xb, len, tag, wave, lane, [&](int i, float v) {
      dal[i] = v + dalc[i] + (pb.dalign1 ? pb.dalign1[bt * Ti + i] : 0.f); }, err_word, dead);
    gather_all(wp + WL.xb + Ti, len, tag, (wave + AW / 2) % AW, lane, [&](int i, float v) {
      da2[i] = v + (pb.dalign2 ? pb.dalign2[bt * Ti + i] : 0.f); }, err_word, dead);
    // rows >= len: d alpha = carry + external only (their context contribution is zero)
    for (int i = len + tid; i < Ti; i += ANT) {
      dal[i] = dalc[i] + (pb.dalign1 ? pb.dalign1[bt * Ti + i] : 0.f);
      da2[i] = (pb.dalign2 ? pb.dalign2[bt * Ti + i] : 0.f);
    }
    __syncthreads();
    PROF(2);
    // (c) forward-attention recursion + softmax backward (redundant)
    if (wave == 0) {
      float S = 0.f, s1 = 0.f;
      for (int tt = lane; tt < Ti; tt += 64) {
        const float w = 0.5f * alprev[tt] + 0.5f * (tt > 0 ? alprev[tt - 1] : 0.f) + 1e-7f;
        S += w * a[tt];
        s1 += dal[tt] * al[tt];
      }
      S = wave_sum(S); s1 = wave_sum(s1);
      const float invS = 1.f / S;
      float s2 = 0.f;
      for (int tt = lane; tt < Ti; tt += 64) {
        const float w = 0.5f * alprev[tt] + 0.5f * (tt > 0 ? alprev[tt - 1] : 0.f) + 1e-7f;
        const float dalp = (dal[tt] - s1) * invS;
        const float da = dalp * w + dac[tt];
        dal[tt] = dalp * a[tt];
        de1[tt] = da;
        s2 += da * a[tt];
      }
      s2 = wave_sum(s2);
      float* g1 = pb.de1 + bt * Ti;
      for (int tt = lane; tt < Ti; tt += 64) { const float v = a[tt] * (de1[tt] - s2); de1[tt] = v; if (c == 0) g1[tt] = v; }
    } else if (wave == 1) {
      float s3 = 0.f;
      for (int tt = lane; tt < Ti; tt += 64) s3 += da2[tt] * a2[tt];
      s3 = wave_sum(s3);
      float* g2 = pb.de2 + bt * Ti;
      for (int tt = lane; tt < Ti; tt += 64) { const float v = a2[tt] * (da2[tt] - s3); da2[tt] = v; if (c == 0) g2[tt] = v; }
    }
    __syncthreads();
    PROF(3);
    for (int i = tid; i < Ti; i += ANT) dalc[i] = 0.5f * dal[i] + 0.5f * (i + 1 < Ti ? dal[i + 1] : 0.f);
    // (d) energy backward for own rows: partial d pq, d location-features of own rows; publish both
    {
      float pqb[NQ], dpqa[NQ];
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) {
        pqb[qq] = (d0 + qq) < U1 ? pqv[d0 + qq] + b1r[qq] : 0.f;
        dpqa[qq] = 0.f;
      }
      const float pq2 = lane < U2 ? pqv[U1 + lane] : 0.f;
      float dpq2a = 0.f;
      float* dflg = pb.dfl + bt * Ti * F;
      for (int i0 = wave; i0 < nown; i0 += RB * AW) {
        float dfp[RB * F];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const int i = i0 + u * AW, tt = c + C * i;
#pragma unroll
          for (int k = 0; k < F; ++k) dfp[u * F + k] = 0.f;
          if (i < nown) {
            const float de = de1[tt];
            float f[F];
#pragma unroll
            for (int k = 0; k < F; ++k) f[k] = fl[tt * F + k];
            float kk[NQ];
            load_key4<KLDS>(keys1 + (KLDS ? 0 : (size_t)tt * U1), K1s, KLDS ? i : 0, U1, d0, actU, kk);
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq) {
              float lf = 0.f;
#pragma unroll
              for (int k = 0; k < F; ++k) lf += f[k] * Ur[qq][k];
              const float th = tanhf_(kk[qq] + pqb[qq] + lf);
              const float g = de * v1r[qq] * (1.f - th * th);
              dpqa[qq] += g;
#pragma unroll
              for (int k = 0; k < F; ++k) dfp[u * F + k] += g * Ur[qq][k];
            }
            if (lane < U2) {
              const float th2 = tanhf_(load_key1<KLDS>(keys2 + (KLDS ? 0 : (size_t)tt * U2), K2s, KLDS ? i : 0, U2, lane, true) + pq2);
              dpq2a += da2[tt] * v2r * (1.f - th2 * th2);
            }
          }
        }
        wave_sum_multi<RB * F>(dfp);
        if (lane < RB * F) {
          const int u = lane / F, k = lane - u * F, i = i0 + u * AW, tt = c + C * i;
          float v = dfp[0];
#pragma unroll
          for (int q2 = 1; q2 < RB * F; ++q2) v = (lane == q2) ? dfp[q2] : v;
          if (i < nown) { gput(wp + WL.xd + C * UQ + tt * F + k, tag, v); dflg[tt * F + k] = v; }
        }
      }
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) { const int d = d0 + qq; if (d < U1) partial[wave * UQ + d] = dpqa[qq]; }
      if (lane < U2) partial[wave * UQ + U1 + lane] = dpq2a;
    }
    __syncthreads();
    if (tid < UQ) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < AW; ++w) s += partial[w * UQ + tid];
      gput(wp + WL.xd + c * UQ + tid, tag, s);
    }
    if (c == 0) { float* dflg = pb.dfl + bt * Ti * F; for (int e = tid + len * F; e < Ti * F; e += ANT) dflg[e] = 0.f; }
    // Xd: all C partial d pq vectors + the d fl rows of every member
    gather_all(wp + WL.xd, C * UQ, tag, wave, lane, [&](int i, float v) { dpart[i] = v; }, err_word, dead);
    gather_all(wp + WL.xd + C * UQ, len * F, tag, (wave + AW / 2) % AW, lane, [&](int i, float v) { dfl[i] = v; },
               err_word, dead);
    __syncthreads();
    PROF(4);
    if (tid < UQ) {
      float s = 0.f;
      for (int k = 0; k < C; ++k) s += dpart[k * UQ + tid];     // fixed order: identical in every member
      dpq[tid] = s;
      if (c == 0) pb.dpq[bt * UQ + tid] = s;
    }
    // (e) location conv backward (redundant): carry for a_{t-1}
    for (int s = tid; s < Ti; s += ANT) {
      float g = 0.f;
      for (int jj = 0; jj < KW; ++jj) {
        const int tt = s - jj + PL;
        if (tt >= 0 && tt < len) {
#pragma unroll
          for (int k = 0; k < F; ++k) g += dfl[tt * F + k] * Fs[jj * F + k];
        }
      }
      dac[s] = g;
    }
    __syncthreads();
    PROF(5);
    // (f) d query = dpq x Wq^T (redundant)
    matvec_bf16<ANT, MVU>(dpq, pb.WqT, UQ, A, partial, dq);
    PROF(6);
    // (g) LSTM cell backward (redundant, all A units)
    float dh_direct = 0.f;
    if (tid < A) {
      const int j = tid;
      const uint32_t idx = (uint32_t)bt * (uint32_t)A + (uint32_t)j;
      float kc, kh, pc, ph;
      if (p.training) {
        kc = (p.zc_thresh == 0 || satt_keep(seed, p.stream_c, idx, p.zc_thresh)) ? 1.f : 0.f; pc = 1.f - kc;
        kh = (p.zh_thresh == 0 || satt_keep(seed, p.stream_h, idx, p.zh_thresh)) ? 1.f : 0.f; ph = 1.f - kh;
      } else {
        kc = 1.f - p.zc; pc = p.zc; kh = 1.f - p.zh; ph = p.zh;
      }
      const float* gr = p.gates + bt * G;
      const float gi = gr[j], gj = gr[A + j], gf = gr[2 * A + j], go = gr[3 * A + j];
      const float cn = p.cnew[bt * A + j];
      const float cp = t > 0 ? p.cstate[(bt - 1) * A + j] : 0.f;
      const float dhn = dout[(size_t)t * OW + j] + dq[j] + kh * dh_state;
      dh_direct = ph * dh_state;
      const float tc = tanhf_(cn);
      const float dcn = dhn * go * (1.f - tc * tc) + kc * dc_state;
      const float d_o = dhn * tc;
      const float dzi = dcn * gj * gi * (1.f - gi);
      const float dzj = dcn * gi * (1.f - gj * gj);
      const float dzf = dcn * cp * gf * (1.f - gf);
      const float dzo = d_o * go * (1.f - go);
      dc_state = dcn * gf + pc * dc_state;
      if (c == 0) {
        float* dr = pb.dxg + bt * G;
        dr[j] = dzi; dr[A + j] = dzj; dr[2 * A + j] = dzf; dr[3 * A + j] = dzo;
      }
      dz[j] = dzi; dz[A + j] = dzj; dz[2 * A + j] = dzf; dz[3 * A + j] = dzo;
    }
    __syncthreads();
    PROF(7);
    // (h) own output columns of dz x Wrec^T, publish, gather all
    if (t > 0) {
      float* yown = smem + L.yown;                                // [NWP] own output columns
      matvec_bf16<ANT, MVU>(dz, WTslice, G, NWP, partial, yown);
      if (tid < NWP) gput(wp + WL.xh + c * NWP + tid, tag, yown[tid]);
      gather_all(wp + WL.xh, C * NWP, tag, wave, lane, [&](int i, float v) { dvec[i] = v; }, err_word, dead);
      __syncthreads();
      if (tid < A) dh_state = dvec[CT + tid] + dh_direct;
      __syncthreads();
    }
    PROF(8);
  }
  if (cb.t0 > 0 && c == 0) {   // hand the carried gradients to the next (earlier) chunk
    for (int i = tid; i < C * NWP; i += ANT) stb[i] = dvec[i];
    if (tid < A) { stb[C * NWP + tid] = dc_state; stb[C * NWP + A + tid] = dh_state; }
    for (int i = tid; i < Ti; i += ANT) { stb[C * NWP + 2 * A + i] = dac[i]; stb[C * NWP + 2 * A + Ti + i] = dalc[i]; }
  }
  PROF_STORE(16);
}

__global__ void attn_cluster_pack_k(const float* __restrict__ W, int64_t ld, uint16_t* __restrict__ WP,
                                    uint16_t* __restrict__ WTP, int K, int A, int C) {
  const int G = 4 * A, AU = A / C, NL = 4 * AU, NWP = nwp_of(K, C);
  const int64_t n1 = (int64_t)C * K * NL, n2 = (int64_t)C * G * NWP;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n1 + n2; e += (int64_t)gridDim.x * blockDim.x) {
    if (e < n1) {
      const int lc = (int)(e % NL); const int64_t r = e / NL; const int k = (int)(r % K), c = (int)(r / K);
      const int g = lc / AU, u = lc - g * AU;
      WP[e] = f2bf(W[(int64_t)k * ld + g * A + c * AU + u]);
    } else {
      const int64_t f = e - n1;
      const int j = (int)(f % NWP); const int64_t r = f / NWP; const int row = (int)(r % G), c = (int)(r / G);
      const int col = c * NWP + j;
      WTP[f] = col < K ? f2bf(W[(int64_t)col * ld + row]) : (uint16_t)0;
    }
  }
}

inline int ccheck(const satt_attn_rnn_params& p, int C) {
  if (p.B <= 0 || p.Td <= 0 || p.Ti <= 0 || C < 2) return SATT_E_BADARG;
  if (p.filters != 5) return SATT_E_UNSUPPORTED;
  if (p.U1 > 64 * NQ || p.V1 > 64 * NQ || p.U2 > 64 || p.V2 > 64 || p.U1 % 4 || p.V1 % 4) return SATT_E_UNSUPPORTED;
  if ((p.U1 + p.U2) % 8 || (p.V1 + p.V2 + p.A) % 8 || p.A % 8) return SATT_E_UNSUPPORTED;
  if (p.A % C || (p.A / C) % 8 || p.A > ANT || 4 * p.A > 8 * ANT) return SATT_E_UNSUPPORTED;
  if (p.U1 + p.U2 > ANT || p.V1 + p.V2 > ANT) return SATT_E_UNSUPPORTED;
  if (nwp_of(p.V1 + p.V2 + p.A, C) > ANT) return SATT_E_UNSUPPORTED;
  if (p.B * C > 256) return SATT_E_UNSUPPORTED;    // every member must be resident (one workgroup per CU)
  return SATT_OK;
}

}  // namespace

extern "C" int64_t satt_attn_cluster_ws_bytes(const satt_attn_rnn_params* f, int C) {
  if (!f) return 0;
  const WsLayout w = ws_layout(f->A, f->Ti, C, f->U1 + f->U2, 5, f->V1 + f->V2 + f->A);
  return (int64_t)sizeof(u64) * 2 * f->B * w.per_parity + 64;
}
extern "C" int64_t satt_attn_cluster_state_floats(const satt_attn_rnn_params* f, int C) {
  if (!f) return 0;
  return (int64_t)f->B * (C * nwp_of(f->V1 + f->V2 + f->A, C) + 2 * f->A + 2 * f->Ti);
}
extern "C" int64_t satt_attn_cluster_pack_elems(int K, int A, int C, int transposed) {
  return transposed ? (int64_t)C * 4 * A * nwp_of(K, C) : (int64_t)C * K * 4 * (A / C);
}
extern "C" int satt_attn_cluster_pack(const float* Wrec, int64_t ld, uint16_t* WrecP, uint16_t* WrecTP, int K, int A,
                                      int C, void* stream) {
  if (C < 1 || A % C || K <= 0) return SATT_E_BADARG;
  hipLaunchKernelGGL(attn_cluster_pack_k, dim3(1024), dim3(256), 0, (hipStream_t)stream, Wrec, ld, WrecP, WrecTP, K, A, C);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

extern "C" int satt_attn_cluster_fwd(const satt_attn_cluster_params* cp, void* stream) {
  if (!cp) return SATT_E_BADARG;
  const satt_attn_rnn_params& p = cp->f;
  int rc = ccheck(p, cp->C);
  if (rc) return rc;
  if (cp->t0 < 0 || cp->t1 > p.Td || cp->t0 >= cp->t1) return SATT_E_BADARG;
  const int C = cp->C, CT = p.V1 + p.V2, UQ = p.U1 + p.U2, NL = 4 * (p.A / C), nown = (p.Ti + C - 1) / C;
  const bool klds = p.keys_lds_bf16 != 0;
  const size_t smem = sizeof(float) * carve_cf(p.A, CT, UQ, p.Ti, 5, p.kernel, NL, nown, klds).total;
  if (smem > 160 * 1024) return SATT_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(cp->ws, 0, (size_t)satt_attn_cluster_ws_bytes(&p, C), s) != hipSuccess) return SATT_E_LAUNCH;
  if (klds) {
    (void)hipFuncSetAttribute((const void*)attn_cluster_fwd_k<5, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((attn_cluster_fwd_k<5, true>), dim3(p.B, C), dim3(ANT), smem, s, *cp);
  } else {
    (void)hipFuncSetAttribute((const void*)attn_cluster_fwd_k<5, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((attn_cluster_fwd_k<5, false>), dim3(p.B, C), dim3(ANT), smem, s, *cp);
  }
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

extern "C" int satt_attn_cluster_bwd(const satt_attn_cluster_bwd_params* cb, void* stream) {
  if (!cb) return SATT_E_BADARG;
  const satt_attn_rnn_params& p = cb->b.f;
  int rc = ccheck(p, cb->C);
  if (rc) return rc;
  if (cb->t0 < 0 || cb->t1 > p.Td || cb->t0 >= cb->t1 || ((cb->t0 > 0 || cb->t1 < p.Td) && !cb->state)) return SATT_E_BADARG;
  const int C = cb->C, CT = p.V1 + p.V2, UQ = p.U1 + p.U2, nown = (p.Ti + C - 1) / C;
  const bool klds = p.keys_lds_bf16 != 0;
  const size_t smem = sizeof(float) * carve_cb(p.A, CT, UQ, p.Ti, 5, p.kernel, C, nown, klds).total;
  if (smem > 160 * 1024) return SATT_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(cb->ws, 0, (size_t)satt_attn_cluster_ws_bytes(&p, C), s) != hipSuccess) return SATT_E_LAUNCH;
  if (klds) {
    (void)hipFuncSetAttribute((const void*)attn_cluster_bwd_k<5, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((attn_cluster_bwd_k<5, true>), dim3(p.B, C), dim3(ANT), smem, s, *cb);
  } else {
    (void)hipFuncSetAttribute((const void*)attn_cluster_bwd_k<5, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((attn_cluster_bwd_k<5, false>), dim3(p.B, C), dim3(ANT), smem, s, *cb);
  }
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

/* host-synchronous (tests / debugging): non-zero if a hand-off of the last launch on `ws` timed out */
extern "C" int satt_attn_cluster_status(const satt_attn_rnn_params* f, int C, const void* ws, void* stream) {
  unsigned int v = 0;
  const char* pz = (const char*)ws + satt_attn_cluster_ws_bytes(f, C) - 64;
  if (hipMemcpyAsync(&v, pz, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return SATT_E_LAUNCH;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return SATT_E_LAUNCH;
  return v ? SATT_E_LAUNCH : SATT_OK;
}

#ifdef SATT_PROFILE
extern "C" int satt_prof_read_cluster(unsigned long long* host32) {
  return hipMemcpyFromSymbol(host32, HIP_SYMBOL(satt_prof_acc), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -3;
}
#endif
