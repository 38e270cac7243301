// Dual-source attention RNN loop (AttentionWrapper[ZoneoutLSTM + ForwardAttention + BahdanauAttention]) as ONE
// persistent 512-thread workgroup per sample walking all Td teacher-forced steps; samples are independent so no
// inter-workgroup communication is needed.
//  * recurrent weights: bf16, streamed from L2 with register double buffering (matvec.h)
//  * keys (constant over all steps): staged ONCE per launch in LDS as bf16 (KLDS) — or read as fp32 from global
//    every step in the exact parity mode
//  * values rows: one wave per memory row, 16 B per lane, 4 rows in flight
//  * attention state, energies and alignments: fp32 in LDS
//  * backward: only the RECURRENT gradient flow runs in the serial loop; gradients that are plain sums over
//    steps (dkeys, dv, dU, db, dF ...) are produced afterwards by attn_param_grads_k / batched GEMMs
// Follows reference modules/forward_attention.py:88-122 (ForwardAttention.__call__), :13-26 (score),
// :128-136 (initial state), TF BahdanauAttention (modules/attentions.py:53-57) and SURVEY.md A.7-A.9.
#include "attn_common.h"

namespace {

struct SmemF {   // forward LDS carve (in floats); bf16 keys follow at kofs
  int vec, z, q, pq, aprev, alA, alB, e1, e2, fl, Fs, bFs, partial, kofs, total;
};
__host__ __device__ inline SmemF carve_fwd(int A, int CT, int UQ, int Ti, int F, int KW, int U1, int U2, bool klds) {
  auto u = [](int x) { return (x + 3) & ~3; };
  SmemF s; int o = 0;
  s.vec = o; o += u(CT + A); s.z = o; o += 4 * A; s.q = o; o += u(A); s.pq = o; o += u(UQ);
  s.aprev = o; o += u(Ti); s.alA = o; o += u(Ti); s.alB = o; o += u(Ti); s.e1 = o; o += u(Ti); s.e2 = o; o += u(Ti);
  s.fl = o; o += u(Ti * F); s.Fs = o; o += u(KW * F); s.bFs = o; o += u(F);
  s.partial = o; o += ANT * 8;
  s.kofs = o; if (klds) o += u((Ti * (U1 + U2) + 1) / 2);
  s.total = o;
  return s;
}

template <int F, bool KLDS>
__global__ __launch_bounds__(ANT) void attn_rnn_fwd_k(const satt_attn_rnn_params p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int A = p.A, G = 4 * A, V1 = p.V1, V2 = p.V2, CT = V1 + V2, U1 = p.U1, U2 = p.U2, UQ = U1 + U2;
  const int Ti = p.Ti, Td = p.Td, KW = p.kernel, PL = (KW - 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const SmemF L = carve_fwd(A, CT, UQ, Ti, F, KW, U1, U2, KLDS);
  float* vec = smem + L.vec;        // [CT + A]  ctx1 | ctx2 | h_state
  float* z = smem + L.z;            // [G]
  float* q = smem + L.q;            // [A]   query = h' (pre-zoneout)
  float* pq = smem + L.pq;          // [UQ]
  float* aprev = smem + L.aprev;    // [Ti]  previous softmax probs (location conv input)
  float* alA = smem + L.alA;        // [Ti]  alpha ping
  float* alB = smem + L.alB;        // [Ti]  alpha pong
  float* e1 = smem + L.e1;          // [Ti]
  float* e2 = smem + L.e2;          // [Ti]
  float* fl = smem + L.fl;          // [Ti*F]
  float* Fs = smem + L.Fs;          // [KW*F]
  float* bFs = smem + L.bFs;        // [F]
  float* partial = smem + L.partial;  // [ANT*8]
  uint16_t* K1s = reinterpret_cast<uint16_t*>(smem + L.kofs);   // bf16 [Ti*U1]
  uint16_t* K2s = K1s + Ti * U1;                                 // bf16 [Ti*U2]

  const int len = (int)p.lengths[b];
  const uint32_t seed = p.seed ? *p.seed : 0u;
  const float* xg = p.xg + (size_t)b * Td * G;
  const float* keys1 = p.keys1 + (size_t)b * Ti * U1;
  const float* values1 = p.values1 + (size_t)b * Ti * V1;
  const float* keys2 = p.keys2 + (size_t)b * Ti * U2;
  const float* values2 = p.values2 + (size_t)b * Ti * V2;
  const int OW = A + CT;
  float* out = p.out + (size_t)b * Td * OW;

  // per-lane constants of the energy pass: lane owns units d0..d0+3
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
  for (int i = tid; i < Ti; i += ANT) { aprev[i] = 0.f; alA[i] = (i == 0) ? 1.f : 0.f; }
  for (int i = tid; i < KW * F; i += ANT) Fs[i] = p.locF[i];
  if (tid < F) bFs[tid] = p.locFb[tid];
  if (KLDS) {
    for (int i = tid; i < len * U1; i += ANT) K1s[i] = f2bf(keys1[i]);
    for (int i = tid; i < len * U2; i += ANT) K2s[i] = f2bf(keys2[i]);
  }
  float c = 0.f, h = 0.f;
  float* alp = alA;  // alpha_{t-1}
  float* aln = alB;  // alpha_t
  __syncthreads();

  PROF_DECL;
  for (int t = 0; t < Td; ++t) {
    PROF(0);
    const size_t bt = (size_t)b * Td + t;
    float xi = 0.f, xj = 0.f, xf = 0.f, xo = 0.f;
    if (tid < A) {
      const float* xr = xg + (size_t)t * G;
      xi = xr[tid]; xj = xr[A + tid]; xf = xr[2 * A + tid]; xo = xr[3 * A + tid];
    }
    // (1) recurrent gate pre-activations: [ctx_{t-1} | h_{t-1}] x Wrec
    matvec_bf16<ANT, MVU>(vec, p.Wrec, CT + A, G, partial, z);
    PROF(1);
    // (2) LSTM cell + zoneout
    if (tid < A) {
      const int j = tid;
      const float gi = sigmoidf_(xi + z[j]);
      const float gj = tanhf_(xj + z[A + j]);
      const float gf = sigmoidf_(xf + z[2 * A + j] + 1.0f);
      const float go = sigmoidf_(xo + z[3 * A + j]);
      const float cn = gf * c + gi * gj;
      const float hn = go * tanhf_(cn);
      float* gr = p.gates + bt * G;
      gr[j] = gi; gr[A + j] = gj; gr[2 * A + j] = gf; gr[3 * A + j] = go;
      p.cnew[bt * A + j] = cn;
      const uint32_t idx = (uint32_t)bt * (uint32_t)A + (uint32_t)j;
      if (p.training) {
        if (p.zc_thresh == 0 || satt_keep(seed, p.stream_c, idx, p.zc_thresh)) c = cn;
        if (p.zh_thresh == 0 || satt_keep(seed, p.stream_h, idx, p.zh_thresh)) h = hn;
      } else {
        c = (1.f - p.zc) * cn + p.zc * c;
        h = (1.f - p.zh) * hn + p.zh * h;
      }
      p.cstate[bt * A + j] = c;
      p.hstate[bt * A + j] = h;
      vec[CT + j] = h;
      q[j] = hn;
      out[(size_t)t * OW + j] = hn;
    }
    __syncthreads();
    PROF(2);
    // (3) processed queries for both mechanisms
    matvec_bf16<ANT, MVU>(q, p.Wq, A, UQ, partial, pq);
    PROF(3);
    if (tid < UQ) p.pq[bt * UQ + tid] = pq[tid];
    // (4) location features f = conv1d_SAME(a_{t-1}) + bias   (saved for the backward pass)
    {
      float* flg = p.fl + bt * Ti * F;
      for (int e = tid; e < Ti * F; e += ANT) {
        const int tt = e / F, k = e - tt * F;
        float s = bFs[k];
        for (int jj = 0; jj < KW; ++jj) {
          const int src = tt + jj - PL;
          if (src >= 0 && src < Ti) s += aprev[src] * Fs[jj * F + k];
        }
        fl[e] = s; flg[e] = s;
      }
    }
    __syncthreads();
    PROF(4);
    // (5) energies: one wave per memory row
    {
      float pqb[NQ];
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) pqb[qq] = (d0 + qq) < U1 ? pq[d0 + qq] + b1r[qq] : 0.f;
      const float pq2 = lane < U2 ? pq[U1 + lane] : 0.f;
      for (int t0 = wave; t0 < len; t0 += RB * AW) {
        float red[2 * RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const int tt = t0 + u * AW;
          float acc = 0.f, acc2 = 0.f;
          if (tt < len) {
            float kk[NQ];
            load_key4<KLDS>(keys1, K1s, tt, U1, d0, actU, kk);
            const float k2 = load_key1<KLDS>(keys2, K2s, tt, U2, lane, lane < U2);
            float f[F];
#pragma unroll
            for (int k = 0; k < F; ++k) f[k] = fl[tt * F + k];
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq) {
              float lf = 0.f;
#pragma unroll
              for (int k = 0; k < F; ++k) lf += f[k] * Ur[qq][k];
              acc += v1r[qq] * tanhf_(kk[qq] + pqb[qq] + lf);     // v1r == 0 for inactive lanes
            }
            acc2 = lane < U2 ? v2r * tanhf_(k2 + pq2) : 0.f;
          }
          red[u] = acc; red[RB + u] = acc2;
        }
        wave_sum_multi<2 * RB>(red);
        if (lane < RB) {
          const int tt = t0 + lane * AW;
          float r1 = red[0], r2 = red[RB];
#pragma unroll
          for (int u = 1; u < RB; ++u) { r1 = (lane == u) ? red[u] : r1; r2 = (lane == u) ? red[RB + u] : r2; }
          if (tt < len) { e1[tt] = r1; e2[tt] = r2; }
        }
      }
    }
    __syncthreads();
    PROF(5);
    // (6) masked softmax (+ forward-attention recursion for mechanism 1)
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
        aln[tt] = v; o1[tt] = v;
        const float a = e1[tt];
        oa[tt] = a; aprev[tt] = a;
      }
    } else if (wave == 1) {
      wave_softmax(e2, len, Ti, lane);
      float* o2 = p.align2 + bt * Ti;
      for (int tt = lane; tt < Ti; tt += 64) o2[tt] = e2[tt];
    }
    __syncthreads();
    PROF(6);
    // (7) contexts: one wave per memory row (16 B per lane, 4 rows in flight), cross-wave reduction through LDS
    {
      float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (actV) {
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
        *reinterpret_cast<float4*>(partial + wave * V1 + d0) = c4;
      }
      const int V2d = V2 > 0 ? V2 : 1;                 // V2 == 0: no second context (single-source decoder)
      const int NS2 = V2 > 0 ? ANT / V2 : 0, c2 = tid % V2d, s2 = tid / V2d;
      if (s2 < NS2) {
        float acc = 0.f;
        for (int tt = s2; tt < len; tt += NS2) acc += e2[tt] * values2[(size_t)tt * V2 + c2];
        partial[AW * V1 + s2 * V2 + c2] = acc;
      }
      __syncthreads();
      if (tid < V1) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < AW; ++k) s += partial[k * V1 + tid];
        vec[tid] = s; out[(size_t)t * OW + A + tid] = s;
      } else if (tid < CT) {
        const int cc = tid - V1;
        float s = 0.f;
        for (int k = 0; k < NS2; ++k) s += partial[AW * V1 + k * V2 + cc];
        vec[V1 + cc] = s; out[(size_t)t * OW + A + V1 + cc] = s;
      }
    }
    { float* tmp = alp; alp = aln; aln = tmp; }
    __syncthreads();
    PROF(7);
  }
  PROF_STORE(0);
}

struct SmemB {
  int dz, dvec, dq, dpq, pqv, dctx, alprev, a, al, a2, dal, da2, de1, dac, dalc, fl, dfl, Fs, partial, kofs, total;
};
__host__ __device__ inline SmemB carve_bwd(int A, int CT, int UQ, int Ti, int F, int KW, int U1, int U2, bool klds) {
  auto u = [](int x) { return (x + 3) & ~3; };
  SmemB s; int o = 0;
  s.dz = o; o += 4 * A; s.dvec = o; o += u(CT + A); s.dq = o; o += u(A); s.dpq = o; o += u(UQ); s.pqv = o; o += u(UQ);
  s.dctx = o; o += u(CT);
  const int T4 = u(Ti);
  s.alprev = o; o += T4; s.a = o; o += T4; s.al = o; o += T4; s.a2 = o; o += T4; s.dal = o; o += T4; s.da2 = o; o += T4;
  s.de1 = o; o += T4; s.dac = o; o += T4; s.dalc = o; o += T4;
  s.fl = o; o += u(Ti * F); s.dfl = o; o += u(Ti * F); s.Fs = o; o += u(KW * F);
  s.partial = o; o += ANT * 8;
  s.kofs = o; if (klds) o += u((Ti * (U1 + U2) + 1) / 2);
  s.total = o;
  return s;
}

template <int F, bool KLDS>
__global__ __launch_bounds__(ANT) void attn_rnn_bwd_k(const satt_attn_rnn_bwd_params pb) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const satt_attn_rnn_params& p = pb.f;
  const int A = p.A, G = 4 * A, V1 = p.V1, V2 = p.V2, CT = V1 + V2, U1 = p.U1, U2 = p.U2, UQ = U1 + U2;
  const int Ti = p.Ti, Td = p.Td, KW = p.kernel, PL = (KW - 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const SmemB L = carve_bwd(A, CT, UQ, Ti, F, KW, U1, U2, KLDS);
  float* dz = smem + L.dz;          // [G]
  float* dvec = smem + L.dvec;      // [CT + A]  grad wrt [ctx_{t-1} | hstate_{t-1}]
  float* dq = smem + L.dq;          // [A]
  float* dpq = smem + L.dpq;        // [UQ]
  float* pqv = smem + L.pqv;        // [UQ]
  float* dctx = smem + L.dctx;      // [CT]
  float* alprev = smem + L.alprev;  // alpha_{t-1}
  float* a = smem + L.a;            // a_t
  float* al = smem + L.al;          // alpha_t
  float* a2 = smem + L.a2;          // a2_t
  float* dal = smem + L.dal;        // d alpha_t  -> reused as dw
  float* da2 = smem + L.da2;        // d a2_t  -> d e2
  float* de1 = smem + L.de1;        // d e1
  float* dac = smem + L.dac;        // carry: grad wrt a_t from step t+1's location conv
  float* dalc = smem + L.dalc;      // carry: grad wrt alpha_t from step t+1's recursion
  float* fl = smem + L.fl;          // [Ti*F]
  float* dfl = smem + L.dfl;        // [Ti*F]
  float* Fs = smem + L.Fs;          // [KW*F]
  float* partial = smem + L.partial;  // [ANT*8]
  uint16_t* K1s = reinterpret_cast<uint16_t*>(smem + L.kofs);
  uint16_t* K2s = K1s + Ti * U1;

  const int len = (int)p.lengths[b];
  const uint32_t seed = p.seed ? *p.seed : 0u;
  const float* keys1 = p.keys1 + (size_t)b * Ti * U1;
  const float* values1 = p.values1 + (size_t)b * Ti * V1;
  const float* keys2 = p.keys2 + (size_t)b * Ti * U2;
  const float* values2 = p.values2 + (size_t)b * Ti * V2;
  const int OW = A + CT;
  const float* dout = pb.dout + (size_t)b * Td * OW;

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

  for (int i = tid; i < CT + A; i += ANT) dvec[i] = 0.f;
  for (int i = tid; i < Ti; i += ANT) { dac[i] = 0.f; dalc[i] = 0.f; }
  for (int i = tid; i < KW * F; i += ANT) Fs[i] = p.locF[i];
  if (KLDS) {
    for (int i = tid; i < len * U1; i += ANT) K1s[i] = f2bf(keys1[i]);
    for (int i = tid; i < len * U2; i += ANT) K2s[i] = f2bf(keys2[i]);
  }
  float dc_state = 0.f, dh_state = 0.f;
  __syncthreads();

  PROF_DECL;
  for (int t = Td - 1; t >= 0; --t) {
    PROF(0);
    const size_t bt = (size_t)b * Td + t;
    // (a) load forward state of this step, total context gradient
    for (int i = tid; i < Ti; i += ANT) {
      alprev[i] = t > 0 ? p.align1[(bt - 1) * Ti + i] : (i == 0 ? 1.f : 0.f);
      a[i] = p.a1[bt * Ti + i];
      al[i] = p.align1[bt * Ti + i];
      a2[i] = p.align2[bt * Ti + i];
    }
    {
      const float* flg = p.fl + bt * Ti * F;
      for (int e = tid; e < Ti * F; e += ANT) fl[e] = flg[e];
    }
    if (tid < UQ) pqv[tid] = p.pq[bt * UQ + tid];
    if (tid < CT) {
      const float g = dout[(size_t)t * OW + A + tid] + dvec[tid];
      dctx[tid] = g;
      pb.dctx[bt * CT + tid] = g;
    }
    __syncthreads();
    PROF(1);
    // (b) d alpha / d a2 from the contexts: one wave per memory row, 16 B per lane, 4 rows in flight
    {
      float dcr[NQ];
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) dcr[qq] = (d0 + qq) < V1 ? dctx[d0 + qq] : 0.f;
      const float dc2 = lane < V2 ? dctx[V1 + lane] : 0.f;
      const float* vb = values1 + d0;
      for (int t0 = wave; t0 < Ti; t0 += 4 * AW) {
        float s1[4], s2[4];
        float4 r[4];
        float w2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int tt = t0 + u * AW;
          r[u] = make_float4(0.f, 0.f, 0.f, 0.f); w2[u] = 0.f;
          if (tt < len) {
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
#pragma unroll
        for (int u = 0; u < 4; ++u) { s1[u] = red8[u]; s2[u] = red8[4 + u]; }
        if (lane == 0) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int tt = t0 + u * AW;
            if (tt < Ti) {
              dal[tt] = s1[u] + dalc[tt] + (pb.dalign1 ? pb.dalign1[bt * Ti + tt] : 0.f);
              da2[tt] = s2[u] + (pb.dalign2 ? pb.dalign2[bt * Ti + tt] : 0.f);
            }
          }
        }
      }
    }
    __syncthreads();
    PROF(2);
    // (c) forward-attention recursion + softmax backward (wave 0), additive softmax backward (wave 1)
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
        const float dalp = (dal[tt] - s1) * invS;       // d alpha'
        const float da = dalp * w + dac[tt];             // total d a_t
        dal[tt] = dalp * a[tt];                          // d w
        de1[tt] = da;
        s2 += da * a[tt];
      }
      s2 = wave_sum(s2);
      float* g1 = pb.de1 + bt * Ti;
      for (int tt = lane; tt < Ti; tt += 64) { const float v = a[tt] * (de1[tt] - s2); de1[tt] = v; g1[tt] = v; }
    } else if (wave == 1) {
      float s3 = 0.f;
      for (int tt = lane; tt < Ti; tt += 64) s3 += da2[tt] * a2[tt];
      s3 = wave_sum(s3);
      float* g2 = pb.de2 + bt * Ti;
      for (int tt = lane; tt < Ti; tt += 64) { const float v = a2[tt] * (da2[tt] - s3); da2[tt] = v; g2[tt] = v; }
    }
    __syncthreads();
    PROF(3);
    // new carry for alpha_{t-1}: d alpha_prev[s] = 0.5*dw[s] + 0.5*dw[s+1]
    for (int i = tid; i < Ti; i += ANT) dalc[i] = 0.5f * dal[i] + 0.5f * (i + 1 < Ti ? dal[i + 1] : 0.f);
    // (d) energy backward (recurrent part only): d pq and d location-features
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
      for (int t0 = wave; t0 < Ti; t0 += RB * AW) {
        float dfp[RB * F];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const int tt = t0 + u * AW;
#pragma unroll
          for (int k = 0; k < F; ++k) dfp[u * F + k] = 0.f;
          if (tt < len) {
            const float de = de1[tt];
            float f[F];
#pragma unroll
            for (int k = 0; k < F; ++k) f[k] = fl[tt * F + k];
            float kk[NQ];
            load_key4<KLDS>(keys1, K1s, tt, U1, d0, actU, kk);
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq) {
              float lf = 0.f;
#pragma unroll
              for (int k = 0; k < F; ++k) lf += f[k] * Ur[qq][k];
              const float th = tanhf_(kk[qq] + pqb[qq] + lf);
              const float g = de * v1r[qq] * (1.f - th * th);     // 0 on inactive lanes (v1r == 0)
              dpqa[qq] += g;
#pragma unroll
              for (int k = 0; k < F; ++k) dfp[u * F + k] += g * Ur[qq][k];
            }
            if (lane < U2) {
              const float th2 = tanhf_(load_key1<KLDS>(keys2, K2s, tt, U2, lane, true) + pq2);
              dpq2a += da2[tt] * v2r * (1.f - th2 * th2);
            }
          }
        }
        wave_sum_multi<RB * F>(dfp);
        if (lane < RB * F) {
          const int u = lane / F, k = lane - u * F, tt = t0 + u * AW;
          float v = dfp[0];
#pragma unroll
          for (int i = 1; i < RB * F; ++i) v = (lane == i) ? dfp[i] : v;
          if (tt < Ti) { dfl[tt * F + k] = v; dflg[tt * F + k] = v; }
        }
      }
      // cross-wave reduction of d pq
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) { const int d = d0 + qq; if (d < U1) partial[wave * UQ + d] = dpqa[qq]; }
      if (lane < U2) partial[wave * UQ + U1 + lane] = dpq2a;
    }
    __syncthreads();
    PROF(4);
    if (tid < UQ) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < AW; ++w) s += partial[w * UQ + tid];
      dpq[tid] = s;
      pb.dpq[bt * UQ + tid] = s;
    }
    // (e) location conv backward: carry for a_{t-1}
    for (int s = tid; s < Ti; s += ANT) {
      float g = 0.f;
      for (int jj = 0; jj < KW; ++jj) {
        const int tt = s - jj + PL;
        if (tt >= 0 && tt < Ti) {
#pragma unroll
          for (int k = 0; k < F; ++k) g += dfl[tt * F + k] * Fs[jj * F + k];
        }
      }
      dac[s] = g;
    }
    __syncthreads();
    PROF(5);
    // (f) d query = dpq x Wq^T
    matvec_bf16<ANT, MVU>(dpq, pb.WqT, UQ, A, partial, dq);
    PROF(6);
    // (g) LSTM cell backward
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
      float* dr = pb.dxg + bt * G;
      dr[j] = dzi; dr[A + j] = dzj; dr[2 * A + j] = dzf; dr[3 * A + j] = dzo;
      dz[j] = dzi; dz[A + j] = dzj; dz[2 * A + j] = dzf; dz[3 * A + j] = dzo;
    }
    __syncthreads();
    PROF(7);
    // (h) gradient wrt [ctx_{t-1} | hstate_{t-1}]
    matvec_bf16<ANT, MVU>(dz, pb.WrecT, G, CT + A, partial, dvec);
    if (tid < A) dh_state = dvec[CT + tid] + dh_direct;
    __syncthreads();
    PROF(8);
  }
  PROF_STORE(16);
}

// ---- post-loop parameter / key gradients: one workgroup per (sample, PG_ROWS memory rows); thread = attention unit
constexpr int PG_ROWS = 4;
template <int F>
__global__ void attn_param_grads_k(const satt_attn_rnn_params p, const float* __restrict__ de1g,
                                   const float* __restrict__ de2g, float* __restrict__ dkeys1,
                                   float* __restrict__ dkeys2, float* __restrict__ dv1, float* __restrict__ db1,
                                   float* __restrict__ dlocU, float* __restrict__ dv2, int t0, int t1, int accumulate) {
  const int U1 = p.U1, U2 = p.U2, UQ = U1 + U2, Ti = p.Ti, Td = p.Td;
  const int b = blockIdx.y, d = threadIdx.x;
  if (d >= UQ) return;
  const bool m1 = d < U1;
  const int len = (int)p.lengths[b];
  const float v = m1 ? p.v1[d] : p.v2[d - U1];
  const float bb = m1 ? p.b1[d] : 0.f;
  float Uc[F];
#pragma unroll
  for (int k = 0; k < F; ++k) Uc[k] = m1 ? p.locU[k * U1 + d] : 0.f;
  float dv = 0.f, db = 0.f, dU[F];
#pragma unroll
  for (int k = 0; k < F; ++k) dU[k] = 0.f;
  const float* pqb = p.pq + (size_t)b * Td * UQ + d;
  if ((Ti & 3) == 0) {
    // Steps outer, the PG_ROWS rows of this workgroup inner: the operands that are uniform over the units (d e and the
    // location features of the 4 rows) are contiguous - 1 + F 16-byte loads per step instead of 4 * (1 + F) scalar ones
    // - and the processed query of a step is loaded once for all rows.  (The row-outer form below issued 7 vector
    // loads per element, 6 of them broadcasts: the kernel was bound by load issue.)
    static_assert(PG_ROWS == 4, "the vector form loads the rows of a workgroup as one float4");
    const int tt0 = blockIdx.x * PG_ROWS;
    float key[PG_ROWS], dk[PG_ROWS];
    bool rv[PG_ROWS];
#pragma unroll
    for (int r = 0; r < PG_ROWS; ++r) {
      const int tt = tt0 + r;
      rv[r] = tt < len;
      float k = m1 ? p.keys1[((size_t)b * Ti + tt) * U1 + d] : p.keys2[((size_t)b * Ti + tt) * U2 + (d - U1)];
      if (p.keys_lds_bf16) k = bf2f(f2bf(k));      // the loop used the bf16-rounded key
      key[r] = k + bb; dk[r] = 0.f;
    }
    const float* deg = (m1 ? de1g : de2g) + (size_t)b * Td * Ti + tt0;
    const float* flg = p.fl + ((size_t)b * Td * Ti + tt0) * F;
#pragma unroll 4
    for (int t = t0; t < t1; ++t) {
      const float4 de4 = *reinterpret_cast<const float4*>(deg + (size_t)t * Ti);
      const float de[PG_ROWS] = {rv[0] ? de4.x : 0.f, rv[1] ? de4.y : 0.f, rv[2] ? de4.z : 0.f, rv[3] ? de4.w : 0.f};
      float fv[PG_ROWS * F];
      const float4* f4 = reinterpret_cast<const float4*>(flg + (size_t)t * Ti * F);
#pragma unroll
      for (int q = 0; q < F; ++q) { const float4 w = f4[q]; fv[4 * q] = w.x; fv[4 * q + 1] = w.y; fv[4 * q + 2] = w.z; fv[4 * q + 3] = w.w; }
      const float pqv = pqb[(size_t)t * UQ];
#pragma unroll
      for (int r = 0; r < PG_ROWS; ++r) {
        float zz = key[r] + pqv;
#pragma unroll
        for (int k = 0; k < F; ++k) zz += fv[r * F + k] * Uc[k];      // Uc == 0 for mechanism 2
        const float th = tanhf_(zz);
        const float g = de[r] * v * (1.f - th * th);
        dk[r] += g; dv += de[r] * th; db += g;
#pragma unroll
        for (int k = 0; k < F; ++k) dU[k] += fv[r * F + k] * g;
      }
    }
#pragma unroll
    for (int r = 0; r < PG_ROWS; ++r) {
      const int tt = tt0 + r;
      float* dst = m1 ? dkeys1 + ((size_t)b * Ti + tt) * U1 + d : dkeys2 + ((size_t)b * Ti + tt) * U2 + (d - U1);
      if (rv[r]) *dst = accumulate ? *dst + dk[r] : dk[r];
      else if (!accumulate) *dst = 0.f;
    }
  } else
  for (int r = 0; r < PG_ROWS; ++r) {
    const int tt = blockIdx.x * PG_ROWS + r;
    if (tt >= Ti) break;
    float* dst = m1 ? dkeys1 + ((size_t)b * Ti + tt) * U1 + d : dkeys2 + ((size_t)b * Ti + tt) * U2 + (d - U1);
    if (tt >= len) { if (!accumulate) *dst = 0.f; continue; }
    float key = m1 ? p.keys1[((size_t)b * Ti + tt) * U1 + d] : p.keys2[((size_t)b * Ti + tt) * U2 + (d - U1)];
    if (p.keys_lds_bf16) key = bf2f(f2bf(key));      // the loop used the bf16-rounded key
    key += bb;
    const float* deg = (m1 ? de1g : de2g) + (size_t)b * Td * Ti + tt;
    const float* flg = p.fl + ((size_t)b * Td * Ti + tt) * F;
    float dk = 0.f;
#pragma unroll 2
    for (int t = t0; t < t1; ++t) {
      const float de = deg[(size_t)t * Ti];
      float zz = key + pqb[(size_t)t * UQ];
      float f[F];
#pragma unroll
      for (int k = 0; k < F; ++k) { f[k] = flg[(size_t)t * Ti * F + k]; zz += f[k] * Uc[k]; }   // Uc == 0 for mech 2
      const float th = tanhf_(zz);
      const float g = de * v * (1.f - th * th);
      dk += g; dv += de * th; db += g;
#pragma unroll
      for (int k = 0; k < F; ++k) dU[k] += f[k] * g;
    }
    *dst = accumulate ? *dst + dk : dk;
  }
  if (m1) {
    atomicAdd(&dv1[d], dv); atomicAdd(&db1[d], db);
#pragma unroll
    for (int k = 0; k < F; ++k) atomicAdd(&dlocU[k * U1 + d], dU[k]);
  } else {
    atomicAdd(&dv2[d - U1], dv);
  }
}

// The same sums from the SAVED s = r - 1/2 of the forward pass (satt_attn_rnn_params.saf; tanh = -2 s): no keys, no processed query,
// no score argument, no exp / rcp.  A WAVE owns one (sample, memory row) item at a time and lane l the units 4l .. 4l+3: the factors
// of a step are one 8-byte load per thread (the rows of a workgroup's eight waves are 4 KB contiguous per step), d e and the
// location features of the row are wave-uniform.  The grid is FIXED (pgs_nwg workgroups, ~2 items per wave at the benchmark
// shape): the per-unit sums (d v, d b, d U) stay in registers across a wave's items, meet in LDS once per workgroup and leave as
// ONE plain read-modify-write of the workgroup's own float64 slot acc[wg][.] - until r4 every workgroup of a (rows / 4, B) grid
// added its 1792 sums with float64 atomics: 2.3 M contended atomics per launch, 40+ us of a 16-step launch (the pieces that run
// after the recurrent loop has ended are that short, and the encoder backward waits for them).
constexpr int PGS_WAVES = 8;
constexpr int PGS_TB = 8;      // steps per load batch
__host__ __device__ inline int pgs_nwg(int B, int Ti) { const int w = (B * Ti + 2 * PGS_WAVES - 1) / (2 * PGS_WAVES); return w < 1 ? 1 : (w > 320 ? 320 : w); }
template <int F>
__global__ __launch_bounds__(64 * PGS_WAVES) void attn_param_grads_saf_k(const satt_attn_rnn_params p, const float* __restrict__ de1g,
                                                                         const float* __restrict__ de2g, float* __restrict__ dkeys1,
                                                                         float* __restrict__ dkeys2, float* __restrict__ dv1,
                                                                         float* __restrict__ db1, float* __restrict__ dlocU,
                                                                         float* __restrict__ dv2, double* __restrict__ acc,
                                                                         int t0, int t1, int accumulate) {
  extern __shared__ float pad_[];                      // (dynamic LDS = the caller's placement pad: never touched)
  __shared__ float red[PGS_WAVES][64 * 4 + 4];
  const int U1 = p.U1, U2 = p.U2, UQ = U1 + U2, Ti = p.Ti, Td = p.Td;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int u0 = 4 * lane;
  const bool act = u0 < UQ, m1 = u0 < U1;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int u = u0 + j;
    v[j] = !act ? 0.f : (m1 ? p.v1[min(u, U1 - 1)] : p.v2[min(u - U1, U2 - 1)]);
  }
  float dv[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f}, dU[F][4];
#pragma unroll
  for (int k = 0; k < F; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) dU[k][j] = 0.f;
  const int nitems = p.B * Ti;
  for (int item = (int)blockIdx.x * PGS_WAVES + w; item < nitems; item += (int)gridDim.x * PGS_WAVES) {
    const int b = item / Ti, tt = item - b * Ti;
    const bool rowok = tt < (int)p.lengths[b];
    float dk[4] = {0.f, 0.f, 0.f, 0.f};
    if (rowok) {
      const float* der = (m1 ? de1g : de2g) + (size_t)b * Td * Ti + tt;      // (per-lane pointer: ONE vector load per step)
      const float* flr = p.fl + ((size_t)b * Td * Ti + tt) * F;
      const __fp16* sr = reinterpret_cast<const __fp16*>(p.saf) + ((size_t)b * Td * Ti + tt) * UQ + min(u0, UQ - 4);
      // PGS_TB steps per batch: every load of the batch is issued (branch-free, step clamped) before the first value is used -
      // with the loads inside a rolled loop each step paid two dependent memory round trips (the compiler does not unroll a loop
      // with a run-time trip count around them)
      for (int tb = t0; tb < t1; tb += PGS_TB) {
        typedef __attribute__((ext_vector_type(4))) __fp16 h4;
        h4 s4[PGS_TB]; float dev[PGS_TB], f[PGS_TB][F];
#pragma unroll
        for (int i = 0; i < PGS_TB; ++i) {
          const size_t t = (size_t)min(tb + i, t1 - 1);
          s4[i] = *reinterpret_cast<const h4*>(sr + t * Ti * UQ);
          dev[i] = der[t * Ti];
#pragma unroll
          for (int k = 0; k < F; ++k) f[i][k] = flr[t * Ti * F + k];
        }
#pragma unroll
        for (int i = 0; i < PGS_TB; ++i) {
          const float de = tb + i < t1 ? dev[i] : 0.f;      // (steps beyond the range: d e = 0 adds nothing)
          float g[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float th = -2.f * (float)s4[i][j];
            g[j] = de * v[j] * (1.f - th * th);
            dk[j] += g[j]; dv[j] += de * th; db[j] += g[j];
          }
          // d U: explicit two-wide products against a SPLAT of the feature.  (The scalar form compiled to v_pk_fma_f32 with an op_sel
          // broadcast of one half of a register pair; with that code the sums of 16 elements - filter 1, even units 192..222 - came
          // out 1e-8 off in about every second step when the kernel ran beside the recurrent kernels, never in isolation, with
          // bit-identical inputs and per-workgroup partials: tools/probes/grad_diff_map.py, saf_determinism.py.  Not understood;
          // this form is clean.)
          typedef __attribute__((ext_vector_type(2))) float f2_t;
          const f2_t g01 = (f2_t){g[0], g[1]}, g23 = (f2_t){g[2], g[3]};
#pragma unroll
          for (int k = 0; k < F; ++k) {
            float fk = f[i][k];
            asm volatile("" : "+v"(fk));                     // a VGPR copy of the feature: no op_sel broadcast out of a pair
            const f2_t fs = (f2_t){fk, fk};
            f2_t a01 = (f2_t){dU[k][0], dU[k][1]}, a23 = (f2_t){dU[k][2], dU[k][3]};
            a01 = g01 * fs + a01; a23 = g23 * fs + a23;
            dU[k][0] = a01.x; dU[k][1] = a01.y; dU[k][2] = a23.x; dU[k][3] = a23.y;
          }
        }
      }
    }
    if (act) {           // d keys of this row: one writer per element
      float* dst = m1 ? dkeys1 + ((size_t)b * Ti + tt) * U1 + u0 : dkeys2 + ((size_t)b * Ti + tt) * U2 + (u0 - U1);
      float4 o = rowok ? make_float4(dk[0], dk[1], dk[2], dk[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (accumulate && rowok) { const float4 c = *reinterpret_cast<const float4*>(dst); o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
      if (rowok || !accumulate) *reinterpret_cast<float4*>(dst) = o;
    }
  }
  // unit sums over the waves of this workgroup, quantity by quantity (q = 0: d v, 1: d b, 2..: d U[q - 2]); layout of a slot and of
  // the fp32 outputs: [dv1 | db1 | dU (F x U1) | dv2]
  const int nq = (2 + F) * U1 + U2;
  double* slot = acc ? acc + (size_t)blockIdx.x * nq : nullptr;
#pragma unroll
  for (int q = 0; q < 2 + F; ++q) {
#pragma unroll
    for (int j = 0; j < 4; ++j) red[w][u0 + j] = q == 0 ? dv[j] : q == 1 ? db[j] : dU[q >= 2 ? q - 2 : 0][j];
    __syncthreads();
    const int u = threadIdx.x;
    if (u < UQ && (u < U1 || q == 0)) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < PGS_WAVES; ++r) sum += red[r][u];
      const int i = u < U1 ? q * U1 + u : (2 + F) * U1 + (u - U1);
      if (slot) slot[i] = accumulate ? slot[i] + (double)sum : (double)sum;
      else if (u < U1) { if (q == 0) atomicAdd(&dv1[u], sum); else if (q == 1) atomicAdd(&db1[u], sum); else atomicAdd(&dlocU[(q - 2) * U1 + u], sum); }
      else atomicAdd(&dv2[u - U1], sum);
    }
    __syncthreads();
  }
}
// the workgroup slots -> the fp32 gradients (+=): float64 sums in a fixed order (these sums cancel heavily - softmax gradients sum to
// zero over the rows - and fp32 atomics in arrival order left 5e-8 of noise on a 6e-7 gradient).  One block = 4 elements x 64 slot
// groups: a thread adds nwg / 64 slots (5 loads in flight, not 40 dependent ones), a wave sum finishes the element.
template <int F>
__global__ __launch_bounds__(256) void attn_param_grads_finish_k(const double* __restrict__ acc, int nwg, int U1, int U2, float* __restrict__ dv1,
                                                                 float* __restrict__ db1, float* __restrict__ dlocU, float* __restrict__ dv2) {
  const int n1 = (2 + F) * U1, n = n1 + U2;
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  double sum = 0.0;
  if (i < n)
    for (int wg = lane; wg < nwg; wg += 64) sum += acc[(size_t)wg * n + i];
  // fixed-order tree over the 64 lanes (DPP-free: two 32-bit halves through ds_bpermute-less shuffles of the compiler)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
  if (lane == 0 && i < n) {
    const float v = (float)sum;
    if (i < U1) dv1[i] += v;
    else if (i < 2 * U1) db1[i - U1] += v;
    else if (i < n1) dlocU[i - 2 * U1] += v;
    else dv2[i - n1] += v;
  }
}

inline int check(const satt_attn_rnn_params& p, bool loop = true) {
  if (p.B <= 0 || p.Td <= 0 || p.Ti <= 0) return SATT_E_BADARG;
  // location_sensitive / cumulative: cluster kernels only (the deferred parameter gradients do not depend on either)
  if (loop && (p.att1_mode != 0 || p.cumulative != 0 || p.agentW != nullptr)) return SATT_E_UNSUPPORTED;
  if (p.filters != 5) return SATT_E_UNSUPPORTED;
  if (p.U1 > 64 * NQ || p.V1 > 64 * NQ || p.U2 > 64 || p.V2 > 64 || p.U1 % 4 || p.V1 % 4) return SATT_E_UNSUPPORTED;
  if ((4 * p.A) % 8 || (p.U1 + p.U2) % 8 || (p.V1 + p.V2 + p.A) % 8 || p.A % 8) return SATT_E_UNSUPPORTED;
  if (4 * p.A > 8 * ANT || p.A > ANT || p.kernel < 1) return SATT_E_UNSUPPORTED;
  if (p.U1 + p.U2 > ANT || p.V1 + p.V2 > ANT) return SATT_E_UNSUPPORTED;
  return SATT_OK;
}

}  // namespace

extern "C" int satt_attn_rnn_fwd(const satt_attn_rnn_params* pp, void* stream) {
  if (!pp) return SATT_E_BADARG;
  if (pp->teach1 || pp->teach2) return SATT_E_UNSUPPORTED;   // forced alignments: cluster kernels only
  int rc = check(*pp);
  if (rc) return rc;
  satt_attn_rnn_params p = *pp;
  single_source_fixup(p);
  const int CT = p.V1 + p.V2, UQ = p.U1 + p.U2;
  const bool klds = p.keys_lds_bf16 != 0;
  const size_t smem = sizeof(float) * carve_fwd(p.A, CT, UQ, p.Ti, 5, p.kernel, p.U1, p.U2, klds).total;
  if (smem > 160 * 1024) return SATT_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (klds) {
    if (smem > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)attn_rnn_fwd_k<5, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem);
    hipLaunchKernelGGL((attn_rnn_fwd_k<5, true>), dim3(p.B), dim3(ANT), smem, s, p);
  } else {
    if (smem > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)attn_rnn_fwd_k<5, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem);
    hipLaunchKernelGGL((attn_rnn_fwd_k<5, false>), dim3(p.B), dim3(ANT), smem, s, p);
  }
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

extern "C" int satt_attn_rnn_bwd(const satt_attn_rnn_bwd_params* pp, void* stream) {
  if (!pp) return SATT_E_BADARG;
  int rc = check(pp->f);
  if (rc) return rc;
  satt_attn_rnn_bwd_params q = *pp;
  single_source_fixup(q.f);
  const satt_attn_rnn_params& p = q.f;
  const int CT = p.V1 + p.V2, UQ = p.U1 + p.U2;
  const bool klds = p.keys_lds_bf16 != 0;
  const size_t smem = sizeof(float) * carve_bwd(p.A, CT, UQ, p.Ti, 5, p.kernel, p.U1, p.U2, klds).total;
  if (smem > 160 * 1024) return SATT_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (klds) {
    if (smem > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)attn_rnn_bwd_k<5, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem);
    hipLaunchKernelGGL((attn_rnn_bwd_k<5, true>), dim3(p.B), dim3(ANT), smem, s, q);
  } else {
    if (smem > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)attn_rnn_bwd_k<5, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem);
    hipLaunchKernelGGL((attn_rnn_bwd_k<5, false>), dim3(p.B), dim3(ANT), smem, s, q);
  }
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

/* steps [t0, t1) only; accumulate != 0 adds to dkeys1/2 (the parameter gradients always accumulate).  lds_pad_bytes
 * of dynamic LDS are requested but not used: a large pad keeps these workgroups off the CUs that host the persistent
 * recurrent kernels when the call is overlapped with them on another stream. */
static int param_grads_launch(const satt_attn_rnn_params* f, const float* de1, const float* de2, float* dkeys1, float* dkeys2,
                              float* dv1, float* db1, float* dlocU, float* dv2, double* acc, int t0, int t1, int accumulate,
                              int lds_pad_bytes, void* stream);
extern "C" int satt_attn_param_grads_range(const satt_attn_rnn_params* f, const float* de1, const float* de2,
                                           float* dkeys1, float* dkeys2, float* dv1, float* db1, float* dlocU,
                                           float* dv2, int t0, int t1, int accumulate, int lds_pad_bytes, void* stream) {
  return param_grads_launch(f, de1, de2, dkeys1, dkeys2, dv1, db1, dlocU, dv2, nullptr, t0, t1, accumulate, lds_pad_bytes, stream);
}
extern "C" int64_t satt_attn_param_grads_acc_doubles(const satt_attn_rnn_params* f) {
  return f ? (int64_t)pgs_nwg(f->B, f->Ti) * ((2 + 5) * f->U1 + f->U2) : 0;      // one slot per workgroup of the fixed grid
}
extern "C" int satt_attn_param_grads_acc(const satt_attn_rnn_params* f, const float* de1, const float* de2, float* dkeys1,
                                         float* dkeys2, double* acc, int t0, int t1, int accumulate, int lds_pad_bytes,
                                         void* stream) {
  if (!f || !acc || !f->saf) return SATT_E_BADARG;
  return param_grads_launch(f, de1, de2, dkeys1, dkeys2, nullptr, nullptr, nullptr, nullptr, acc, t0, t1, accumulate, lds_pad_bytes, stream);
}
extern "C" int satt_attn_param_grads_finish(const satt_attn_rnn_params* f, double* acc, float* dv1, float* db1, float* dlocU,
                                            float* dv2, void* stream) {
  if (!f || !acc || !dv1 || !db1 || !dlocU || (f->U2 > 0 && !dv2)) return SATT_E_BADARG;
  const int n = (2 + 5) * f->U1 + f->U2;
  hipLaunchKernelGGL(attn_param_grads_finish_k<5>, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, acc, pgs_nwg(f->B, f->Ti),
                     f->U1, f->U2, dv1, db1, dlocU, dv2);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}
static int param_grads_launch(const satt_attn_rnn_params* f, const float* de1, const float* de2, float* dkeys1, float* dkeys2,
                              float* dv1, float* db1, float* dlocU, float* dv2, double* acc, int t0, int t1, int accumulate,
                              int lds_pad_bytes, void* stream) {
  if (!f) return SATT_E_BADARG;
  int rc = check(*f, false);
  if (rc) return rc;
  if (t0 < 0 || t1 > f->Td || t0 >= t1 || lds_pad_bytes < 0 || lds_pad_bytes > 160 * 1024) return SATT_E_BADARG;
  satt_attn_rnn_params fp = *f;
  single_source_fixup(fp);
  const int UQ = f->U1 + f->U2;
  const int nt = (UQ + 63) / 64 * 64;
  if (lds_pad_bytes > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)attn_param_grads_k<5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_pad_bytes);
  static const bool nosaf = getenv("SATT_PG_NOSAF") != nullptr;      // diagnosis switch
  if (!nosaf && fp.saf && UQ == 256 && f->U1 % 4 == 0 && f->U2 % 4 == 0) {     // saved factors of the forward pass: see the kernel
    const int pad = std::max(0, lds_pad_bytes - (int)(sizeof(float) * PGS_WAVES * (64 * 4 + 4)));
    if (pad > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)attn_param_grads_saf_k<5>, hipFuncAttributeMaxDynamicSharedMemorySize, pad);
    hipLaunchKernelGGL(attn_param_grads_saf_k<5>, dim3(pgs_nwg(f->B, f->Ti)), dim3(64 * PGS_WAVES), (size_t)pad,
                       (hipStream_t)stream, fp, de1, de2, dkeys1, dkeys2, dv1, db1, dlocU, dv2, acc, t0, t1, accumulate);
  } else if (acc) return SATT_E_UNSUPPORTED;
  else
  hipLaunchKernelGGL(attn_param_grads_k<5>, dim3((f->Ti + PG_ROWS - 1) / PG_ROWS, f->B), dim3(nt), (size_t)lds_pad_bytes,
                     (hipStream_t)stream, fp, de1, de2, dkeys1, dkeys2, dv1, db1, dlocU, dv2, t0, t1, accumulate);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}
extern "C" int satt_attn_param_grads(const satt_attn_rnn_params* f, const float* de1, const float* de2, float* dkeys1,
                                     float* dkeys2, float* dv1, float* db1, float* dlocU, float* dv2, void* stream) {
  if (!f) return SATT_E_BADARG;
  return satt_attn_param_grads_range(f, de1, de2, dkeys1, dkeys2, dv1, db1, dlocU, dv2, 0, f->Td, 0, 0, stream);
}

#ifdef SATT_PROFILE
extern "C" int satt_prof_read(unsigned long long* host32) {
  return hipMemcpyFromSymbol(host32, HIP_SYMBOL(satt_prof_acc), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -3;
}
#endif
