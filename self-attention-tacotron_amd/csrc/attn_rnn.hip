// Dual-source attention RNN loop (AttentionWrapper[ZoneoutLSTM + ForwardAttention + BahdanauAttention]) as ONE
// persistent 512-thread workgroup per sample walking all Td teacher-forced steps; samples are independent so no
// inter-workgroup communication is needed.  Recurrent weights are bf16 streamed from L2 (matvec.h); keys/values
// rows are read coalesced (one wave per memory row); attention state, energies and alignments stay fp32 in LDS.
// Follows reference modules/forward_attention.py:88-122 (ForwardAttention.__call__), :13-26 (score),
// :128-136 (initial state), TF BahdanauAttention (modules/attentions.py:53-57) and SURVEY.md A.7-A.9.
#include "matvec.h"

namespace {

constexpr int ANT = 512;
constexpr int AW = ANT / 64;  // waves
constexpr int NQ = 4;         // U1, V1 <= 256

__device__ __forceinline__ int up4(int x) { return (x + 3) & ~3; }

// softmax over v[0..len) by ONE wave (in place), zeros beyond len up to n
__device__ __forceinline__ void wave_softmax(float* v, int len, int n, int lane) {
  float m = -INFINITY;
  for (int t = lane; t < len; t += 64) m = fmaxf(m, v[t]);
  m = wave_max(m);
  float s = 0.f;
  for (int t = lane; t < len; t += 64) { float e = expf(v[t] - m); v[t] = e; s += e; }
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int t = lane; t < n; t += 64) v[t] = (t < len) ? v[t] * inv : 0.f;
}

template <int F>
__global__ __launch_bounds__(ANT) void attn_rnn_fwd_k(const satt_attn_rnn_params p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int A = p.A, G = 4 * A, V1 = p.V1, V2 = p.V2, CT = V1 + V2, U1 = p.U1, U2 = p.U2, UQ = U1 + U2;
  const int Ti = p.Ti, Td = p.Td, KW = p.kernel, PL = (KW - 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  float* vec = smem;                      // [CT + A]  ctx1 | ctx2 | h_state
  float* z = vec + up4(CT + A);           // [G]
  float* q = z + G;                       // [A]   query = h' (pre-zoneout)
  float* pq = q + up4(A);                 // [UQ]
  float* aprev = pq + up4(UQ);            // [Ti]  previous softmax probs (location conv input)
  float* alA = aprev + up4(Ti);           // [Ti]  alpha ping
  float* alB = alA + up4(Ti);             // [Ti]  alpha pong
  float* e1 = alB + up4(Ti);              // [Ti]
  float* e2 = e1 + up4(Ti);               // [Ti]
  float* fl = e2 + up4(Ti);               // [Ti*F]
  float* Fs = fl + up4(Ti * F);           // [KW*F]
  float* bFs = Fs + up4(KW * F);          // [F]
  float* partial = bFs + up4(F);          // [ANT*8]

  const int len = (int)p.lengths[b];
  const uint32_t seed = p.seed ? *p.seed : 0u;
  const float* xg = p.xg + (size_t)b * Td * G;
  const float* keys1 = p.keys1 + (size_t)b * Ti * U1;
  const float* values1 = p.values1 + (size_t)b * Ti * V1;
  const float* keys2 = p.keys2 + (size_t)b * Ti * U2;
  const float* values2 = p.values2 + (size_t)b * Ti * V2;
  const int OW = A + CT;
  float* out = p.out + (size_t)b * Td * OW;

  // per-lane constants of the energy pass
  float v1r[NQ], b1r[NQ], Ur[NQ][F];
#pragma unroll
  for (int qq = 0; qq < NQ; ++qq) {
    const int d = lane + 64 * qq;
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
  float c = 0.f, h = 0.f;
  float* alp = alA;  // alpha_{t-1}
  float* aln = alB;  // alpha_t
  __syncthreads();

  for (int t = 0; t < Td; ++t) {
    float xi = 0.f, xj = 0.f, xf = 0.f, xo = 0.f;
    if (tid < A) {
      const float* xr = xg + (size_t)t * G;
      xi = xr[tid]; xj = xr[A + tid]; xf = xr[2 * A + tid]; xo = xr[3 * A + tid];
    }
    // (1) recurrent gate pre-activations: [ctx_{t-1} | h_{t-1}] x Wrec
    matvec_bf16<ANT>(vec, p.Wrec, CT + A, G, partial, z);
    // (2) LSTM cell + zoneout
    if (tid < A) {
      const int j = tid;
      const float gi = sigmoidf_(xi + z[j]);
      const float gj = tanhf(xj + z[A + j]);
      const float gf = sigmoidf_(xf + z[2 * A + j] + 1.0f);
      const float go = sigmoidf_(xo + z[3 * A + j]);
      const float cn = gf * c + gi * gj;
      const float hn = go * tanhf(cn);
      const size_t bt = (size_t)b * Td + t;
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
    // (3) processed queries for both mechanisms
    matvec_bf16<ANT>(q, p.Wq, A, UQ, partial, pq);
    if (tid < UQ) p.pq[((size_t)b * Td + t) * UQ + tid] = pq[tid];
    // (4) location features f = conv1d_SAME(a_{t-1}) + bias
    for (int e = tid; e < Ti * F; e += ANT) {
      const int tt = e / F, k = e - tt * F;
      float s = bFs[k];
      for (int jj = 0; jj < KW; ++jj) {
        const int src = tt + jj - PL;
        if (src >= 0 && src < Ti) s += aprev[src] * Fs[jj * F + k];
      }
      fl[e] = s;
    }
    __syncthreads();
    // (5) energies: one wave per memory row
    {
      float pqb[NQ];
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) { const int d = lane + 64 * qq; pqb[qq] = d < U1 ? pq[d] + b1r[qq] : 0.f; }
      const float pq2 = lane < U2 ? pq[U1 + lane] : 0.f;
      for (int tt = wave; tt < len; tt += AW) {
        const float* kr = keys1 + (size_t)tt * U1;
        float f[F];
#pragma unroll
        for (int k = 0; k < F; ++k) f[k] = fl[tt * F + k];
        float acc = 0.f;
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) {
          const int d = lane + 64 * qq;
          if (d < U1) {
            float lf = 0.f;
#pragma unroll
            for (int k = 0; k < F; ++k) lf += f[k] * Ur[qq][k];
            acc += v1r[qq] * tanhf(kr[d] + pqb[qq] + lf);
          }
        }
        float acc2 = lane < U2 ? v2r * tanhf(keys2[(size_t)tt * U2 + lane] + pq2) : 0.f;
        acc = wave_sum(acc);
        acc2 = wave_sum(acc2);
        if (lane == 0) { e1[tt] = acc; e2[tt] = acc2; }
      }
    }
    __syncthreads();
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
      float* o1 = p.align1 + ((size_t)b * Td + t) * Ti;
      float* oa = p.a1 + ((size_t)b * Td + t) * Ti;
      for (int tt = lane; tt < Ti; tt += 64) {
        const float v = aln[tt] * inv;
        aln[tt] = v; o1[tt] = v;
        const float a = e1[tt];
        oa[tt] = a; aprev[tt] = a;
      }
    } else if (wave == 1) {
      wave_softmax(e2, len, Ti, lane);
      float* o2 = p.align2 + ((size_t)b * Td + t) * Ti;
      for (int tt = lane; tt < Ti; tt += 64) o2[tt] = e2[tt];
    }
    __syncthreads();
    // (7) contexts
    {
      const int NS1 = ANT / V1, c1 = tid % V1, s1 = tid / V1;
      if (s1 < NS1) {
        float acc = 0.f;
        for (int tt = s1; tt < len; tt += NS1) acc += aln[tt] * values1[(size_t)tt * V1 + c1];
        partial[s1 * V1 + c1] = acc;
      }
      const int NS2 = ANT / V2, c2 = tid % V2, s2 = tid / V2;
      if (s2 < NS2) {
        float acc = 0.f;
        for (int tt = s2; tt < len; tt += NS2) acc += e2[tt] * values2[(size_t)tt * V2 + c2];
        partial[ANT + s2 * V2 + c2] = acc;
      }
      __syncthreads();
      if (tid < V1) {
        float s = 0.f;
        for (int k = 0; k < NS1; ++k) s += partial[k * V1 + tid];
        vec[tid] = s; out[(size_t)t * OW + A + tid] = s;
      } else if (tid < CT) {
        const int cc = tid - V1;
        float s = 0.f;
        for (int k = 0; k < NS2; ++k) s += partial[ANT + k * V2 + cc];
        vec[V1 + cc] = s; out[(size_t)t * OW + A + V1 + cc] = s;
      }
    }
    { float* tmp = alp; alp = aln; aln = tmp; }
    __syncthreads();
  }
}

template <int F>
__global__ __launch_bounds__(ANT) void attn_rnn_bwd_k(const satt_attn_rnn_bwd_params pb) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const satt_attn_rnn_params& p = pb.f;
  const int A = p.A, G = 4 * A, V1 = p.V1, V2 = p.V2, CT = V1 + V2, U1 = p.U1, U2 = p.U2, UQ = U1 + U2;
  const int Ti = p.Ti, Td = p.Td, KW = p.kernel, PL = (KW - 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const int TiP = up4(Ti);
  float* dz = smem;                       // [G]
  float* dvec = dz + G;                   // [CT + A]  grad wrt [ctx_{t-1} | hstate_{t-1}]
  float* dq = dvec + up4(CT + A);         // [A]
  float* dpq = dq + up4(A);               // [UQ]
  float* pqv = dpq + up4(UQ);             // [UQ]
  float* dctx = pqv + up4(UQ);            // [CT]
  float* aprev = dctx + up4(CT);          // a_{t-1}
  float* alprev = aprev + TiP;            // alpha_{t-1}
  float* a = alprev + TiP;                // a_t
  float* al = a + TiP;                    // alpha_t
  float* a2 = al + TiP;                   // a2_t
  float* dal = a2 + TiP;                  // d alpha_t  -> reused as dw
  float* da2 = dal + TiP;                 // d a2_t  -> de2
  float* de1 = da2 + TiP;                 // d e1
  float* dac = de1 + TiP;                 // carry: grad wrt a_t from step t+1's location conv
  float* dalc = dac + TiP;                // carry: grad wrt alpha_t from step t+1's recursion
  float* fl = dalc + TiP;                 // [Ti*F]
  float* dfl = fl + up4(Ti * F);          // [Ti*F]
  float* Fs = dfl + up4(Ti * F);          // [KW*F]
  float* bFs = Fs + up4(KW * F);          // [F]
  float* dFacc = bFs + up4(F);            // [KW*F + F]
  float* partial = dFacc + up4(KW * F + F);  // [ANT*8]

  const int len = (int)p.lengths[b];
  const uint32_t seed = p.seed ? *p.seed : 0u;
  const float* keys1 = p.keys1 + (size_t)b * Ti * U1;
  const float* values1 = p.values1 + (size_t)b * Ti * V1;
  const float* keys2 = p.keys2 + (size_t)b * Ti * U2;
  const float* values2 = p.values2 + (size_t)b * Ti * V2;
  float* dkeys1 = pb.dkeys1 + (size_t)b * Ti * U1;
  float* dkeys2 = pb.dkeys2 + (size_t)b * Ti * U2;
  const int OW = A + CT;
  const float* dout = pb.dout + (size_t)b * Td * OW;

  float v1r[NQ], b1r[NQ], Ur[NQ][F], dv1a[NQ], db1a[NQ], dUa[NQ][F];
#pragma unroll
  for (int qq = 0; qq < NQ; ++qq) {
    const int d = lane + 64 * qq;
    v1r[qq] = d < U1 ? p.v1[d] : 0.f;
    b1r[qq] = d < U1 ? p.b1[d] : 0.f;
    dv1a[qq] = 0.f; db1a[qq] = 0.f;
#pragma unroll
    for (int k = 0; k < F; ++k) { Ur[qq][k] = d < U1 ? p.locU[k * U1 + d] : 0.f; dUa[qq][k] = 0.f; }
  }
  const float v2r = lane < U2 ? p.v2[lane] : 0.f;
  float dv2a = 0.f;

  for (int i = tid; i < CT + A; i += ANT) dvec[i] = 0.f;
  for (int i = tid; i < Ti; i += ANT) { dac[i] = 0.f; dalc[i] = 0.f; }
  for (int i = tid; i < KW * F; i += ANT) Fs[i] = p.locF[i];
  if (tid < F) bFs[tid] = p.locFb[tid];
  for (int i = tid; i < KW * F + F; i += ANT) dFacc[i] = 0.f;
  // rows beyond the sequence length never receive gradient
  for (int i = tid + len * U1; i < Ti * U1; i += ANT) dkeys1[i] = 0.f;
  for (int i = tid + len * U2; i < Ti * U2; i += ANT) dkeys2[i] = 0.f;
  float dc_state = 0.f, dh_state = 0.f;
  __syncthreads();

  for (int t = Td - 1; t >= 0; --t) {
    const size_t bt = (size_t)b * Td + t;
    const bool first = (t == Td - 1);
    // (a) load forward state of this step, total context gradient
    for (int i = tid; i < Ti; i += ANT) {
      aprev[i] = t > 0 ? p.a1[(bt - 1) * Ti + i] : 0.f;
      alprev[i] = t > 0 ? p.align1[(bt - 1) * Ti + i] : (i == 0 ? 1.f : 0.f);
      a[i] = p.a1[bt * Ti + i];
      al[i] = p.align1[bt * Ti + i];
      a2[i] = p.align2[bt * Ti + i];
    }
    if (tid < UQ) pqv[tid] = p.pq[bt * UQ + tid];
    if (tid < CT) {
      const float g = dout[(size_t)t * OW + A + tid] + dvec[tid];
      dctx[tid] = g;
      pb.dctx[bt * CT + tid] = g;
    }
    __syncthreads();
    // (a2) location features of this step (recomputed)
    for (int e = tid; e < Ti * F; e += ANT) {
      const int tt = e / F, k = e - tt * F;
      float s = bFs[k];
      for (int jj = 0; jj < KW; ++jj) {
        const int src = tt + jj - PL;
        if (src >= 0 && src < Ti) s += aprev[src] * Fs[jj * F + k];
      }
      fl[e] = s;
    }
    // (b) d alpha / d a2 from the contexts: one wave per memory row
    for (int tt = wave; tt < Ti; tt += AW) {
      float s1 = 0.f, s2 = 0.f;
      if (tt < len) {
        const float* vr = values1 + (size_t)tt * V1;
#pragma unroll
        for (int qq = 0; qq < NQ; ++qq) { const int cc = lane + 64 * qq; if (cc < V1) s1 += vr[cc] * dctx[cc]; }
        if (lane < V2) s2 = values2[(size_t)tt * V2 + lane] * dctx[V1 + lane];
        s1 = wave_sum(s1); s2 = wave_sum(s2);
      }
      if (lane == 0) {
        dal[tt] = s1 + dalc[tt] + (pb.dalign1 ? pb.dalign1[bt * Ti + tt] : 0.f);
        da2[tt] = s2 + (pb.dalign2 ? pb.dalign2[bt * Ti + tt] : 0.f);
      }
    }
    __syncthreads();
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
      for (int tt = lane; tt < Ti; tt += 64) de1[tt] = a[tt] * (de1[tt] - s2);
    } else if (wave == 1) {
      float s3 = 0.f;
      for (int tt = lane; tt < Ti; tt += 64) s3 += da2[tt] * a2[tt];
      s3 = wave_sum(s3);
      for (int tt = lane; tt < Ti; tt += 64) da2[tt] = a2[tt] * (da2[tt] - s3);   // d e2
    }
    __syncthreads();
    // new carry for alpha_{t-1}: d alpha_prev[s] = 0.5*dw[s] + 0.5*dw[s+1]
    for (int i = tid; i < Ti; i += ANT) dalc[i] = 0.5f * dal[i] + 0.5f * (i + 1 < Ti ? dal[i + 1] : 0.f);
    // (d) energy backward: one wave per memory row
    {
      float pqb[NQ], dpqa[NQ];
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) {
        const int d = lane + 64 * qq;
        pqb[qq] = d < U1 ? pqv[d] + b1r[qq] : 0.f;
        dpqa[qq] = 0.f;
      }
      const float pq2 = lane < U2 ? pqv[U1 + lane] : 0.f;
      float dpq2a = 0.f;
      for (int tt = wave; tt < Ti; tt += AW) {
        float dfp[F];
#pragma unroll
        for (int k = 0; k < F; ++k) dfp[k] = 0.f;
        if (tt < len) {
          const float* kr = keys1 + (size_t)tt * U1;
          float* dkr = dkeys1 + (size_t)tt * U1;
          const float de = de1[tt];
          float f[F];
#pragma unroll
          for (int k = 0; k < F; ++k) f[k] = fl[tt * F + k];
#pragma unroll
          for (int qq = 0; qq < NQ; ++qq) {
            const int d = lane + 64 * qq;
            if (d < U1) {
              float lf = 0.f;
#pragma unroll
              for (int k = 0; k < F; ++k) lf += f[k] * Ur[qq][k];
              const float th = tanhf(kr[d] + pqb[qq] + lf);
              const float g = de * v1r[qq] * (1.f - th * th);
              dpqa[qq] += g; db1a[qq] += g; dv1a[qq] += de * th;
              dkr[d] = first ? g : dkr[d] + g;
#pragma unroll
              for (int k = 0; k < F; ++k) { dUa[qq][k] += f[k] * g; dfp[k] += g * Ur[qq][k]; }
            }
          }
          if (lane < U2) {
            const float th2 = tanhf(keys2[(size_t)tt * U2 + lane] + pq2);
            const float de2 = da2[tt];
            const float g2 = de2 * v2r * (1.f - th2 * th2);
            dpq2a += g2; dv2a += de2 * th2;
            float* dk2 = dkeys2 + (size_t)tt * U2 + lane;
            *dk2 = first ? g2 : *dk2 + g2;
          }
#pragma unroll
          for (int k = 0; k < F; ++k) dfp[k] = wave_sum(dfp[k]);
        }
        if (lane == 0) {
#pragma unroll
          for (int k = 0; k < F; ++k) dfl[tt * F + k] = dfp[k];
        }
      }
      // cross-wave reduction of d pq
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) { const int d = lane + 64 * qq; if (d < U1) partial[wave * UQ + d] = dpqa[qq]; }
      if (lane < U2) partial[wave * UQ + U1 + lane] = dpq2a;
    }
    __syncthreads();
    if (tid < UQ) {
      float s = 0.f;
      for (int w = 0; w < AW; ++w) s += partial[w * UQ + tid];
      dpq[tid] = s;
      pb.dpq[bt * UQ + tid] = s;
    }
    // (e) location conv backward: carry for a_{t-1}, filter / bias gradients
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
    if (tid >= ANT / 2 && tid < ANT / 2 + KW * F) {
      const int e = tid - ANT / 2, jj = e / F, k = e - jj * F;
      float g = 0.f;
      for (int tt = 0; tt < Ti; ++tt) {
        const int src = tt + jj - PL;
        if (src >= 0 && src < Ti) g += aprev[src] * dfl[tt * F + k];
      }
      dFacc[e] += g;
    } else if (tid >= ANT - 64 && tid < ANT - 64 + F) {
      const int k = tid - (ANT - 64);
      float g = 0.f;
      for (int tt = 0; tt < Ti; ++tt) g += dfl[tt * F + k];
      dFacc[KW * F + k] += g;
    }
    __syncthreads();
    // (f) d query = dpq x Wq^T
    matvec_bf16<ANT>(dpq, pb.WqT, UQ, A, partial, dq);
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
      const float tc = tanhf(cn);
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
    // (h) gradient wrt [ctx_{t-1} | hstate_{t-1}]
    matvec_bf16<ANT>(dz, pb.WrecT, G, CT + A, partial, dvec);
    if (tid < A) dh_state = dvec[CT + tid] + dh_direct;
    __syncthreads();
  }

  // small parameter gradients
#pragma unroll
  for (int qq = 0; qq < NQ; ++qq) {
    const int d = lane + 64 * qq;
    if (d < U1) {
      atomicAdd(&pb.dv1[d], dv1a[qq]);
      atomicAdd(&pb.db1[d], db1a[qq]);
#pragma unroll
      for (int k = 0; k < F; ++k) atomicAdd(&pb.dlocU[k * U1 + d], dUa[qq][k]);
    }
  }
  if (lane < U2) atomicAdd(&pb.dv2[lane], dv2a);
  for (int i = tid; i < KW * F; i += ANT) atomicAdd(&pb.dlocF[i], dFacc[i]);
  if (tid < F) atomicAdd(&pb.dlocFb[tid], dFacc[KW * F + tid]);
}

inline size_t fwd_smem(const satt_attn_rnn_params& p, int F) {
  auto u = [](int x) { return (size_t)((x + 3) & ~3); };
  const int CT = p.V1 + p.V2;
  return sizeof(float) * (u(CT + p.A) + 4 * p.A + u(p.A) + u(p.U1 + p.U2) + 5 * u(p.Ti) + u(p.Ti * F) +
                          u(p.kernel * F) + u(F) + (size_t)ANT * 8);
}
inline size_t bwd_smem(const satt_attn_rnn_params& p, int F) {
  auto u = [](int x) { return (size_t)((x + 3) & ~3); };
  const int CT = p.V1 + p.V2;
  return sizeof(float) * (4 * p.A + u(CT + p.A) + u(p.A) + 2 * u(p.U1 + p.U2) + u(CT) + 10 * u(p.Ti) +
                          2 * u(p.Ti * F) + u(p.kernel * F) + u(F) + u(p.kernel * F + F) + (size_t)ANT * 8);
}
inline int check(const satt_attn_rnn_params& p) {
  if (p.B <= 0 || p.Td <= 0 || p.Ti <= 0) return SATT_E_BADARG;
  if (p.filters != 5) return SATT_E_UNSUPPORTED;
  if (p.U1 > 64 * NQ || p.V1 > 64 * NQ || p.U2 > 64 || p.V2 > 64) return SATT_E_UNSUPPORTED;
  if ((4 * p.A) % 8 || (p.U1 + p.U2) % 8 || (p.V1 + p.V2 + p.A) % 8 || p.A % 8) return SATT_E_UNSUPPORTED;
  if (4 * p.A > 8 * ANT || p.A > 512 || p.kernel * 5 > ANT / 2 - 64) return SATT_E_UNSUPPORTED;
  if (p.U1 + p.U2 > ANT || p.V1 + p.V2 > ANT) return SATT_E_UNSUPPORTED;
  return SATT_OK;
}

}  // namespace

extern "C" int satt_attn_rnn_fwd(const satt_attn_rnn_params* pp, void* stream) {
  if (!pp) return SATT_E_BADARG;
  int rc = check(*pp);
  if (rc) return rc;
  const size_t smem = fwd_smem(*pp, 5);
  if (smem > 160 * 1024) return SATT_E_UNSUPPORTED;
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)attn_rnn_fwd_k<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(attn_rnn_fwd_k<5>, dim3(pp->B), dim3(ANT), smem, (hipStream_t)stream, *pp);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

extern "C" int satt_attn_rnn_bwd(const satt_attn_rnn_bwd_params* pp, void* stream) {
  if (!pp) return SATT_E_BADARG;
  int rc = check(pp->f);
  if (rc) return rc;
  const size_t smem = bwd_smem(pp->f, 5);
  if (smem > 160 * 1024) return SATT_E_UNSUPPORTED;
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)attn_rnn_bwd_k<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(attn_rnn_bwd_k<5>, dim3(pp->f.B), dim3(ANT), smem, (hipStream_t)stream, *pp);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}
