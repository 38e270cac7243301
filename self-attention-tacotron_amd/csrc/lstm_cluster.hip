// Cluster form of the recurrent ZoneoutLSTM (decoder LSTM1 / LSTM2, no sequence lengths): C workgroups per
// sample, each owning H/C hidden units.  Its [H x 4H/C] slice of the recurrent weights (bf16) stays RESIDENT IN
// LDS for all T steps (H=256, C=4: 128 KB of the 160 KB), so the serial loop no longer streams 512 KB per step from
// L2; the only inter-workgroup traffic is one all-gather of H floats per step through 8-byte {tag,value} granules
// (MI355X hand-off recipe R2: the data is the flag; relaxed agent-scope 8-byte stores/loads = sc1 accesses; no
// fences).  Granule buffers are double-buffered by step parity (a workgroup can be at most one step ahead), zeroed
// by a memset node at launch, tags = step+1.  Every spin is bounded; a timeout sets a sticky error word and the
// kernel runs to completion without further waiting.  Grid = (B, C): block id = c*B + b, so with B % 8 == 0 the C
// workgroups of a sample land on the SAME XCD (observed placement: block id mod 8) and hand off through one L2 — a
// speed choice only, correctness never depends on placement.  B*C <= #CUs and > 80 KB LDS per workgroup => one
// workgroup per CU, all co-resident.
#include "common.h"

namespace {

constexpr int CNT = 512;       // threads per workgroup
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

__device__ __forceinline__ void granule_put(u64* g, uint32_t tag, float v) {
  __hip_atomic_store((gu64*)g, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ONE wave gathers n granules g[idx(i)] (i = lane, lane+64, ...; at most 4 per lane) into dst[idx(i)]
// idx(i) skips the caller's own range [own0, own0+ownn)
__device__ __forceinline__ void granule_gather(u64* g, uint32_t tag, float* dst, int n_total, int own0, int ownn,
                                               int lane, unsigned int* err_word, int* dead) {
  const int nf = n_total - ownn;      // foreign granules
  float v[4]; bool ok[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { v[q] = 0.f; ok[q] = (lane + 64 * q) >= nf; }
  if (!*dead) {
    for (unsigned spins = 0;; ++spins) {
      bool all_ok = true;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = lane + 64 * q;
        if (!ok[q]) {
          const int j = i < own0 ? i : i + ownn;
          const u64 x = __hip_atomic_load((gu64*)(g + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((uint32_t)(x >> 32) == tag) { v[q] = __uint_as_float((uint32_t)x); ok[q] = true; }
          else all_ok = false;
        }
      }
      if (__all(all_ok)) break;
      if (spins > (1u << 21)) {          // ~ seconds: give up, mark, never wait again
        if (lane == 0) { __hip_atomic_store((gu32*)err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        *dead = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = lane + 64 * q;
    if (i < nf) { const int j = i < own0 ? i : i + ownn; dst[j] = v[q]; }
  }
}

// y[n] = sum_k x[k] * Ws[k][n], Ws bf16 [K][NL] in LDS; thread (ks, cg) owns 8 columns; partial [KS][NL] in LDS
__device__ __forceinline__ void matvec_lds(const float* __restrict__ x, const uint16_t* __restrict__ Ws, int K, int NL,
                                           float* __restrict__ partial, float* __restrict__ y) {
  const int tid = threadIdx.x;
  const int CG = NL >> 3, KS = CNT / CG;
  const int cg = tid % CG, ks = tid / CG;
  if (ks < KS) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int k = ks; k < K; k += KS) {
      const uint4 w = *reinterpret_cast<const uint4*>(Ws + (size_t)k * NL + cg * 8);
      const float xv = x[k];
      acc[0] += xv * __uint_as_float(w.x << 16); acc[1] += xv * __uint_as_float(w.x & 0xFFFF0000u);
      acc[2] += xv * __uint_as_float(w.y << 16); acc[3] += xv * __uint_as_float(w.y & 0xFFFF0000u);
      acc[4] += xv * __uint_as_float(w.z << 16); acc[5] += xv * __uint_as_float(w.z & 0xFFFF0000u);
      acc[6] += xv * __uint_as_float(w.w << 16); acc[7] += xv * __uint_as_float(w.w & 0xFFFF0000u);
    }
    float4* pp = reinterpret_cast<float4*>(partial + (size_t)ks * NL + cg * 8);
    pp[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    pp[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
  __syncthreads();
  for (int n = tid; n < NL; n += CNT) {
    float s = 0.f;
    for (int q = 0; q < KS; ++q) s += partial[(size_t)q * NL + n];
    y[n] = s;
  }
  __syncthreads();
}

struct CArgs {
  const float* xg; const uint16_t* W;   // fwd: Wh [H][4H]; bwd: WhT [4H][H]
  int B, T, H, C, training;
  float zc, zh; uint32_t zct, zht; const uint32_t* seed; uint32_t sc, sh;
  float* hout; int64_t ld;               // fwd out / bwd: dhout (const)
  float *gates, *cnew, *cstate, *hstate; // fwd: outputs; bwd: inputs (hstate unused)
  float* dxg;                            // bwd out
  u64* xbuf;                             // [2][B][H] granules + error word after them
  int t0, t1;                            // time range [t0, t1) processed by this launch (chunked stream pipelining)
  float* bstate;                         // bwd only: [B][2][H] carried (dc_state, dh_state) across chunk launches
};

// LDS: Ws bf16 [H][NL] | x [H or NL] | y [NL or H] | partial [KS*NL...]
__global__ __launch_bounds__(CNT) void lstm_cluster_fwd_k(const CArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H, C = a.C, HU = H / C, NL = 4 * HU, T = a.T;
  uint16_t* Ws = reinterpret_cast<uint16_t*>(smem);            // [H][NL]
  float* hvec = smem + (size_t)H * NL / 2;                      // [H]
  float* z = hvec + H;                                          // [NL]
  float* partial = z + NL;                                      // [CNT*8]
  int& dead = *reinterpret_cast<int*>(partial + CNT * 8);       // sticky hand-off timeout flag
  const int b = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = 4 * H, u0 = c * HU;
  // stage the weight slice: local column lc = g*HU + u  <->  global column g*H + u0 + u
  for (int e = tid; e < H * (NL / 8); e += CNT) {
    const int k = e / (NL / 8), v8 = e - k * (NL / 8);
    const int lc = v8 * 8, g = lc / HU, u = lc - g * HU;      // HU % 8 == 0: 8 columns stay inside one gate
    *reinterpret_cast<uint4*>(Ws + (size_t)k * NL + lc) =
        *reinterpret_cast<const uint4*>(a.W + (size_t)k * G + g * H + u0 + u);
  }
  const size_t bT = (size_t)b * T;
  if (tid < H) hvec[tid] = a.t0 > 0 ? a.hstate[(bT + a.t0 - 1) * H + tid] : 0.f;
  if (tid == 0) dead = 0;
  const uint32_t seed = a.seed ? *a.seed : 0u;
  unsigned int* err_word = reinterpret_cast<unsigned int*>(a.xbuf + (size_t)2 * a.B * C * H);
  float cst = 0.f, hst = 0.f;
  if (a.t0 > 0 && tid < HU) { cst = a.cstate[(bT + a.t0 - 1) * H + u0 + tid]; hst = a.hstate[(bT + a.t0 - 1) * H + u0 + tid]; }
  __syncthreads();
  for (int t = a.t0; t < a.t1; ++t) {
    float xi = 0.f, xj = 0.f, xf = 0.f, xo = 0.f;
    if (tid < HU) {
      const float* xr = a.xg + (bT + t) * G + u0 + tid;
      xi = xr[0]; xj = xr[H]; xf = xr[2 * H]; xo = xr[3 * H];
    }
    matvec_lds(hvec, Ws, H, NL, partial, z);
    u64* xb = a.xbuf + ((size_t)(t & 1) * a.B + b) * H;
    if (tid < HU) {
      const int j = u0 + tid;
      const float gi = sigmoidf_(xi + z[tid]);
      const float gj = tanhf_(xj + z[HU + tid]);
      const float gf = sigmoidf_(xf + z[2 * HU + tid] + 1.0f);
      const float go = sigmoidf_(xo + z[3 * HU + tid]);
      const float cn = gf * cst + gi * gj;
      const float hn = go * tanhf_(cn);
      float* gr = a.gates + (bT + t) * G;
      gr[j] = gi; gr[H + j] = gj; gr[2 * H + j] = gf; gr[3 * H + j] = go;
      a.cnew[(bT + t) * H + j] = cn;
      a.hout[(bT + t) * a.ld + j] = hn;
      const uint32_t idx = (uint32_t)(bT + t) * (uint32_t)H + (uint32_t)j;
      if (a.training) {
        if (a.zct == 0 || satt_keep(seed, a.sc, idx, a.zct)) cst = cn;
        if (a.zht == 0 || satt_keep(seed, a.sh, idx, a.zht)) hst = hn;
      } else {
        cst = (1.f - a.zc) * cn + a.zc * cst;
        hst = (1.f - a.zh) * hn + a.zh * hst;
      }
      a.cstate[(bT + t) * H + j] = cst;
      a.hstate[(bT + t) * H + j] = hst;
      hvec[j] = hst;
      if (t + 1 < a.t1) granule_put(xb + j, (uint32_t)(t + 1), hst);
    }
    if (t + 1 < a.t1 && wave == CNT / 64 - 1)  // the last wave gathers the other workgroups' units
      granule_gather(xb, (uint32_t)(t + 1), hvec, H, u0, HU, lane, err_word, &dead);
    __syncthreads();
  }
}

__global__ __launch_bounds__(CNT) void lstm_cluster_bwd_k(const CArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H, C = a.C, HU = H / C, NL = 4 * HU, T = a.T;
  uint16_t* Ws = reinterpret_cast<uint16_t*>(smem);            // WhT slice [NL rows (own gate cols)][H]
  float* dz = smem + (size_t)NL * H / 2;                        // [NL]
  float* dhp = dz + NL;                                         // [H]  partial d h_prev from own gate columns
  float* dhf = dhp + H;                                         // [H]  gathered foreign partials (only own units used)
  float* partial = dhf + H;                                     // [CNT*8]
  int& dead = *reinterpret_cast<int*>(partial + CNT * 8);
  const int b = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = 4 * H, u0 = c * HU;
  for (int e = tid; e < NL * (H / 8); e += CNT) {
    const int lc = e / (H / 8), v8 = e - lc * (H / 8);
    const int g = lc / HU, u = lc - g * HU;
    *reinterpret_cast<uint4*>(Ws + (size_t)lc * H + v8 * 8) =
        *reinterpret_cast<const uint4*>(a.W + (size_t)(g * H + u0 + u) * H + v8 * 8);
  }
  if (tid == 0) dead = 0;
  const uint32_t seed = a.seed ? *a.seed : 0u;
  const size_t bT = (size_t)b * T;
  // granule layout for the reduce-scatter: xbuf[par][b][src c][H]  (each workgroup publishes its H partials)
  unsigned int* err_word = reinterpret_cast<unsigned int*>(a.xbuf + (size_t)2 * a.B * C * H);
  float dc_state = 0.f, dh_state = 0.f;
  if (a.t1 < T && tid < HU) {           // continue from the chunk that processed steps >= t1
    dc_state = a.bstate[((size_t)b * 2 + 0) * H + u0 + tid];
    dh_state = a.bstate[((size_t)b * 2 + 1) * H + u0 + tid];
  }
  __syncthreads();
  for (int t = a.t1 - 1; t >= a.t0; --t) {
    float dh_direct = 0.f;
    if (tid < HU) {
      const int j = u0 + tid;
      const uint32_t idx = (uint32_t)(bT + t) * (uint32_t)H + (uint32_t)j;
      float kc, kh, pc, ph;
      if (a.training) {
        kc = (a.zct == 0 || satt_keep(seed, a.sc, idx, a.zct)) ? 1.f : 0.f; pc = 1.f - kc;
        kh = (a.zht == 0 || satt_keep(seed, a.sh, idx, a.zht)) ? 1.f : 0.f; ph = 1.f - kh;
      } else {
        kc = 1.f - a.zc; pc = a.zc; kh = 1.f - a.zh; ph = a.zh;
      }
      const float* gr = a.gates + (bT + t) * G;
      const float gi = gr[j], gj = gr[H + j], gf = gr[2 * H + j], go = gr[3 * H + j];
      const float cn = a.cnew[(bT + t) * H + j];
      const float cp = t > 0 ? a.cstate[(bT + t - 1) * H + j] : 0.f;
      const float dhn = a.hout[(bT + t) * a.ld + j] + kh * dh_state;    // a.hout carries dhout here
      dh_direct = ph * dh_state;
      const float tc = tanhf_(cn);
      const float dcn = dhn * go * (1.f - tc * tc) + kc * dc_state;
      const float d_o = dhn * tc;
      const float dzi = dcn * gj * gi * (1.f - gi);
      const float dzj = dcn * gi * (1.f - gj * gj);
      const float dzf = dcn * cp * gf * (1.f - gf);
      const float dzo = d_o * go * (1.f - go);
      dc_state = dcn * gf + pc * dc_state;
      float* dr = a.dxg + (bT + t) * G;
      dr[j] = dzi; dr[H + j] = dzj; dr[2 * H + j] = dzf; dr[3 * H + j] = dzo;
      dz[tid] = dzi; dz[HU + tid] = dzj; dz[2 * HU + tid] = dzf; dz[3 * HU + tid] = dzo;
    }
    __syncthreads();
    if (t == 0) break;                                         // no earlier step needs d h_prev
    matvec_lds(dz, Ws, NL, H, partial, dhp);                    // partial d h_prev[k], all k, from own gate columns
    u64* xb = a.xbuf + (((size_t)(t & 1) * a.B + b) * C) * H;
    // publish the partials the OTHER workgroups need (units outside own range)
    if (tid < H && (tid < u0 || tid >= u0 + HU)) granule_put(xb + (size_t)c * H + tid, (uint32_t)(t + 1), dhp[tid]);
    // gather, for own units, the partials of the C-1 other workgroups: (C-1)*HU granules by the last wave
    if (wave == CNT / 64 - 1) {
      const int nf = (C - 1) * HU;
      float v[4]; bool ok[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { v[q] = 0.f; ok[q] = (lane + 64 * q) >= nf; }
      if (!dead) {
        for (unsigned spins = 0;; ++spins) {
          bool all_ok = true;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = lane + 64 * q;
            if (!ok[q]) {
              int src = i / HU; const int u = i - src * HU; if (src >= c) ++src;
              const u64 x = __hip_atomic_load((gu64*)(xb + (size_t)src * H + u0 + u), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
              if ((uint32_t)(x >> 32) == (uint32_t)(t + 1)) { v[q] = __uint_as_float((uint32_t)x); ok[q] = true; }
              else all_ok = false;
            }
          }
          if (__all(all_ok)) break;
          if (spins > (1u << 21)) {
            if (lane == 0) __hip_atomic_store((gu32*)err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dead = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      // sum the C-1 foreign partials per own unit (lanes q hold (src, u) pairs): accumulate through LDS
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = lane + 64 * q;
        if (i < nf) dhf[i] = v[q];
      }
    }
    __syncthreads();
    if (tid < HU) {
      float s = dhp[u0 + tid];
      for (int k = 0; k < C - 1; ++k) s += dhf[k * HU + tid];
      dh_state = s + dh_direct;
    }
    __syncthreads();
  }
  if (a.t0 > 0 && tid < HU) {           // hand the carried gradients to the next (earlier) chunk
    a.bstate[((size_t)b * 2 + 0) * H + u0 + tid] = dc_state;
    a.bstate[((size_t)b * 2 + 1) * H + u0 + tid] = dh_state;
  }
}

inline size_t cluster_smem(int H, int C) {
  const int NL = 4 * (H / C);
  return (size_t)H * NL * 2 + sizeof(float) * ((size_t)2 * H + NL + (size_t)CNT * 8 + 4);
}
inline int cluster_check(int B, int T, int H, int C) {
  if (B <= 0 || T <= 0 || H <= 0 || C < 2) return SATT_E_BADARG;
  if (H % C || (H / C) % 8 || H % 8 || H > CNT) return SATT_E_UNSUPPORTED;
  const int NL = 4 * (H / C);
  if (NL / 8 > CNT || (C - 1) * (H / C) > 256 || H - H / C > 256) return SATT_E_UNSUPPORTED;
  if (cluster_smem(H, C) > 160 * 1024) return SATT_E_UNSUPPORTED;
  if (B * C > 256) return SATT_E_UNSUPPORTED;      // all workgroups must be co-resident (one per CU)
  return SATT_OK;
}

}  // namespace

extern "C" int64_t satt_lstm_cluster_ws_bytes(int B, int H, int C) {
  return (int64_t)sizeof(u64) * 2 * B * C * H + 64;
}

extern "C" int satt_lstm_cluster_fwd(const float* xg, const uint16_t* Wh, int B, int T, int H, int C, int training,
                                     float zc, float zh, uint32_t zc_thresh, uint32_t zh_thresh, const uint32_t* seed,
                                     uint32_t stream_c, uint32_t stream_h, float* hout, int64_t ld_hout, float* gates,
                                     float* cnew, float* cstate, float* hstate, void* ws, int t0, int t1,
                                     void* stream) {
  int rc = cluster_check(B, T, H, C);
  if (rc) return rc;
  if (t0 < 0 || t1 > T || t0 >= t1) return SATT_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(ws, 0, (size_t)satt_lstm_cluster_ws_bytes(B, H, C), s) != hipSuccess) return SATT_E_LAUNCH;
  CArgs a;
  a.xg = xg; a.W = Wh; a.B = B; a.T = T; a.H = H; a.C = C; a.training = training; a.zc = zc; a.zh = zh;
  a.zct = zc_thresh; a.zht = zh_thresh; a.seed = seed; a.sc = stream_c; a.sh = stream_h;
  a.hout = hout; a.ld = ld_hout; a.gates = gates; a.cnew = cnew; a.cstate = cstate; a.hstate = hstate;
  a.dxg = nullptr; a.xbuf = (u64*)ws; a.t0 = t0; a.t1 = t1; a.bstate = nullptr;
  const size_t smem = cluster_smem(H, C);
  (void)hipFuncSetAttribute((const void*)lstm_cluster_fwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(lstm_cluster_fwd_k, dim3(B, C), dim3(CNT), smem, s, a);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

extern "C" int satt_lstm_cluster_bwd(const float* dhout, int64_t ld_dhout, const uint16_t* WhT, int B, int T, int H,
                                     int C, int training, float zc, float zh, uint32_t zc_thresh, uint32_t zh_thresh,
                                     const uint32_t* seed, uint32_t stream_c, uint32_t stream_h, const float* gates,
                                     const float* cnew, const float* cstate, float* dxg, void* ws, int t0, int t1,
                                     float* bstate, void* stream) {
  int rc = cluster_check(B, T, H, C);
  if (rc) return rc;
  if (t0 < 0 || t1 > T || t0 >= t1 || ((t0 > 0 || t1 < T) && !bstate)) return SATT_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(ws, 0, (size_t)satt_lstm_cluster_ws_bytes(B, H, C), s) != hipSuccess) return SATT_E_LAUNCH;
  CArgs a;
  a.xg = nullptr; a.W = WhT; a.B = B; a.T = T; a.H = H; a.C = C; a.training = training; a.zc = zc; a.zh = zh;
  a.zct = zc_thresh; a.zht = zh_thresh; a.seed = seed; a.sc = stream_c; a.sh = stream_h;
  a.hout = const_cast<float*>(dhout); a.ld = ld_dhout;
  a.gates = const_cast<float*>(gates); a.cnew = const_cast<float*>(cnew); a.cstate = const_cast<float*>(cstate);
  a.hstate = nullptr; a.dxg = dxg; a.xbuf = (u64*)ws; a.t0 = t0; a.t1 = t1; a.bstate = bstate;
  const size_t smem = cluster_smem(H, C);
  (void)hipFuncSetAttribute((const void*)lstm_cluster_bwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(lstm_cluster_bwd_k, dim3(B, C), dim3(CNT), smem, s, a);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

/* 0 if no hand-off of the last cluster launch on `ws` timed out (host-synchronous read; tests / debugging only) */
extern "C" int satt_lstm_cluster_status(const void* ws, int B, int H, int C, void* stream) {
  unsigned int v = 0;
  const char* p = (const char*)ws + sizeof(u64) * 2 * (size_t)B * C * H;
  if (hipMemcpyAsync(&v, p, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return SATT_E_LAUNCH;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return SATT_E_LAUNCH;
  return v ? SATT_E_LAUNCH : SATT_OK;
}
