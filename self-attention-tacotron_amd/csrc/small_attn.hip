// Fused scaled-dot-product self-attention for a SMALL head depth (16): the encoder block of SelfAttentionCBHGEncoder
// (reference modules/self_attention.py:45-65, :108-128; 2 heads x 16, T = padded text length, no mask - the reference's
// padding mask is off on this path).  VALU form (T > 256, the fallback; r5 added the matrix-core form below for
// T <= 256): at depth 16 a bf16 MFMA tile is mostly padding, and the block must RETURN
// its probabilities (the `alignment3..` outputs, models/models.py:397-408), so this is not flash attention: plain fp32 FMAs
// against K / V rows staged in LDS, one launch forward (QK^T -> softmax -> dropout -> PV; the probabilities are written once)
// and two launches backward (per query block: row sums + dQ; per key block: dK + dV), every gradient element with exactly one
// writer.  It replaces GEMM -> softmax -> GEMM (3 launches, two [B*H, T, T] round trips) and GEMM x 4 + softmax backward
// (5 launches) of the round-2 engine.  fp32 throughout: the same kernel serves precision f32 and bf16.
// Thread layout: 8 lanes per row (row = query in the forward / dQ kernels, key in the dK/dV kernel), lane g of a row walks the
// other axis at g, g + 8, ...; the row's 8 partial results are folded with three xor steps.
#include <algorithm>
#include "common.h"

namespace {

constexpr int HD = 16, RB = 32, G8 = 8, NT = RB * G8, LP = HD + 1;      // LDS row pitch 17: lanes g..g+7 hit distinct banks

__device__ __forceinline__ float xor8_sum(float v) {       // sum over the 8 lanes of a row (lanes differ in bits 0..2)
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));     // xor 1
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));     // xor 2
  v += swz_xor(v, 4);
  return v;
}
__device__ __forceinline__ float xor8_max(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false)));
  return fmaxf(v, swz_xor(v, 4));
}

struct SArgs {
  const float* kvq; int64_t ld;          // [B*T, ld]: K | V | Q, each D wide, head h at columns h*16
  float* p;                              // [B*H, T, T] probabilities (pre-dropout)
  float* o; int64_t ldo;                 // forward out [B*T, ldo], head h at columns h*16
  const float* dout; int64_t lddo;       // backward in
  float* dkvq; int64_t ldd;              // backward out: dK | dV | dQ
  float* rowsum;                         // [B*H, T] scratch of the backward
  int T, D, H; float scale;
  uint32_t thresh; float dscale; uint32_t stream; const uint32_t* seed;
};

// rows of one (batch, head) slice -> LDS [T][LP]
__device__ __forceinline__ void stage(float* dst, const float* src, int64_t ld, int T, int tid) {
  for (int e = tid; e < T * (HD / 4); e += NT) {
    const int t = e >> 2, c = (e & 3) * 4;
    const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)t * ld + c);
    float* d = dst + t * LP + c;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}

__global__ __launch_bounds__(NT) void small_attn_fwd_k(const SArgs a) {
  extern __shared__ float sm[];
  float* Ks = sm; float* Vs = sm + a.T * LP;
  const int tid = threadIdx.x, bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H, T = a.T;
  const float* base = a.kvq + (int64_t)b * T * a.ld + h * HD;
  stage(Ks, base, a.ld, T, tid);
  stage(Vs, base + a.D, a.ld, T, tid);
  __syncthreads();
  const int ql = tid >> 3, g = tid & 7, qi = blockIdx.y * RB + ql, qc = min(qi, T - 1);
  float q[HD];
  {
    const float4* qp = reinterpret_cast<const float4*>(base + 2 * a.D + (int64_t)qc * a.ld);
#pragma unroll
    for (int c = 0; c < 4; ++c) { const float4 v = qp[c]; q[4 * c] = v.x * a.scale; q[4 * c + 1] = v.y * a.scale; q[4 * c + 2] = v.z * a.scale; q[4 * c + 3] = v.w * a.scale; }
  }
  auto score = [&](int k) {
    const float* kr = Ks + k * LP;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s += q[d] * kr[d];
    return s;
  };
  float m = -INFINITY;
  for (int k = g; k < T; k += G8) m = fmaxf(m, score(k));
  m = xor8_max(m);
  float l = 0.f;
  for (int k = g; k < T; k += G8) l += expf(score(k) - m);
  l = xor8_sum(l);
  const float inv = 1.f / l;
  const uint32_t seed = (a.thresh && a.seed) ? *a.seed : 0u;
  float acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = 0.f;
  const int64_t prow = ((int64_t)bh * T + qc) * T;
  for (int k = g; k < T; k += G8) {
    const float pr = expf(score(k) - m) * inv;
    if (qi < T) a.p[prow + k] = pr;
    float pd = pr;
    if (a.thresh) pd = satt_keep(seed, a.stream, (uint32_t)(prow + k), a.thresh) ? pr * a.dscale : 0.f;
    const float* vr = Vs + k * LP;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] += pd * vr[d];
  }
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = xor8_sum(acc[d]);
  if (qi < T) {        // lane g writes columns 2g, 2g + 1 of the row
    float* op = a.o + (int64_t)(b * T + qi) * a.ldo + h * HD + 2 * g;
    float v0 = acc[0], v1 = acc[1];
#pragma unroll
    for (int d = 1; d < G8; ++d) { v0 = g == d ? acc[2 * d] : v0; v1 = g == d ? acc[2 * d + 1] : v1; }
    op[0] = v0; op[1] = v1;
  }
}

// backward, per query block: rowsum[q] = sum_k dP[q,k] P[q,k]  and  dQ[q] = scale * sum_k P (dP - rowsum) K[k]
__global__ __launch_bounds__(NT) void small_attn_bwd_q_k(const SArgs a) {
  extern __shared__ float sm[];
  float* Ks = sm; float* Vs = sm + a.T * LP;
  const int tid = threadIdx.x, bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H, T = a.T;
  const float* base = a.kvq + (int64_t)b * T * a.ld + h * HD;
  stage(Ks, base, a.ld, T, tid);
  stage(Vs, base + a.D, a.ld, T, tid);
  __syncthreads();
  const int ql = tid >> 3, g = tid & 7, qi = blockIdx.y * RB + ql, qc = min(qi, T - 1);
  float dO[HD];
  {
    const float4* dp = reinterpret_cast<const float4*>(a.dout + (int64_t)(b * T + qc) * a.lddo + h * HD);
#pragma unroll
    for (int c = 0; c < 4; ++c) { const float4 v = dp[c]; dO[4 * c] = v.x; dO[4 * c + 1] = v.y; dO[4 * c + 2] = v.z; dO[4 * c + 3] = v.w; }
  }
  const uint32_t seed = (a.thresh && a.seed) ? *a.seed : 0u;
  const int64_t prow = ((int64_t)bh * T + qc) * T;
  auto dprob = [&](int k) {      // d P[q,k]: through the PV product and the dropout mask
    const float* vr = Vs + k * LP;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s += dO[d] * vr[d];
    if (a.thresh) s = satt_keep(seed, a.stream, (uint32_t)(prow + k), a.thresh) ? s * a.dscale : 0.f;
    return s;
  };
  float rs = 0.f;
  for (int k = g; k < T; k += G8) rs += dprob(k) * a.p[prow + k];
  rs = xor8_sum(rs);
  float dq[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) dq[d] = 0.f;
  for (int k = g; k < T; k += G8) {
    const float ds = a.p[prow + k] * (dprob(k) - rs) * a.scale;
    const float* kr = Ks + k * LP;
#pragma unroll
    for (int d = 0; d < HD; ++d) dq[d] += ds * kr[d];
  }
#pragma unroll
  for (int d = 0; d < HD; ++d) dq[d] = xor8_sum(dq[d]);
  if (qi < T) {
    if (g == 0) a.rowsum[(int64_t)bh * T + qi] = rs;
    float* op = a.dkvq + (int64_t)(b * T + qi) * a.ldd + 2 * a.D + h * HD + 2 * g;
    float v0 = dq[0], v1 = dq[1];
#pragma unroll
    for (int d = 1; d < G8; ++d) { v0 = g == d ? dq[2 * d] : v0; v1 = g == d ? dq[2 * d + 1] : v1; }
    op[0] = v0; op[1] = v1;
  }
}

// backward, per key block: dV[k] = sum_q Pd[q,k] dO[q],  dK[k] = scale * sum_q P (dP - rowsum[q]) Q[q]
__global__ __launch_bounds__(NT) void small_attn_bwd_kv_k(const SArgs a) {
  extern __shared__ float sm[];
  float* Qs = sm; float* Os = sm + a.T * LP; float* Rs = Os + a.T * LP;       // Q rows, dO rows, rowsum
  const int tid = threadIdx.x, bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H, T = a.T;
  const float* base = a.kvq + (int64_t)b * T * a.ld + h * HD;
  stage(Qs, base + 2 * a.D, a.ld, T, tid);
  stage(Os, a.dout + (int64_t)b * T * a.lddo + h * HD, a.lddo, T, tid);
  for (int t = tid; t < T; t += NT) Rs[t] = a.rowsum[(int64_t)bh * T + t];
  __syncthreads();
  // lanes of a key sit 8 apart in THIS kernel's probability reads?  No: adjacent lanes must read adjacent KEYS of one query row
  // (coalesced P reads), so here the 32 keys are the fast index and the 8 query groups the slow one; the fold goes through LDS.
  const int kl = tid & 31, g = tid >> 5, ki = blockIdx.y * RB + kl, kc = min(ki, T - 1);
  float v[HD];
  {
    const float4* vp = reinterpret_cast<const float4*>(base + a.D + (int64_t)kc * a.ld);
#pragma unroll
    for (int c = 0; c < 4; ++c) { const float4 x = vp[c]; v[4 * c] = x.x; v[4 * c + 1] = x.y; v[4 * c + 2] = x.z; v[4 * c + 3] = x.w; }
  }
  const uint32_t seed = (a.thresh && a.seed) ? *a.seed : 0u;
  float dk[HD], dv[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
  for (int qi = g; qi < T; qi += G8) {
    const int64_t pidx = ((int64_t)bh * T + qi) * T + kc;
    const float pr = a.p[pidx];
    const float* orow = Os + qi * LP;
    const float* qrow = Qs + qi * LP;
    float dpd = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) dpd += orow[d] * v[d];
    float pd = pr, dp = dpd;
    if (a.thresh) {
      const bool keep = satt_keep(seed, a.stream, (uint32_t)pidx, a.thresh);
      pd = keep ? pr * a.dscale : 0.f; dp = keep ? dpd * a.dscale : 0.f;
    }
    const float ds = pr * (dp - Rs[qi]) * a.scale;
#pragma unroll
    for (int d = 0; d < HD; ++d) { dv[d] += pd * orow[d]; dk[d] += ds * qrow[d]; }
  }
  __syncthreads();                 // Qs / Os are dead: reuse the LDS for the fold over the 8 query groups
  float* red = sm;                 // [8][32][2 * HD + 1]
  constexpr int RP = 2 * HD + 1;
  {
    float* r = red + (g * RB + kl) * RP;
#pragma unroll
    for (int d = 0; d < HD; ++d) { r[d] = dk[d]; r[HD + d] = dv[d]; }
  }
  __syncthreads();
  for (int e = tid; e < RB * 2 * HD; e += NT) {
    const int k = e / (2 * HD), d = e - k * 2 * HD;
    float s = 0.f;
#pragma unroll
    for (int gg = 0; gg < G8; ++gg) s += red[(gg * RB + k) * RP + d];
    const int kk = blockIdx.y * RB + k;
    if (kk < T) a.dkvq[(int64_t)(b * T + kk) * a.ldd + (d < HD ? 0 : a.D) + h * HD + (d & (HD - 1))] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// r5: the same block on the MATRIX cores - v_mfma_f32_16x16x4_f32 (fp32 operands, fp32 accumulation: the results stay at the fp32
// level the tests hold the VALU kernels to; a bf16 tile at depth 16 would need a 3-way operand split to get there).  One WAVE owns
// 16 rows (queries in the forward / dQ kernel, keys in the dK/dV kernel) against ALL T positions of the other axis, everything in
// registers: no LDS, no barrier, MWPB independent waves per workgroup.
//   S^T tile  C[key 4g+r][query c] = sum_d K[key c'][d] Q[query c][d]     A = K rows (lane (g, c'): K[16t + c'][4g + i], one 16-byte load),
//                                                                        B = Q rows (lane (g, c): Q[q0 + c][4g + i]);  the reduction index of
//                                                                        step i, slot g is d = 4g + i - any enumeration both operands share
//   O^T       C[d 4g+r][query c]   = sum_key V[key][d] Pd[query c][key]   step r, slot g <-> key 16t + 4g + r: element r of the S^T tile IS the
//                                                                        B operand (no transpose, no LDS), A = V[16t + 4g + r][c]
// and the backward kernels use the same two shapes with (V, dO), (K, dS^T) and - per key tile - (dO, V), (Q, dS), (dO, Pd).
// lane = 16 g + c.  NTL = tiles of 16 along the other axis (T <= 16 NTL), a compile-time bound of the register arrays.
constexpr int MWPB = 4;      // waves per workgroup (independent 16-row tiles); 2 measured slower (11.4 -> 13.7 us forward: more, emptier workgroups)
template <int NTL>
__global__ __launch_bounds__(64 * MWPB) void small_attn_mfma_fwd_k(const SArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H, T = a.T;
  const int q0 = (blockIdx.y * MWPB + wave) * 16;
  if (q0 >= T) return;                                    // (wave-uniform; the kernel has no barrier)
  const float* base = a.kvq + (int64_t)b * T * a.ld + h * HD;
  const int qi = q0 + c, qc = min(qi, T - 1), nt = (T + 15) >> 4;
  float4 qv = *reinterpret_cast<const float4*>(base + 2 * a.D + (int64_t)qc * a.ld + 4 * g);
  qv.x *= a.scale; qv.y *= a.scale; qv.z *= a.scale; qv.w *= a.scale;
  f32x4_t s[NTL];
  float4 kv[NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t) kv[t] = *reinterpret_cast<const float4*>(base + (int64_t)min(16 * t + c, T - 1) * a.ld + 4 * g);
  // value rows of the PV product: requested now, consumed after the softmax
  float vr[NTL][4];
#pragma unroll
  for (int t = 0; t < NTL; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) vr[t][r] = base[a.D + (int64_t)min(16 * t + 4 * g + r, T - 1) * a.ld + c];
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[t].x, qv.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[t].y, qv.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[t].z, qv.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[t].w, qv.w, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = t < nt && 16 * t + 4 * g + r < T;
      acc[r] = ok ? acc[r] : -INFINITY;
      m = fmaxf(m, acc[r]);
    }
    s[t] = acc;
  }
  // the statistics of query c live in the four lane groups: fold over lanes l ^ 16, l ^ 32
  m = fmaxf(m, swz_xor(m, 16));
  m = fmaxf(m, __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(m))));
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < NTL; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float e = expf(s[t][r] - m); s[t][r] = e; l += e; }      // masked keys: exp(-inf) = 0
  l += swz_xor(l, 16);
  l += __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(l)));
  const float inv = 1.f / l;
  const uint32_t seed = (a.thresh && a.seed) ? *a.seed : 0u;
  const int64_t prow = ((int64_t)bh * T + qc) * T;
  f32x4_t o = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    if (t < nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * t + 4 * g + r;
        const float pr = s[t][r] * inv;
        if (qi < T && key < T) a.p[prow + key] = pr;
        float pd = pr;
        if (a.thresh) pd = satt_keep(seed, a.stream, (uint32_t)(prow + min(key, T - 1)), a.thresh) ? pr * a.dscale : 0.f;
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[t][r], pd, o, 0, 0, 0);
      }
    }
  }
  if (qi < T) *reinterpret_cast<float4*>(a.o + (int64_t)(b * T + qi) * a.ldo + h * HD + 4 * g) = make_float4(o[0], o[1], o[2], o[3]);
}

// backward, per query tile: rowsum[q] = sum_k dP[q,k] P[q,k]  and  dQ[q] = scale * sum_k P (dP - rowsum) K[k]
template <int NTL>
__global__ __launch_bounds__(64 * MWPB) void small_attn_mfma_bwd_q_k(const SArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H, T = a.T;
  const int q0 = (blockIdx.y * MWPB + wave) * 16;
  if (q0 >= T) return;
  const float* base = a.kvq + (int64_t)b * T * a.ld + h * HD;
  const int qi = q0 + c, qc = min(qi, T - 1), nt = (T + 15) >> 4;
  const float4 dov = *reinterpret_cast<const float4*>(a.dout + (int64_t)(b * T + qc) * a.lddo + h * HD + 4 * g);
  const uint32_t seed = (a.thresh && a.seed) ? *a.seed : 0u;
  const int64_t prow = ((int64_t)bh * T + qc) * T;
  f32x4_t dp[NTL], pv[NTL];
  float4 vv[NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t) vv[t] = *reinterpret_cast<const float4*>(base + a.D + (int64_t)min(16 * t + c, T - 1) * a.ld + 4 * g);
#pragma unroll
  for (int t = 0; t < NTL; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int key = 16 * t + 4 * g + r; pv[t][r] = a.p[prow + min(key, T - 1)]; }
  float kr[NTL][4];
#pragma unroll
  for (int t = 0; t < NTL; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) kr[t][r] = base[(int64_t)min(16 * t + 4 * g + r, T - 1) * a.ld + c];
  float rs = 0.f;
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[t].x, dov.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[t].y, dov.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[t].z, dov.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[t].w, dov.w, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * t + 4 * g + r;
      const bool ok = t < nt && key < T;
      float d = acc[r];
      if (a.thresh) d = satt_keep(seed, a.stream, (uint32_t)(prow + min(key, T - 1)), a.thresh) ? d * a.dscale : 0.f;
      pv[t][r] = ok ? pv[t][r] : 0.f;
      dp[t][r] = d;
      rs += d * pv[t][r];
    }
  }
  rs += swz_xor(rs, 16);
  rs += __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __float_as_int(rs)));
  f32x4_t dq = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    if (t < nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ds = pv[t][r] * (dp[t][r] - rs) * a.scale;
        dq = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[t][r], ds, dq, 0, 0, 0);
      }
    }
  }
  if (qi < T) {
    if (g == 0) a.rowsum[(int64_t)bh * T + qi] = rs;
    *reinterpret_cast<float4*>(a.dkvq + (int64_t)(b * T + qi) * a.ldd + 2 * a.D + h * HD + 4 * g) = make_float4(dq[0], dq[1], dq[2], dq[3]);
  }
}

// backward, per key tile: dV[k] = sum_q Pd[q,k] dO[q],  dK[k] = scale * sum_q P (dP - rowsum[q]) Q[q]
//   dP tile  C[query 4g+r][key c] = sum_d dO[query c'][d] V[key c][d]        A = dO rows (16-byte loads), B = the wave's V rows
//   dK^T     C[d 4g+r][key c]     = sum_query Q[query][d] dS[query][key c]    step r, slot g <-> query 16t + 4g + r;  dV^T with (dO, Pd)
template <int NTL>
__global__ __launch_bounds__(64 * MWPB) void small_attn_mfma_bwd_kv_k(const SArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H, T = a.T;
  const int k0 = (blockIdx.y * MWPB + wave) * 16;
  if (k0 >= T) return;
  const float* base = a.kvq + (int64_t)b * T * a.ld + h * HD;
  const float* dob = a.dout + (int64_t)b * T * a.lddo + h * HD;
  const int ki = k0 + c, kc = min(ki, T - 1);
  const float4 vv = *reinterpret_cast<const float4*>(base + a.D + (int64_t)kc * a.ld + 4 * g);
  const uint32_t seed = (a.thresh && a.seed) ? *a.seed : 0u;
  f32x4_t dk = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dv = dk;
#pragma unroll
  for (int t = 0; t < NTL; ++t) {      // (no tile guard: rows beyond T are clamped loads with zero weight - the loads of all tiles can fly together)
    const float4 da = *reinterpret_cast<const float4*>(dob + (int64_t)min(16 * t + c, T - 1) * a.lddo + 4 * g);
    float pr[4], rsv[4], qr[4], dor[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = min(16 * t + 4 * g + r, T - 1);
      pr[r] = a.p[((int64_t)bh * T + q) * T + kc];
      rsv[r] = a.rowsum[(int64_t)bh * T + q];
      qr[r] = base[2 * a.D + (int64_t)q * a.ld + c];
      dor[r] = dob[(int64_t)q * a.lddo + c];
    }
    f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(da.x, vv.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(da.y, vv.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(da.z, vv.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(da.w, vv.w, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = 16 * t + 4 * g + r;
      const bool ok = q < T && ki < T;
      const float p1 = ok ? pr[r] : 0.f;
      float pd = p1, d = acc[r];
      if (a.thresh) {
        const bool keep = satt_keep(seed, a.stream, (uint32_t)(((int64_t)bh * T + min(q, T - 1)) * T + kc), a.thresh);
        pd = keep ? p1 * a.dscale : 0.f; d = keep ? d * a.dscale : 0.f;
      }
      const float ds = p1 * (d - rsv[r]) * a.scale;
      dk = __builtin_amdgcn_mfma_f32_16x16x4f32(qr[r], ds, dk, 0, 0, 0);
      dv = __builtin_amdgcn_mfma_f32_16x16x4f32(dor[r], pd, dv, 0, 0, 0);
    }
  }
  if (ki < T) {
    float* dst = a.dkvq + (int64_t)(b * T + ki) * a.ldd + h * HD + 4 * g;
    *reinterpret_cast<float4*>(dst) = make_float4(dk[0], dk[1], dk[2], dk[3]);
    *reinterpret_cast<float4*>(dst + a.D) = make_float4(dv[0], dv[1], dv[2], dv[3]);
  }
}
constexpr int MFMA_MAX_T = 256;
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline int check(const float* kvq, int64_t ld, int B, int T, int D, int H) {
  if (!kvq || B <= 0 || T <= 0 || H <= 0 || D != H * HD) return SATT_E_UNSUPPORTED;
  if ((ld & 3) || (reinterpret_cast<uintptr_t>(kvq) & 15)) return SATT_E_BADARG;
  const size_t need = sizeof(float) * std::max((size_t)(2 * T * LP + T), (size_t)G8 * RB * (2 * HD + 1));
  if (need > 160 * 1024) return SATT_E_UNSUPPORTED;
  return SATT_OK;
}

}  // namespace

extern "C" int satt_small_attn_supported(int head_dim, int T) {
  return head_dim == HD && T > 0 && sizeof(float) * (size_t)(2 * T * LP + T) <= 160 * 1024;
}

extern "C" int satt_small_attn_fwd(const float* kvq, int64_t ld, float* p, float* o, int64_t ldo, int B, int T, int D, int H,
                                   float scale, uint32_t drop_thresh, float drop_scale, uint32_t drop_stream,
                                   const uint32_t* seed, void* stream) {
  int rc = check(kvq, ld, B, T, D, H);
  if (rc) return rc;
  if (!p || !o || (ldo & 1)) return SATT_E_BADARG;
  SArgs a{};
  a.kvq = kvq; a.ld = ld; a.p = p; a.o = o; a.ldo = ldo; a.T = T; a.D = D; a.H = H; a.scale = scale;
  a.thresh = drop_thresh; a.dscale = drop_scale; a.stream = drop_stream; a.seed = seed;
  if (T <= MFMA_MAX_T && !(ldo & 3) && al16(o)) {          // matrix-core form (16 query rows per wave, 4 waves per workgroup)
    const dim3 grid(B * H, (T + 16 * MWPB - 1) / (16 * MWPB));
    if (T <= 64) hipLaunchKernelGGL(small_attn_mfma_fwd_k<4>, grid, dim3(64 * MWPB), 0, (hipStream_t)stream, a);
    else if (T <= 160) hipLaunchKernelGGL(small_attn_mfma_fwd_k<10>, grid, dim3(64 * MWPB), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(small_attn_mfma_fwd_k<16>, grid, dim3(64 * MWPB), 0, (hipStream_t)stream, a);
    SATT_LAUNCH_CHECK();
    return SATT_OK;
  }
  const size_t smem = sizeof(float) * 2 * T * LP;
  (void)hipFuncSetAttribute((const void*)small_attn_fwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(small_attn_fwd_k, dim3(B * H, (T + RB - 1) / RB), dim3(NT), smem, (hipStream_t)stream, a);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

extern "C" int satt_small_attn_bwd(const float* kvq, int64_t ld, const float* p, const float* dout, int64_t lddo, float* dkvq,
                                   int64_t ldd, float* rowsum, int B, int T, int D, int H, float scale, uint32_t drop_thresh,
                                   float drop_scale, uint32_t drop_stream, const uint32_t* seed, void* stream) {
  int rc = check(kvq, ld, B, T, D, H);
  if (rc) return rc;
  if (!p || !dout || !dkvq || !rowsum || (lddo & 3) || (ldd & 1) || (reinterpret_cast<uintptr_t>(dout) & 15)) return SATT_E_BADARG;
  SArgs a{};
  a.kvq = kvq; a.ld = ld; a.p = const_cast<float*>(p); a.dout = dout; a.lddo = lddo; a.dkvq = dkvq; a.ldd = ldd; a.rowsum = rowsum;
  a.T = T; a.D = D; a.H = H; a.scale = scale;
  a.thresh = drop_thresh; a.dscale = drop_scale; a.stream = drop_stream; a.seed = seed;
  if (T <= MFMA_MAX_T && !(ldd & 3) && al16(dkvq)) {
    const dim3 gm(B * H, (T + 16 * MWPB - 1) / (16 * MWPB));
    hipStream_t st = (hipStream_t)stream;
    if (T <= 64) { hipLaunchKernelGGL(small_attn_mfma_bwd_q_k<4>, gm, dim3(64 * MWPB), 0, st, a); }
    else if (T <= 160) { hipLaunchKernelGGL(small_attn_mfma_bwd_q_k<10>, gm, dim3(64 * MWPB), 0, st, a); }
    else { hipLaunchKernelGGL(small_attn_mfma_bwd_q_k<16>, gm, dim3(64 * MWPB), 0, st, a); }
    if (T <= 64) { hipLaunchKernelGGL(small_attn_mfma_bwd_kv_k<4>, gm, dim3(64 * MWPB), 0, st, a); }
    else if (T <= 160) { hipLaunchKernelGGL(small_attn_mfma_bwd_kv_k<10>, gm, dim3(64 * MWPB), 0, st, a); }
    else { hipLaunchKernelGGL(small_attn_mfma_bwd_kv_k<16>, gm, dim3(64 * MWPB), 0, st, a); }
    SATT_LAUNCH_CHECK();
    return SATT_OK;
  }
  const dim3 grid(B * H, (T + RB - 1) / RB);
  const size_t smq = sizeof(float) * 2 * T * LP;
  const size_t smk = sizeof(float) * std::max((size_t)(2 * T * LP + T), (size_t)G8 * RB * (2 * HD + 1));
  (void)hipFuncSetAttribute((const void*)small_attn_bwd_q_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smq);
  (void)hipFuncSetAttribute((const void*)small_attn_bwd_kv_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smk);
  hipLaunchKernelGGL(small_attn_bwd_q_k, grid, dim3(NT), smq, (hipStream_t)stream, a);
  hipLaunchKernelGGL(small_attn_bwd_kv_k, grid, dim3(NT), smk, (hipStream_t)stream, a);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}
