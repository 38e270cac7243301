// Cluster form of the recurrent ZoneoutLSTM (decoder LSTM1 / LSTM2, no sequence lengths): C workgroups per
// sample, each owning HU = H/C hidden units, i.e. the NL = 4*HU gate columns of those units.  The member's
// [H x NL] slice of the recurrent weights (bf16) is REGISTER-resident for the whole launch (H = 256, C = 4: 16 MFMA
// B operands = 64 accumulation registers per lane), the recurrence is v_mfma_f32_16x16x32_bf16 on the exactly
// split fp32 state (mfma_rec.h), and nothing is re-read from L2 inside the time loop.
//   forward : z_own = h_{t-1} x Wh[:, own]  ->  cell for the own units  ->  all-gather of h_state (H granules)
//   backward: cell backward for the own units -> partial d h_prev[all H] = dz_own x Wh[:, own]^T (K-split: the
//             SAME slice, transposed) -> reduce-scatter: every member gathers the C-1 foreign partials of its units
// Exchange: 8-byte {tag, value} granules (cluster_xchg.h), double-buffered by step parity (a member can be at most
// one step ahead), zeroed by a memset at launch, tag = step+1; plain L2-resident stores when the XCC-id handshake
// shows that the cluster shares an XCD (grid (B, C): block id = c*B + b, so B % 8 == 0 puts the C members of a sample
// on one XCD), agent-scope stores otherwise.  Every spin is bounded; a timeout sets a sticky error word and the
// kernel runs to completion without further waiting.  B*C <= #CUs: all members are co-resident.
#include "mfma_rec.h"
#include "cluster_xchg.h"
#ifdef SATT_CLUSTER_JITTER
#define lds_barrier() do { lds_barrier(); cluster_jitter(); } while (0)
#endif

namespace {

constexpr int CNT = 512;       // threads per workgroup (XW waves)
constexpr int LKT = 8;         // K tiles of the forward slice (H <= 256)

struct CArgs {
  const float* xg; const uint16_t* W;   // recurrent weights in register order (lstm_cluster_pack_k): fwd / bwd pack
  int B, T, H, C, training;
  float zc, zh; uint32_t zct, zht; const uint32_t* seed; uint32_t sc, sh;
  float* hout; int64_t ld;               // fwd out / bwd: dhout (const)
  float *gates, *cnew, *cstate, *hstate; // fwd: outputs; bwd: inputs (hstate unused)
  float* dxg;                            // bwd out
  u64* xbuf;                             // granules: [2][B][C][H] | [B][C] XCC ids | error word
  int t0, t1;                            // time range [t0, t1) processed by this launch (chunked stream pipelining)
  float* bstate;                         // bwd only: [B][2][H] carried (dc_state, dh_state) across chunk launches
  // fwd only, optional (x != nullptr): the input projection of the chunk is formed by the kernel itself (see the prologue)
  const float* x; int64_t ldx; int Kin; const uint16_t* Win; const float* bin;
};
constexpr int XKT = 17;        // K tiles of the fused input projection at most: Kin <= 544 (LSTM1 of the decoder: [h_att | ctx1 | ctx2])

// are all members of this sample's cluster on one XCD?  (granules with a tag no step can produce)
// `xtag` is unique per launch of a pass (the workspace is zeroed by the first launch of a pass only)
__device__ __forceinline__ bool cluster_same_xcd(u64* xi, int C, int c, unsigned int* err_word, int* dead, int* flag,
                                                 uint32_t xtag) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef SATT_XCHG_DEBUG
  if (tid == 0) atomicAdd(err_word + 12, 1u);
#endif
  if (tid == 0) gput(xi + c, xtag, __int_as_float(xcc_id()), false);
  if (wave == 0) {
    int mism = 0;
    const int mine = xcc_id();
    gather_span(xi, C, xtag, 0, 1, lane, [&](int, float v) { if (__float_as_int(v) != mine) mism = 1; }, err_word, dead);
    mism = __any(mism) || *dead;
    if (lane == 0) *flag = mism ? 0 : 1;
  }
  __syncthreads();
  // word 1 behind the error word: workgroup-launches of this pass that exchange with plain same-XCD stores
  if (tid == 0) atomicAdd(err_word + (*flag != 0 ? 1 : 2), 1u);
  return *flag != 0;
}

// forward: wave w owns the local gate columns [32w, 32w+32) (2 N tiles) x LKT K tiles = 16 B operands
#ifdef SATT_LSTM_FWD_PIN      // (r6 experiment, tools/build_variant.sh pin lstm_cluster.hip -DSATT_LSTM_FWD_PIN: 128 registers - two workgroups per
__attribute__((amdgpu_waves_per_eu(4, 4)))      // CU as the backward kernel, 20 spilled registers, +3.6 % per step - what SATT_LSTM_CU_CHARGE=packed needs;
#endif                                          // not the default: DESIGN.md 1, "Residency")
__global__ __launch_bounds__(CNT) void lstm_cluster_fwd_k(const CArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t hs[4 * (LKT * 32 + APAD)];   // bf16 [4][HS]: split h_state, row 3 = 0
  __shared__ float z[256];
  __shared__ int flags[4];
  constexpr int HS = LKT * 32 + APAD;
  const int H = a.H, C = a.C, HU = H / C, NL = 4 * HU, T = a.T, G = 4 * H;
  const int b = blockIdx.x, c = blockIdx.y, u0 = c * HU;
  const size_t bT = (size_t)b * T;
  const uint32_t seed = a.seed ? *a.seed : 0u;
  u64* xi = a.xbuf + (size_t)2 * a.B * C * H + (size_t)b * C;
  unsigned int* err_word = reinterpret_cast<unsigned int*>(a.xbuf + (size_t)2 * a.B * C * H + (size_t)a.B * C);
  int* dead = &flags[0];
  // B operand (kt, j): rows kt*32 + (l>>4)*8 .. +8 of local column (wave*2 + j)*16 + (l&15) = g*HU + u, pre-packed in
  // exactly this order (one 16-byte load per operand; gathering the 2-byte elements here cost ~10 us per launch)
  i32x4_t w[LKT][2];
  {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const i32x4_t* pw = reinterpret_cast<const i32x4_t*>(a.W) + ((size_t)(c * XW + wave) * LKT * 2) * 64 + lane;
    // every load first, then the pins: a pin needs its value, i.e. pin-after-each-load is one L2 round trip per operand
    // (16 serial trips in front of every chunk launch)
    i32x4_t tmpw[LKT * 2];
#pragma unroll
    for (int q = 0; q < LKT * 2; ++q) tmpw[q] = pw[q * 64];
#pragma unroll
    for (int kt = 0; kt < LKT; ++kt)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        i32x4_t t = tmpw[kt * 2 + j];
        asm volatile("" : "+a"(t));
        w[kt][j] = t;
      }
    for (int i = tid; i < 4 * HS; i += CNT) hs[i] = 0;
    if (tid == 0) *dead = 0;
    __syncthreads();
    if (a.t0 > 0 && tid < H) xs_put(hs, HS, tid, a.hstate[(bT + a.t0 - 1) * H + tid]);
  }
  // FUSED INPUT PROJECTION (r4, the short chunks at the end of the forward pipeline): xg[b, t, own gate columns] = x[b, t, :] W_in[:,
  // own columns] + bias for the steps of this launch, 16 steps per pass: the rows go to LDS as a bf16 A image, the member's W_in slice
  // streams from L2 in MFMA B-operand order (satt_lstm_cluster_pack_in: 2 N tiles x KT K tiles per wave, as the recurrent slice),
  // results are written to the xg buffer the step loop reads (L2 hits a moment later).  Same roundings and the same accumulation
  // order as the GEMM it replaces (csrc/gemm_tile.hip: bf16 operands, one MFMA per 32-wide K step into an fp32 accumulator, bias
  // behind it): bit-identical - but ~4 us inside this launch instead of a 9 - 16 us launch of its own in the chain behind the
  // attention kernel's last steps.
  if (a.x) {
    __shared__ __attribute__((aligned(16))) uint16_t xa[16 * (XKT * 32 + APAD)];
    constexpr int XS = XKT * 32 + APAD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Kin = a.Kin, KT = (Kin + 31) / 32;
    const i32x4_t* pwin = reinterpret_cast<const i32x4_t*>(a.Win) + ((size_t)(c * XW + wave) * KT * 2) * 64 + lane;
    for (int m0 = a.t0; m0 < a.t1; m0 += 16) {
      // rows m0 .. m0+15 (clamped to the chunk) -> bf16 [16][XS], zero beyond Kin
      for (int e = tid; e < 16 * (KT * 8); e += CNT) {
        const int r = e / (KT * 8), k = (e - r * (KT * 8)) * 4;
        const float* src = a.x + (bT + min(m0 + r, a.t1 - 1)) * a.ldx + min(k, Kin - 4);
        float4 v = *reinterpret_cast<const float4*>(src);
        if (k >= Kin) v = make_float4(0.f, 0.f, 0.f, 0.f);
        uint2 wv; wv.x = pack_bf16x2(v.x, v.y); wv.y = pack_bf16x2(v.z, v.w);
        *reinterpret_cast<uint2*>(xa + r * XS + k) = wv;
      }
      __syncthreads();
      f32x4_t acc0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      const uint16_t* arow = xa + (lane & 15) * XS + (lane >> 4) * 8;
      constexpr int KB = 6;                          // K tiles per batch of operand loads (12 x 16 bytes per lane in flight)
      for (int k0 = 0; k0 < KT; k0 += KB) {
        i32x4_t bw[KB][2];
#pragma unroll
        for (int q = 0; q < KB; ++q) {
          const int kt = min(k0 + q, KT - 1);
          bw[q][0] = pwin[(size_t)(kt * 2) * 64]; bw[q][1] = pwin[(size_t)(kt * 2 + 1) * 64];
        }
#pragma unroll
        for (int q = 0; q < KB; ++q) {
          if (k0 + q < KT) {
            const bf16x8_t av = *reinterpret_cast<const bf16x8_t*>(arow + (k0 + q) * 32);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8_t, bw[q][0]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8_t, bw[q][1]), acc1, 0, 0, 0);
          }
        }
      }
      // D[row = 4 (l >> 4) + r][col = l & 15] of N tile j: local column (2 wave + j) 16 + (l & 15) = g HU + u
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int lc = (wave * 2 + j) * 16 + (lane & 15), g = lc / HU, u = lc - g * HU;
        if (lc < NL) {
          const int col = g * H + u0 + u;
          const float bj = a.bin ? a.bin[col] : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int t = m0 + 4 * (lane >> 4) + r;
            if (t < a.t1) const_cast<float*>(a.xg)[(bT + t) * G + col] = (j == 0 ? acc0[r] : acc1[r]) + bj;
          }
        }
      }
      __syncthreads();                               // (the A image is rewritten by the next pass)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own rows of xg have reached L2 before any thread of the workgroup reads them
    __syncthreads();
  }
  const bool same_xcd = cluster_same_xcd(xi, C, c, err_word, dead, &flags[1], 0xFFFF0000u | (uint32_t)(a.t0 & 0xFFFF));
  float cst = 0.f, hst = 0.f;
  if (a.t0 > 0 && threadIdx.x < HU) {
    cst = a.cstate[(bT + a.t0 - 1) * H + u0 + threadIdx.x]; hst = a.hstate[(bT + a.t0 - 1) * H + u0 + threadIdx.x];
  }
  float nxg[4];
  {
    const float* xr = a.xg + (bT + a.t0) * G + u0 + min((int)threadIdx.x, HU - 1);
    nxg[0] = xr[0]; nxg[1] = xr[H]; nxg[2] = xr[2 * H]; nxg[3] = xr[3 * H];
  }
  __syncthreads();
  for (int t = a.t0; t < a.t1; ++t) {
    int oz = 0;
    asm volatile("" : "+v"(oz));                    // keeps index arithmetic inside the step (see attn_cluster.hip)
    const int tid = (int)threadIdx.x + oz, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // input contributions of the own units, requested one step ahead (branch-free, clamped; see attn_cluster_fwd_k)
    const float xi_ = nxg[0], xj = nxg[1], xf = nxg[2], xo = nxg[3];
    {
      const float* xr = a.xg + (bT + min(t + 1, a.t1 - 1)) * G + u0 + min(tid, HU - 1);
      nxg[0] = xr[0]; nxg[1] = xr[H]; nxg[2] = xr[2 * H]; nxg[3] = xr[3 * H];
    }
    {
      f32x4_t acc0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      const uint16_t* hrow = hs + min(lane & 15, 3) * HS + (lane >> 4) * 8;
      // r5: every A operand is requested before the first block (an LDS read cannot move across an asm block: with the reads
      // inside the K loop each of the four blocks waited for its own LDS round trip).  The forward kernel runs one workgroup
      // per CU either way (144 registers before): the 32 operand registers cost no occupancy.
      bf16x8_t av[LKT];
#pragma unroll
      for (int kt = 0; kt < LKT; ++kt) av[kt] = *reinterpret_cast<const bf16x8_t*>(hrow + kt * 32);
#pragma unroll
      for (int kt = 0; kt < LKT; kt += 2) {
        // one chain: operand cover in front of the first block, result cover after the last (mfma_rec.h)
        if (kt == 0) mfma22_a<false>(acc0, acc1, av[kt], av[kt + 1], w[kt][0], w[kt][1], w[kt + 1][0], w[kt + 1][1]);
        else if (kt + 2 < LKT) mfma22_a<false, false>(acc0, acc1, av[kt], av[kt + 1], w[kt][0], w[kt][1], w[kt + 1][0], w[kt + 1][1]);
        else mfma22_a<true, false>(acc0, acc1, av[kt], av[kt + 1], w[kt][0], w[kt][1], w[kt + 1][0], w[kt + 1][1]);
      }
      if (lane < 16) {
        z[wave * 32 + lane] = acc0[0] + acc0[1] + acc0[2];
        z[wave * 32 + 16 + lane] = acc1[0] + acc1[1] + acc1[2];
      }
    }
    lds_barrier();
    u64* xb = a.xbuf + ((size_t)(t & 1) * a.B + b) * H;
    const uint32_t tag = (uint32_t)(t + 1);
    if (tid < HU) {
      const int j = u0 + tid;
      const float gi = sigmoidf_(xi_ + z[tid]);
      const float gj = tanhf_(xj + z[HU + tid]);
      const float gf = sigmoidf_(xf + z[2 * HU + tid] + 1.0f);
      const float go = sigmoidf_(xo + z[3 * HU + tid]);
      const float cn = gf * cst + gi * gj;
      const float hn = go * tanhf_(cn);
      const uint32_t idx = (uint32_t)(bT + t) * (uint32_t)H + (uint32_t)j;
      if (a.training) {
        if (a.zct == 0 || satt_keep(seed, a.sc, idx, a.zct)) cst = cn;
        if (a.zht == 0 || satt_keep(seed, a.sh, idx, a.zht)) hst = hn;
      } else {
        cst = (1.f - a.zc) * cn + a.zc * cst;
        hst = (1.f - a.zh) * hn + a.zh * hst;
      }
      if (t + 1 < a.t1) gput(xb + j, tag, hst, same_xcd);
      float* gr = a.gates + (bT + t) * G;
      gr[j] = gi; gr[H + j] = gj; gr[2 * H + j] = gf; gr[3 * H + j] = go;
      a.cnew[(bT + t) * H + j] = cn;
      a.hout[(bT + t) * a.ld + j] = hn;
      a.cstate[(bT + t) * H + j] = cst;
      a.hstate[(bT + t) * H + j] = hst;
    }
    // all-gather of h_state (own units included: one code path), H <= 384 granules: an even share for each of the last four waves
    if (t + 1 < a.t1 && wave >= XW - 4)
      gather_span(xb, H, tag, wave - (XW - 4), 4, lane, [&](int i, float v) { xs_put(hs, HS, i, v); }, err_word, dead);
    lds_barrier();
  }
#ifdef SATT_XCHG_DEBUG
  if (threadIdx.x == 0) atomicAdd(err_word + 13, 1u);
#endif
}

#ifndef SATT_LSTMC_ZAHEAD
#define SATT_LSTMC_ZAHEAD 8      // all LKT tiles: still 128 registers, no spills (tools/kernel_regs.py)
#endif
// backward: wave w owns the output units [32w, 32w+32) (2 N tiles) x ALL K tiles of the own gate columns (NL <= 256) = 16 B
// operands and accumulates over K inside the MFMA accumulators: the partial d h_prev of a unit leaves the registers of lanes
// 0..15 straight into the exchange (or the own-unit buffer).  Until r4 the waves owned one K tile each and the eight per-tile
// partials crossed LDS, a barrier and an 8-way sum first.
// (4 waves per SIMD = two workgroups per CU: LSTM1 and LSTM2 of the layer pipeline share CUs; 128 registers, checked with
// tools/kernel_regs.py - without the bound the allocator drifted to 140 when the polling loop moved into poll_until)
__attribute__((amdgpu_waves_per_eu(4, 4)))
__global__ __launch_bounds__(CNT) void lstm_cluster_bwd_k(const CArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t dzs[4 * (256 + APAD)];       // bf16 [4][256 + APAD]: split own dz, row 3 = 0
  __shared__ float dhf[64 * GQ];                                        // gathered foreign partials of the own units
  __shared__ float dhown[64];
  __shared__ int flags[4];
  constexpr int DZS = 256 + APAD;
  const int H = a.H, C = a.C, HU = H / C, NL = 4 * HU, T = a.T, G = 4 * H;
  const int b = blockIdx.x, c = blockIdx.y, u0 = c * HU;
  const size_t bT = (size_t)b * T;
  const uint32_t seed = a.seed ? *a.seed : 0u;
  u64* xi = a.xbuf + (size_t)2 * a.B * C * H + (size_t)b * C;
  unsigned int* err_word = reinterpret_cast<unsigned int*>(a.xbuf + (size_t)2 * a.B * C * H + (size_t)a.B * C);
  int* dead = &flags[0];
  // B operand (kt, j): own gate columns (local order g*HU + u) kt*32 + (l>>4)*8 .. +8 of hidden unit (2 wave + j)*16 + (l&15);
  // the pack is [c][kt][nt][lane][8] (lstm_cluster_pack_k), zero beyond NL / H
  i32x4_t w[LKT][2];
  {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const i32x4_t* pw = reinterpret_cast<const i32x4_t*>(a.W) + ((size_t)(c * XW) * 16 + 2 * wave) * 64 + lane;
    i32x4_t tmpw[LKT * 2];                  // every load first, then the pins (see the forward kernel)
#pragma unroll
    for (int q = 0; q < LKT * 2; ++q) tmpw[q] = pw[((q >> 1) * 16 + (q & 1)) * 64];
#pragma unroll
    for (int kt = 0; kt < LKT; ++kt)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        i32x4_t t = tmpw[kt * 2 + j];
        asm volatile("" : "+a"(t));
        w[kt][j] = t;
      }
    for (int i = tid; i < 4 * DZS; i += CNT) dzs[i] = 0;
    if (tid == 0) *dead = 0;
    __syncthreads();
  }
  const bool same_xcd = cluster_same_xcd(xi, C, c, err_word, dead, &flags[1], 0xFFFF0000u | (uint32_t)(a.t1 & 0xFFFF));
  float dc_state = 0.f, dh_state = 0.f;
  if (a.t1 < T && threadIdx.x < HU) {           // continue from the chunk that processed steps >= t1
    dc_state = a.bstate[((size_t)b * 2 + 0) * H + u0 + threadIdx.x];
    dh_state = a.bstate[((size_t)b * 2 + 1) * H + u0 + threadIdx.x];
  }
  float pg[4] = {0.f, 0.f, 0.f, 0.f}, pcn = 0.f, pcp = 0.f, pdh = 0.f;
  // saved tensors of the step processed next, requested one step ahead: branch-free (clamped step / unit) and issued BEHIND the
  // last use of the current values - loads inside a branch, or into registers whose old values are still live, are waited for
  // at the loop's back edge (s_waitcnt vmcnt(0) in front of the copies), i.e. never overlap anything (see lstm.hip)
  auto prefetch = [&](int t, int tid) {
    const int j = u0 + min(tid, HU - 1), tc = max(t, a.t0);
    const float* gr = a.gates + (bT + tc) * G;
    pg[0] = gr[j]; pg[1] = gr[H + j]; pg[2] = gr[2 * H + j]; pg[3] = gr[3 * H + j];
    pcn = a.cnew[(bT + tc) * H + j];
    pcp = a.cstate[(bT + max(tc - 1, 0)) * H + j];
    pdh = a.hout[(bT + tc) * a.ld + j];          // a.hout carries dhout here
  };
  prefetch(a.t1 - 1, threadIdx.x);
  __syncthreads();
  for (int t = a.t1 - 1; t >= a.t0; --t) {
    int oz = 0;
    asm volatile("" : "+v"(oz));
    const int tid = (int)threadIdx.x + oz, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float gi = pg[0], gj = pg[1], gf = pg[2], go = pg[3], cn = pcn, cp = t > 0 ? pcp : 0.f, dho = pdh;
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
      const float dhn = dho + kh * dh_state;
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
      xs_put(dzs, DZS, tid, dzi); xs_put(dzs, DZS, HU + tid, dzj);
      xs_put(dzs, DZS, 2 * HU + tid, dzf); xs_put(dzs, DZS, 3 * HU + tid, dzo);
    }
    prefetch(t - 1, tid);
    if (t == 0) break;                                        // no earlier step needs d h_prev
    lds_barrier();
    // granule layout for the reduce-scatter: xbuf[par][b][src c][H]: every member publishes the partials the OTHERS need
    u64* xb = a.xbuf + (((size_t)(t & 1) * a.B + b) * C) * H;
    const uint32_t tag = (uint32_t)(t + 1);
    if (wave * 32 < H) {
      f32x4_t acc0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      const uint16_t* zrow = dzs + min(lane & 15, 3) * DZS + (lane >> 4) * 8;
      // r5: the A operands run ZAHEAD tiles ahead of the chain (an LDS read cannot move across an asm block: next to their block,
      // every pair of K tiles waited for its own LDS round trip); no deeper: the kernel must stay within 128 registers
      constexpr int ZAHEAD = SATT_LSTMC_ZAHEAD;
      bf16x8_t za[LKT];
#pragma unroll
      for (int kt = 0; kt < ZAHEAD; ++kt) za[kt] = *reinterpret_cast<const bf16x8_t*>(zrow + kt * 32);
#pragma unroll
      for (int kt = 0; kt < LKT; kt += 2) {
        // one chain: operand cover in front of the first block, result cover after the last (mfma_rec.h)
        if (kt == 0) mfma22_a<false>(acc0, acc1, za[kt], za[kt + 1], w[kt][0], w[kt][1], w[kt + 1][0], w[kt + 1][1]);
        else if (kt + 2 < LKT) mfma22_a<false, false>(acc0, acc1, za[kt], za[kt + 1], w[kt][0], w[kt][1], w[kt + 1][0], w[kt + 1][1]);
        else mfma22_a<true, false>(acc0, acc1, za[kt], za[kt + 1], w[kt][0], w[kt][1], w[kt + 1][0], w[kt + 1][1]);
        if (kt + ZAHEAD < LKT) {
          za[kt + ZAHEAD] = *reinterpret_cast<const bf16x8_t*>(zrow + (kt + ZAHEAD) * 32);
          za[kt + ZAHEAD + 1] = *reinterpret_cast<const bf16x8_t*>(zrow + (kt + ZAHEAD + 1) * 32);
        }
      }
      if (lane < 16) {
        const float s0 = acc0[0] + acc0[1] + acc0[2], s1 = acc1[0] + acc1[1] + acc1[2];
        const int n0 = wave * 32 + lane, n1 = n0 + 16;
        if (n0 < H) {
          if (n0 >= u0 && n0 < u0 + HU) dhown[n0 - u0] = s0;
          else gput(xb + (size_t)c * H + n0, tag, s0, same_xcd);
        }
        if (n1 < H) {
          if (n1 >= u0 && n1 < u0 + HU) dhown[n1 - u0] = s1;
          else gput(xb + (size_t)c * H + n1, tag, s1, same_xcd);
        }
      }
    }
    // gather, for own units, the partials of the C-1 other members: (C-1)*HU <= 384 granules, an even share for each of the
    // last three waves, every load of a poll in flight before the first tag is inspected (cluster_xchg.h; until r4 one wave
    // polled them with a branch per granule: (C-1) serial L2 round trips per attempt)
    if (wave >= XW - 3) {
      const int nf = (C - 1) * HU;
      gather_span_map(xb, nf, [&](int i) { int src = i / HU; const int u = i - src * HU; if (src >= c) ++src; return src * H + u0 + u; },
                      tag, wave - (XW - 3), 3, lane,
                      [&](int ph, float v) { const int src = ph / H, u = ph - src * H - u0; dhf[(src > c ? src - 1 : src) * HU + u] = v; },
                      err_word, dead);
    }
    lds_barrier();
    if (tid < HU) {
      float s = dhown[tid];
      for (int k = 0; k < C - 1; ++k) s += dhf[k * HU + tid];
      dh_state = s + dh_direct;
    }
  }
  if (a.t0 > 0 && threadIdx.x < HU) {           // hand the carried gradients to the next (earlier) chunk
    a.bstate[((size_t)b * 2 + 0) * H + u0 + threadIdx.x] = dc_state;
    a.bstate[((size_t)b * 2 + 1) * H + u0 + threadIdx.x] = dh_state;
  }
#ifdef SATT_XCHG_DEBUG
  if (threadIdx.x == 0) atomicAdd(err_word + 13, 1u);
#endif
}

// Recurrent weights [H][4H] fp32 -> the two register-order bf16 packs of the cluster kernels, 8 elements (16 bytes)
// per thread: pf[c][wave][kt][j][lane][8] (forward B operands), pb[c][wave][nt][lane][8] (backward, transposed slice)
constexpr int PACK_GROUPS = XW * 16 * 64;      // 16-byte groups per member and direction (LKT * 2 == 16)
__global__ void lstm_cluster_pack_k(const float* __restrict__ W, int64_t ld, int H, int C, uint16_t* __restrict__ pf,
                                    uint16_t* __restrict__ pb) {
  const int HU = H / C, NL = 4 * HU;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= 2 * C * PACK_GROUPS) return;
  const bool bwd = gid >= C * PACK_GROUPS;
  const int e = bwd ? gid - C * PACK_GROUPS : gid;
  const int lane = e & 63, op = (e >> 6) & 15, wave = (e >> 10) & (XW - 1), c = e >> 13;
  uint16_t v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float x = 0.f;
    if (!bwd) {
      const int kt = op >> 1, j = op & 1;
      const int n = (wave * 2 + j) * 16 + (lane & 15), g = n / HU, u = n - g * HU, k = kt * 32 + (lane >> 4) * 8 + i;
      if (k < H && n < NL) x = W[(int64_t)k * ld + g * H + c * HU + u];
    } else {
      const int n = op * 16 + (lane & 15), k = wave * 32 + (lane >> 4) * 8 + i, g = k / HU, u = k - g * HU;
      if (k < NL && n < H) x = W[(int64_t)n * ld + g * H + c * HU + u];
    }
    v[i] = f2bf(x);
  }
  uint16_t* dst = (bwd ? pb : pf) + (size_t)e * 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[i] = v[i];
}

// Input weights [K][4H] fp32 -> the register-order bf16 pack of the fused input projection: [c][wave][kt][j][lane][8], KT = ceil(K / 32)
__global__ void lstm_cluster_pack_in_k(const float* __restrict__ W, int64_t ld, int K, int H, int C, uint16_t* __restrict__ pin) {
  const int HU = H / C, NL = 4 * HU, KT = (K + 31) / 32;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= C * XW * KT * 2 * 64) return;
  const int lane = gid & 63, j = (gid >> 6) & 1, rest = gid >> 7, kt = rest % KT, wave = (rest / KT) % XW, c = rest / (KT * XW);
  const int n = (wave * 2 + j) * 16 + (lane & 15), g = n / HU, u = n - g * HU;
  uint16_t* dst = pin + (size_t)gid * 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = kt * 32 + (lane >> 4) * 8 + i;
    dst[i] = f2bf((k < K && n < NL) ? W[(int64_t)k * ld + g * H + c * HU + u] : 0.f);
  }
}

inline int cluster_check(int B, int T, int H, int C) {
  if (B <= 0 || T <= 0 || H <= 0 || C < 2) return SATT_E_BADARG;
  if (H % C || (H / C) % 8 || H % 8) return SATT_E_UNSUPPORTED;
  const int HU = H / C, NL = 4 * HU;
  if (H > 32 * LKT || NL > 256 || HU > 64) return SATT_E_UNSUPPORTED;       // register-resident slice
  if ((C - 1) * HU > 64 * GQ || H > 64 * GQ) return SATT_E_UNSUPPORTED;     // single-wave gathers
  if (B * C > 1024) return SATT_E_UNSUPPORTED;     // (host-only sanity bound; the launchers compare with the device's resident capacity)
  return SATT_OK;
}

}  // namespace

extern "C" int64_t satt_lstm_cluster_ws_bytes(int B, int H, int C) {
  return (int64_t)sizeof(u64) * ((int64_t)2 * B * C * H + (int64_t)B * C) + 64;
}

/* SATT_OK if the cluster LSTM kernels accept (B, T, H) with C workgroups per sample (host-only check) */
extern "C" int satt_lstm_cluster_check(int B, int T, int H, int C) { return cluster_check(B, T, H, C); }

extern "C" int64_t satt_lstm_cluster_pack_elems(int C) { return (int64_t)C * PACK_GROUPS * 8; }

extern "C" int satt_lstm_cluster_pack(const float* Wh, int64_t ld, int H, int C, uint16_t* pack_fwd, uint16_t* pack_bwd,
                                      void* stream) {
  int rc = cluster_check(1, 1, H, C);
  if (rc) return rc;
  if (!Wh || !pack_fwd || !pack_bwd || ld < 4 * (int64_t)H) return SATT_E_BADARG;
  if ((reinterpret_cast<uintptr_t>(pack_fwd) | reinterpret_cast<uintptr_t>(pack_bwd)) & 15) return SATT_E_BADARG;
  const int n = 2 * C * PACK_GROUPS;
  hipLaunchKernelGGL(lstm_cluster_pack_k, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, Wh, ld, H, C,
                     pack_fwd, pack_bwd);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

extern "C" int64_t satt_lstm_cluster_pack_in_elems(int K, int C) { return (int64_t)C * XW * ((K + 31) / 32) * 2 * 64 * 8; }

extern "C" int satt_lstm_cluster_pack_in(const float* Win, int64_t ld, int K, int H, int C, uint16_t* pack, void* stream) {
  int rc = cluster_check(1, 1, H, C);
  if (rc) return rc;
  if (!Win || !pack || K <= 0 || ld < 4 * (int64_t)H || (reinterpret_cast<uintptr_t>(pack) & 15)) return SATT_E_BADARG;
  if (K > 32 * XKT || K % 4) return SATT_E_UNSUPPORTED;
  const int n = C * XW * ((K + 31) / 32) * 2 * 64;
  hipLaunchKernelGGL(lstm_cluster_pack_in_k, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, Win, ld, K, H, C, pack);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

static int lstm_cluster_fwd_impl(const float* xg, const uint16_t* Wh, int B, int T, int H, int C, int training,
                                 float zc, float zh, uint32_t zc_thresh, uint32_t zh_thresh, const uint32_t* seed,
                                 uint32_t stream_c, uint32_t stream_h, float* hout, int64_t ld_hout, float* gates,
                                 float* cnew, float* cstate, float* hstate, void* ws, int t0, int t1, const float* x, int64_t ldx,
                                 int Kin, const uint16_t* Win, const float* bin, void* stream);

extern "C" int satt_lstm_cluster_fwd(const float* xg, const uint16_t* Wh, int B, int T, int H, int C, int training,
                                     float zc, float zh, uint32_t zc_thresh, uint32_t zh_thresh, const uint32_t* seed,
                                     uint32_t stream_c, uint32_t stream_h, float* hout, int64_t ld_hout, float* gates,
                                     float* cnew, float* cstate, float* hstate, void* ws, int t0, int t1,
                                     void* stream) {
  return lstm_cluster_fwd_impl(xg, Wh, B, T, H, C, training, zc, zh, zc_thresh, zh_thresh, seed, stream_c, stream_h, hout, ld_hout,
                               gates, cnew, cstate, hstate, ws, t0, t1, nullptr, 0, 0, nullptr, nullptr, stream);
}

extern "C" int satt_lstm_cluster_fwd_x(float* xg, const uint16_t* Wh, int B, int T, int H, int C, int training,
                                       float zc, float zh, uint32_t zc_thresh, uint32_t zh_thresh, const uint32_t* seed,
                                       uint32_t stream_c, uint32_t stream_h, float* hout, int64_t ld_hout, float* gates,
                                       float* cnew, float* cstate, float* hstate, void* ws, int t0, int t1, const float* x,
                                       int64_t ldx, int Kin, const uint16_t* Win, const float* bin, void* stream) {
  if (!x || !Win || Kin <= 0 || Kin > 32 * XKT || Kin % 4 || ldx % 4 || ldx < Kin) return SATT_E_BADARG;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(Win)) & 15) return SATT_E_BADARG;
  return lstm_cluster_fwd_impl(xg, Wh, B, T, H, C, training, zc, zh, zc_thresh, zh_thresh, seed, stream_c, stream_h, hout, ld_hout,
                               gates, cnew, cstate, hstate, ws, t0, t1, x, ldx, Kin, Win, bin, stream);
}

static int lstm_cluster_fwd_impl(const float* xg, const uint16_t* Wh, int B, int T, int H, int C, int training,
                                 float zc, float zh, uint32_t zc_thresh, uint32_t zh_thresh, const uint32_t* seed,
                                 uint32_t stream_c, uint32_t stream_h, float* hout, int64_t ld_hout, float* gates,
                                 float* cnew, float* cstate, float* hstate, void* ws, int t0, int t1, const float* x, int64_t ldx,
                                 int Kin, const uint16_t* Win, const float* bin, void* stream) {
  int rc = cluster_check(B, T, H, C);
  if (rc) return rc;
  if (t0 < 0 || t1 > T || t0 >= t1 || (reinterpret_cast<uintptr_t>(Wh) & 15)) return SATT_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  // the first launch of a pass zeroes the workspace; the later chunk launches of the pass continue on it (step tags
  // t+1 and the placement handshake tag are unique per launch within a pass)
  // (the granules only: the 64-byte tail - error word, exchange-path counters - is sticky until the caller zeroes it)
  if (t0 == 0 && hipMemsetAsync(ws, 0, (size_t)satt_lstm_cluster_ws_bytes(B, H, C) - 64, s) != hipSuccess) return SATT_E_LAUNCH;
  CArgs a;
  a.xg = xg; a.W = Wh; a.B = B; a.T = T; a.H = H; a.C = C; a.training = training; a.zc = zc; a.zh = zh;
  a.zct = zc_thresh; a.zht = zh_thresh; a.seed = seed; a.sc = stream_c; a.sh = stream_h;
  a.hout = hout; a.ld = ld_hout; a.gates = gates; a.cnew = cnew; a.cstate = cstate; a.hstate = hstate;
  a.dxg = nullptr; a.xbuf = (u64*)ws; a.t0 = t0; a.t1 = t1; a.bstate = nullptr;
  a.x = x; a.ldx = ldx; a.Kin = Kin; a.Win = Win; a.bin = bin;
  const int cap = cluster_capacity((const void*)lstm_cluster_fwd_k, CNT, 0);
  if (cap >= 0 && B * C > cap) return SATT_E_UNSUPPORTED;          // not every member could be resident (cluster_xchg.h)
  hipLaunchKernelGGL(lstm_cluster_fwd_k, dim3(B, C), dim3(CNT), 0, s, a);
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
  if (reinterpret_cast<uintptr_t>(WhT) & 15) return SATT_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (t1 == T && hipMemsetAsync(ws, 0, (size_t)satt_lstm_cluster_ws_bytes(B, H, C) - 64, s) != hipSuccess) return SATT_E_LAUNCH;
  CArgs a;
  a.xg = nullptr; a.W = WhT; a.B = B; a.T = T; a.H = H; a.C = C; a.training = training; a.zc = zc; a.zh = zh;
  a.zct = zc_thresh; a.zht = zh_thresh; a.seed = seed; a.sc = stream_c; a.sh = stream_h;
  a.hout = const_cast<float*>(dhout); a.ld = ld_dhout;
  a.gates = const_cast<float*>(gates); a.cnew = const_cast<float*>(cnew); a.cstate = const_cast<float*>(cstate);
  a.hstate = nullptr; a.dxg = dxg; a.xbuf = (u64*)ws; a.t0 = t0; a.t1 = t1; a.bstate = bstate;
  a.x = nullptr; a.ldx = 0; a.Kin = 0; a.Win = nullptr; a.bin = nullptr;
  const int cap = cluster_capacity((const void*)lstm_cluster_bwd_k, CNT, 0);
  if (cap >= 0 && B * C > cap) return SATT_E_UNSUPPORTED;
  hipLaunchKernelGGL(lstm_cluster_bwd_k, dim3(B, C), dim3(CNT), 0, s, a);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

/* Resident footprint of a cluster LSTM launch (needs a device; see satt_attn_cluster_residency): *workgroups = B*C, *per_cu =
 * workgroups of the kernel one CU can hold (occupancy calculator), *cus = CUs of the current device */
extern "C" int satt_lstm_cluster_residency(int B, int T, int H, int C, int backward, int* workgroups, int* per_cu, int* cus) {
  if (!workgroups || !per_cu || !cus) return SATT_E_BADARG;
  int rc = cluster_check(B, T, H, C);
  if (rc) return rc;
  const void* fn = backward ? (const void*)lstm_cluster_bwd_k : (const void*)lstm_cluster_fwd_k;
  if (cluster_capacity(fn, CNT, 0, per_cu, cus) < 0) return SATT_E_LAUNCH;
  *workgroups = B * C;
  return SATT_OK;
}

/* 0 if no hand-off of any cluster launch on `ws` timed out since the caller zeroed it (host-synchronous read; the error
 * word is sticky - launches clear the granules only - and satt_adam_step reads it on the device) */
extern "C" int satt_lstm_cluster_status(const void* ws, int B, int H, int C, void* stream) {
  unsigned int v = 0;
  const char* p = (const char*)ws + satt_lstm_cluster_ws_bytes(B, H, C) - 64;
  if (hipMemcpyAsync(&v, p, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return SATT_E_LAUNCH;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return SATT_E_LAUNCH;
  return v ? SATT_E_LAUNCH : SATT_OK;
}

/* host-synchronous (tests): *count = workgroup-launches since the caller zeroed the workspace that found their whole cluster
 * on one XCD and exchanged with plain stores; *slow (optional) = those that took the write-through path */
extern "C" int satt_lstm_cluster_fastpath(const void* ws, int B, int H, int C, void* stream, int* count, int* slow) {
  if (!ws || !count) return SATT_E_BADARG;
  unsigned int v[2] = {0, 0};
  const char* p = (const char*)ws + satt_lstm_cluster_ws_bytes(B, H, C) - 64 + 4;
  if (hipMemcpyAsync(v, p, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return SATT_E_LAUNCH;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return SATT_E_LAUNCH;
  *count = (int)v[0];
  if (slow) *slow = (int)v[1];
  return SATT_OK;
}
