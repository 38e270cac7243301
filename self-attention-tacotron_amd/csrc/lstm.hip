// Persistent recurrent ZoneoutLSTM kernels: one 1024-thread workgroup per (sample, direction) walks all time
// steps; no inter-workgroup communication (samples are independent).  The input projection is hoisted into a
// batched MFMA GEMM (satt_gemm); only the h-recurrence [H]x[H,4H] runs here, with bf16 weights streamed from L2
// and fp32 state.  Latency-bound: per step cost ~ (bytes of W_h) / (per-CU L2 bandwidth).
// H <= 128 (the encoder BiLSTM) takes the register-resident form: the whole [H, 4H] bf16 recurrent matrix lives in
// the accumulation registers of a 512-thread workgroup (64 per lane) and the recurrence is 16 MFMAs per wave and
// step on the exactly split fp32 state (mfma_rec.h); nothing is re-read from L2 inside the time loop.
#include "matvec.h"
#include "mfma_rec.h"

namespace {

#ifdef SATT_LSTM_PROF      // per-phase shader-clock sums of workgroup (0, 0), wave 0 (tools/lstm_time.py with a -DSATT_LSTM_PROF variant)
static __device__ unsigned long long satt_lstm_prof[8];
#define LPROF(i) do { if (prof_on) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pacc[i] += n_ - plast; plast = n_; } } while (0)
#else
#define LPROF(i)
#endif
constexpr int LNT = 1024;
constexpr int MNT = 512;          // threads of the register-resident kernels
constexpr int MW = MNT / 64;      // waves
constexpr int MH = 128;           // largest H of the register-resident kernels: 4 K tiles x 32 N tiles

struct LstmArgs {
  const float* xg; const uint16_t* Wh; const int64_t* lengths;
  int B, T, H, training;
  float zc, zh; uint32_t zct, zht; const uint32_t* seed; uint32_t sc[2], sh[2];
  float* hout; int64_t ld;
  float *gates, *cnew, *cstate, *hstate;
};

__global__ __launch_bounds__(LNT) void lstm_fwd_k(const LstmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H, G = 4 * a.H, T = a.T;
  float* hvec = smem;            // [H]
  float* z = hvec + H;           // [4H]
  float* partial = z + G;        // [LNT*8]
  const int b = blockIdx.x, d = blockIdx.y, j = threadIdx.x;
  const int len = a.lengths ? (int)a.lengths[b] : T;
  const bool rev = (d == 1);
  const size_t dirBT = ((size_t)d * a.B + b) * T;
  const float* xg = a.xg + dirBT * G;
  const uint16_t* Wh = a.Wh + (size_t)d * H * G;
  float* gates = a.gates + dirBT * G;
  float* cnew = a.cnew + dirBT * H;
  float* cstate = a.cstate + dirBT * H;
  float* hstate = a.hstate + dirBT * H;
  float* hout = a.hout + (size_t)b * T * a.ld + (size_t)d * H;
  const uint32_t seed = a.seed ? *a.seed : 0u;
  float c = 0.f, h = 0.f;
  if (j < H) hvec[j] = 0.f;
  __syncthreads();
  for (int s = 0; s < len; ++s) {
    const int t = rev ? (len - 1 - s) : s;
    float xi = 0.f, xj = 0.f, xf = 0.f, xo = 0.f;
    if (j < H) {
      const float* xr = xg + (size_t)t * G;
      xi = xr[j]; xj = xr[H + j]; xf = xr[2 * H + j]; xo = xr[3 * H + j];
    }
    matvec_bf16<LNT, 4>(hvec, Wh, H, G, partial, z);
    if (j < H) {
      const float gi = sigmoidf_(xi + z[j]);
      const float gj = tanhf_(xj + z[H + j]);
      const float gf = sigmoidf_(xf + z[2 * H + j] + 1.0f);
      const float go = sigmoidf_(xo + z[3 * H + j]);
      const float cn = gf * c + gi * gj;
      const float hn = go * tanhf_(cn);
      float* gr = gates + (size_t)t * G;
      gr[j] = gi; gr[H + j] = gj; gr[2 * H + j] = gf; gr[3 * H + j] = go;
      cnew[(size_t)t * H + j] = cn;
      hout[(size_t)t * a.ld + j] = hn;
      const uint32_t idx = ((uint32_t)b * (uint32_t)T + (uint32_t)t) * (uint32_t)H + (uint32_t)j;
      if (a.training) {
        if (a.zct == 0 || satt_keep(seed, a.sc[d], idx, a.zct)) c = cn;
        if (a.zht == 0 || satt_keep(seed, a.sh[d], idx, a.zht)) h = hn;
      } else {
        c = (1.f - a.zc) * cn + a.zc * c;
        h = (1.f - a.zh) * hn + a.zh * h;
      }
      cstate[(size_t)t * H + j] = c;
      hstate[(size_t)t * H + j] = h;
      hvec[j] = h;
    }
    __syncthreads();
  }
  // beyond the sequence length: outputs and saved tensors are zero (dynamic_rnn zero-output semantics)
  for (int t = len; t < T; ++t) {
    if (j < H) {
      hout[(size_t)t * a.ld + j] = 0.f;
      cnew[(size_t)t * H + j] = 0.f; cstate[(size_t)t * H + j] = 0.f; hstate[(size_t)t * H + j] = 0.f;
    }
    for (int n = j; n < G; n += LNT) gates[(size_t)t * G + n] = 0.f;
  }
}

struct LstmBwdArgs {
  const float* dhout; int64_t ld; const uint16_t* WhT; const int64_t* lengths;
  int B, T, H, training;
  float zc, zh; uint32_t zct, zht; const uint32_t* seed; uint32_t sc[2], sh[2];
  const float *gates, *cnew, *cstate;
  float* dxg;
};

__global__ __launch_bounds__(LNT) void lstm_bwd_k(const LstmBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H, G = 4 * a.H, T = a.T;
  float* dz = smem;              // [4H]
  float* dhv = dz + G;           // [H]
  float* partial = dhv + H;      // [LNT*8]
  const int b = blockIdx.x, d = blockIdx.y, j = threadIdx.x;
  const int len = a.lengths ? (int)a.lengths[b] : T;
  const bool rev = (d == 1);
  const size_t dirBT = ((size_t)d * a.B + b) * T;
  const uint16_t* WhT = a.WhT + (size_t)d * G * H;
  const float* gates = a.gates + dirBT * G;
  const float* cnew = a.cnew + dirBT * H;
  const float* cstate = a.cstate + dirBT * H;
  const float* dhout = a.dhout + (size_t)b * T * a.ld + (size_t)d * H;
  float* dxg = a.dxg + dirBT * G;
  const uint32_t seed = a.seed ? *a.seed : 0u;
  float dc_state = 0.f, dh_state = 0.f;  // gradients wrt the carried (post-zoneout) state
  for (int s = len - 1; s >= 0; --s) {
    const int t = rev ? (len - 1 - s) : s;
    float dh_direct = 0.f;
    if (j < H) {
      const uint32_t idx = ((uint32_t)b * (uint32_t)T + (uint32_t)t) * (uint32_t)H + (uint32_t)j;
      float kc, kh, pc, ph;  // d(state)/d(new), d(state)/d(prev)
      if (a.training) {
        kc = (a.zct == 0 || satt_keep(seed, a.sc[d], idx, a.zct)) ? 1.f : 0.f; pc = 1.f - kc;
        kh = (a.zht == 0 || satt_keep(seed, a.sh[d], idx, a.zht)) ? 1.f : 0.f; ph = 1.f - kh;
      } else {
        kc = 1.f - a.zc; pc = a.zc; kh = 1.f - a.zh; ph = a.zh;
      }
      const float* gr = gates + (size_t)t * G;
      const float gi = gr[j], gj = gr[H + j], gf = gr[2 * H + j], go = gr[3 * H + j];
      const float cn = cnew[(size_t)t * H + j];
      const int tp = rev ? t + 1 : t - 1;
      const float cp = (s > 0) ? cstate[(size_t)tp * H + j] : 0.f;
      const float dhn = dhout[(size_t)t * a.ld + j] + kh * dh_state;
      dh_direct = ph * dh_state;
      const float tc = tanhf_(cn);
      const float dcn = dhn * go * (1.f - tc * tc) + kc * dc_state;
      const float d_o = dhn * tc;
      const float dzi = dcn * gj * gi * (1.f - gi);
      const float dzj = dcn * gi * (1.f - gj * gj);
      const float dzf = dcn * cp * gf * (1.f - gf);
      const float dzo = d_o * go * (1.f - go);
      dc_state = dcn * gf + pc * dc_state;
      float* dr = dxg + (size_t)t * G;
      dr[j] = dzi; dr[H + j] = dzj; dr[2 * H + j] = dzf; dr[3 * H + j] = dzo;
      dz[j] = dzi; dz[H + j] = dzj; dz[2 * H + j] = dzf; dz[3 * H + j] = dzo;
    }
    __syncthreads();
    matvec_bf16<LNT, 4>(dz, WhT, G, H, partial, dhv);
    if (j < H) dh_state = dhv[j] + dh_direct;
    __syncthreads();
  }
  for (int t = len; t < T; ++t)
    for (int n = j; n < G; n += LNT) dxg[(size_t)t * G + n] = 0.f;
}

// ---- register-resident forms (H <= MH) -------------------------------------------------------------------
// forward: wave w owns the FOUR gate columns of the hidden units [16w, 16w+16) (N tile nt = gate nt of those units) x 4 K tiles
// = 16 B operands: lanes 0..15 of the wave end the MFMA chain holding the four gate pre-activations of "their" unit, so the
// cell runs right there (16 lanes of each of the 8 waves) and the step needs ONE barrier - behind the publication of the new
// h rows, which are double-buffered (a fast wave writes the image of step s+1 while a slow one still reads that of step s).
// Until r4 the waves owned 64 consecutive gate columns, the pre-activations crossed LDS and a second barrier to reach the
// cell threads (waves 0-1): 0.97 -> see profiles/r04_encoder_lstm.txt per step.
constexpr int LPD = 4;     // steps of saved / input rows in flight in the MFMA kernels' step loops (even: the LDS images alternate)
// (row strides of the split A images carry APAD, mfma_rec.h)

__global__ __launch_bounds__(MNT) void lstm_fwd_mfma_k(const LstmArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t hs[2][4 * (MH + APAD)];  // bf16 [parity][4][MH + APAD]: split h_state, row 3 = 0
  const int H = a.H, G = 4 * a.H, T = a.T;
  constexpr int HSS = MH + APAD;
  const int b = blockIdx.x, d = blockIdx.y;
  const int len = a.lengths ? (int)a.lengths[b] : T;
  const bool rev = (d == 1);
  const size_t dirBT = ((size_t)d * a.B + b) * T;
  const float* xg = a.xg + dirBT * G;
  const uint16_t* Wh = a.Wh + (size_t)d * H * G;
  float* gates = a.gates + dirBT * G;
  float* cnew = a.cnew + dirBT * H;
  float* cstate = a.cstate + dirBT * H;
  float* hstate = a.hstate + dirBT * H;
  float* hout = a.hout + (size_t)b * T * a.ld + (size_t)d * H;
  const uint32_t seed = a.seed ? *a.seed : 0u;
  i32x4_t w[4][4];                 // [kt][nt]: rows kt*32 + (l>>4)*8 .. +8 of column nt*H + wave*16 + (l&15)  (gate nt of the lane's unit)
  {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // BRANCH-FREE, all 128 loads in flight before the first use: `ok ? W[...] : 0` compiles to a load inside a divergent
    // branch that is waited for at the end of the branch - 64 serial L2 round trips in front of the step loop
    uint16_t raw[4][4][8];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int n = nt * H + min(wave * 16 + (lane & 15), H - 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) raw[kt][nt][i] = Wh[(size_t)min(kt * 32 + (lane >> 4) * 8 + i, H - 1) * G + n];
      }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int un = wave * 16 + (lane & 15);
        i32x4_t t = (i32x4_t){0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = kt * 32 + (lane >> 4) * 8 + i;
          const uint32_t v = (k < H && un < H) ? (uint32_t)raw[kt][nt][i] : 0u;
          t[i >> 1] |= (int)(v << ((i & 1) * 16));
        }
        asm volatile("" : "+a"(t));
        w[kt][nt] = t;
      }
    for (int i = tid; i < 2 * 4 * HSS; i += MNT) hs[0][i] = 0;
  }
  float c = 0.f, h = 0.f;
  // Input-gate rows of the next LPD steps are in flight while a step computes (register ring, the loop is unrolled by LPD):
  // xg was just written by a GEMM and comes from the MALL / HBM (1 - 2 us) - longer than a whole step - and with the loads at
  // the top of the step the compiler also parked an s_waitcnt vmcnt(0) at the loop header, i.e. every step waited for the write
  // acknowledgements of the previous step's eight stores.  The ring loads are branch-free (clamped step / unit); steps beyond
  // len in the last group run with their stores and state updates masked off.
#ifdef SATT_LSTM_PROF
  const bool prof_on = blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64;
  unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = __builtin_amdgcn_s_memtime();
#endif
  float px[LPD][4];
  const int lenc = max(len, 1);
  auto issue = [&](int s, float (&dst)[4]) {
    const int sc = min(s, lenc - 1), t = rev ? (lenc - 1 - sc) : sc;
    const unsigned ju = (unsigned)min((int)(threadIdx.x >> 6) * 16 + (int)(threadIdx.x & 15), H - 1);   // the lane's unit
    const float* xr = xg + (size_t)t * G;              // (scalar row base + 32-bit offsets: common.h ld_su / st_su)
    dst[0] = ld_su(xr, ju); dst[1] = ld_su(xr, H + ju); dst[2] = ld_su(xr, 2 * H + ju); dst[3] = ld_su(xr, 3 * H + ju);
  };
#pragma unroll
  for (int u = 0; u < LPD; ++u) issue(u, px[u]);
  float pv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // results of the previous step (stored one step late, see the loop)
  bool pv_on = false; int pv_t = 0;
  __syncthreads();
  for (int s0 = 0; s0 < len; s0 += LPD) {
#pragma unroll
    for (int u = 0; u < LPD; ++u) {
      const int s = s0 + u;
      const bool live = s < len;
      int oz = 0;
      asm volatile("" : "+v"(oz));                   // keeps index arithmetic inside the step (see attn_cluster.hip)
      const int tid = (int)threadIdx.x + oz, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
      const int j = wave * 16 + (lane & 15);          // the unit whose gates end up in this lane's accumulators (lanes 0..15)
      const int t = rev ? (len - 1 - s) : s;
      const float xi = px[u][0], xj = px[u][1], xf = px[u][2], xo = px[u][3];
      LPROF(0);
      f32x4_t q0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, q1 = q0, q2 = q0, q3 = q0;
      const uint16_t* hrow = hs[u & 1] + min(lane & 15, 3) * HSS + (lane >> 4) * 8;
      // r5: EVERY A operand is requested before the first MFMA block (an LDS read cannot move across an asm block: with the reads
      // inside the K loop each block waited for its own LDS round trip - four exposed latencies per step), and the step's zoneout
      // decisions - functions of (seed, step, unit) only, branch-free - are formed while the reads fly.
      bf16x8_t av[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) av[kt] = *reinterpret_cast<const bf16x8_t*>(hrow + kt * 32);
      const bool mine = lane < 16 && j < H && live;
      const uint32_t idx = ((uint32_t)b * (uint32_t)T + (uint32_t)t) * (uint32_t)H + (uint32_t)j;
      const bool keep_c = (satt_hash(seed, a.sc[d], idx) >= a.zct) | (a.zct == 0);
      const bool keep_h = (satt_hash(seed, a.sh[d], idx) >= a.zht) | (a.zht == 0);
      // The eight result stores of the PREVIOUS step (and their address arithmetic: a third of the cell phase's instruction
      // stream, which is issue bound) are issued between the MFMA blocks of this one, where the wave otherwise waits for the
      // matrix pipe: without the stores a step measured 0.80 us against 0.97 (profiles/r04_encoder_lstm.txt).
      mfma14_a<false>(q0, q1, q2, q3, av[0], w[0][0], w[0][1], w[0][2], w[0][3]);
      if (pv_on) {
        float* gr = gates + (size_t)pv_t * G;
        st_su(gr, (unsigned)j, pv[0]); st_su(gr, (unsigned)(H + j), pv[1]); st_su(gr, (unsigned)(2 * H + j), pv[2]); st_su(gr, (unsigned)(3 * H + j), pv[3]);
      }
      mfma14_a<false, false>(q0, q1, q2, q3, av[1], w[1][0], w[1][1], w[1][2], w[1][3]);
      if (pv_on) {
        st_su(cnew + (size_t)pv_t * H, (unsigned)j, pv[4]);
        st_su(hout + (size_t)pv_t * a.ld, (unsigned)j, pv[5]);
      }
      mfma14_a<false, false>(q0, q1, q2, q3, av[2], w[2][0], w[2][1], w[2][2], w[2][3]);
      if (pv_on) {
        st_su(cstate + (size_t)pv_t * H, (unsigned)j, pv[6]);
        st_su(hstate + (size_t)pv_t * H, (unsigned)j, pv[7]);
      }
      mfma14_a<true, false>(q0, q1, q2, q3, av[3], w[3][0], w[3][1], w[3][2], w[3][3]);
      LPROF(1);
      pv_on = mine; pv_t = t;
      if (mine) {
        const float gi = sigmoidf_(xi + (q0[0] + q0[1] + q0[2]));
        const float gj = tanhf_(xj + (q1[0] + q1[1] + q1[2]));
        const float gf = sigmoidf_(xf + (q2[0] + q2[1] + q2[2]) + 1.0f);
        const float go = sigmoidf_(xo + (q3[0] + q3[1] + q3[2]));
        const float cn = gf * c + gi * gj;
        const float hn = go * tanhf_(cn);
        if (a.training) {
          c = keep_c ? cn : c;
          h = keep_h ? hn : h;
        } else {
          c = (1.f - a.zc) * cn + a.zc * c;
          h = (1.f - a.zh) * hn + a.zh * h;
        }
        xs_put(hs[(u + 1) & 1], HSS, j, h);
        pv[0] = gi; pv[1] = gj; pv[2] = gf; pv[3] = go; pv[4] = cn; pv[5] = hn; pv[6] = c; pv[7] = h;
      }
      // the ring slot is refilled BEHIND its last use: issued at the top of the step the new rows needed registers of their own
      // (the old ones were still live in the cell), and the copies back into the loop-carried registers at the loop's back
      // edge came with an s_waitcnt vmcnt(0) - the whole ring was drained every LPD steps
      LPROF(2);
      issue(s + LPD, px[u]);
      LPROF(3);
      lds_barrier();
      LPROF(4);
    }
  }
#ifdef SATT_LSTM_PROF
  if (prof_on && threadIdx.x == 0) for (int i = 0; i < 8; ++i) satt_lstm_prof[i] = pacc[i];
#endif
  if (pv_on) {          // the last step's results
    const int jj = (int)(threadIdx.x >> 6) * 16 + (int)(threadIdx.x & 15);
    float* gr = gates + (size_t)pv_t * G;
    gr[jj] = pv[0]; gr[H + jj] = pv[1]; gr[2 * H + jj] = pv[2]; gr[3 * H + jj] = pv[3];
    cnew[(size_t)pv_t * H + jj] = pv[4]; hout[(size_t)pv_t * a.ld + jj] = pv[5];
    cstate[(size_t)pv_t * H + jj] = pv[6]; hstate[(size_t)pv_t * H + jj] = pv[7];
  }
  const int j = threadIdx.x;
  for (int t = len; t < T; ++t) {
    if (j < H) {
      hout[(size_t)t * a.ld + j] = 0.f;
      cnew[(size_t)t * H + j] = 0.f; cstate[(size_t)t * H + j] = 0.f; hstate[(size_t)t * H + j] = 0.f;
    }
    for (int n = j; n < G; n += MNT) gates[(size_t)t * G + n] = 0.f;
  }
}

// backward: wave w owns the 16 output units [16w, 16w+16) (one N tile) x all 16 K tiles of dz[4H] = 16 B operands; the cell
// backward of those units runs in lanes 0..15 of the same wave (where the MFMA chain leaves d h_prev of exactly these units),
// so the step has ONE barrier - behind the publication of the split dz rows (double-buffered, as in the forward kernel)
__global__ __launch_bounds__(MNT) void lstm_bwd_mfma_k(const LstmBwdArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t dzs[2][4 * (4 * MH + APAD)];   // bf16 [parity][4][4*MH + APAD]: split dz, row 3 = 0
  const int H = a.H, G = 4 * a.H, T = a.T;
  constexpr int DZS = 4 * MH + APAD;
  const int b = blockIdx.x, d = blockIdx.y;
  const int len = a.lengths ? (int)a.lengths[b] : T;
  const bool rev = (d == 1);
  const size_t dirBT = ((size_t)d * a.B + b) * T;
  const uint16_t* WhT = a.WhT + (size_t)d * G * H;
  const float* gates = a.gates + dirBT * G;
  const float* cnew = a.cnew + dirBT * H;
  const float* cstate = a.cstate + dirBT * H;
  const float* dhout = a.dhout + (size_t)b * T * a.ld + (size_t)d * H;
  float* dxg = a.dxg + dirBT * G;
  const uint32_t seed = a.seed ? *a.seed : 0u;
  // dz is staged gate-major with a fixed stride MH per gate (k = g*MH + j), so the B rows follow that order
  i32x4_t w[16];
  {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = wave * 16 + (lane & 15);
    // branch-free, all 128 loads in flight before the first use (see lstm_fwd_mfma_k)
    uint16_t raw[16][8];
#pragma unroll
    for (int kt = 0; kt < 16; ++kt)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = kt * 32 + (lane >> 4) * 8 + i, g = k / MH, jj = k - g * MH;
        raw[kt][i] = WhT[(size_t)(g * H + min(jj, H - 1)) * H + min(n, H - 1)];
      }
#pragma unroll
    for (int kt = 0; kt < 16; ++kt) {
      i32x4_t t = (i32x4_t){0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = kt * 32 + (lane >> 4) * 8 + i, g = k / MH, jj = k - g * MH;
        const uint32_t v = (jj < H && n < H) ? (uint32_t)raw[kt][i] : 0u;
        t[i >> 1] |= (int)(v << ((i & 1) * 16));
      }
      asm volatile("" : "+a"(t));
      w[kt] = t;
    }
    for (int i = tid; i < 2 * 4 * DZS; i += MNT) dzs[0][i] = 0;
  }
  float dc_state = 0.f, dh_state = 0.f;
  // saved tensors of the next LPD steps in flight (register ring, loop unrolled by LPD; see lstm_fwd_mfma_k): seven values per
  // step, branch-free loads at clamped step / unit
  float pq[LPD][7];
  const int lenc = max(len, 1);
  auto issue = [&](int s, float (&dst)[7]) {
    const int sc = min(max(s, 0), lenc - 1), t = rev ? (lenc - 1 - sc) : sc;
    const int jc = min((int)(threadIdx.x >> 6) * 16 + (int)(threadIdx.x & 15), H - 1);        // the lane's unit
    const float* gr = gates + (size_t)t * G;           // (scalar row bases + 32-bit offsets: common.h ld_su / st_su)
    const unsigned ju = (unsigned)jc;
    dst[0] = ld_su(gr, ju); dst[1] = ld_su(gr, H + ju); dst[2] = ld_su(gr, 2 * H + ju); dst[3] = ld_su(gr, 3 * H + ju);
    dst[4] = ld_su(cnew + (size_t)t * H, ju);
    const int tp = min(max(rev ? t + 1 : t - 1, 0), T - 1);
    dst[5] = ld_su(cstate + (size_t)tp * H, ju);
    dst[6] = ld_su(dhout + (size_t)t * a.ld, ju);
  };
#pragma unroll
  for (int u = 0; u < LPD; ++u) issue(len - 1 - u, pq[u]);
  __syncthreads();
  for (int s0 = len - 1; s0 >= 0; s0 -= LPD) {
#pragma unroll
    for (int u = 0; u < LPD; ++u) {
      const int s = s0 - u;
      const bool live = s >= 0;
      int oz = 0;
      asm volatile("" : "+v"(oz));
      const int tid = (int)threadIdx.x + oz, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
      const int j = wave * 16 + (lane & 15);          // the unit this lane differentiates (lanes 0..15)
      const bool mine = lane < 16 && j < H && live;
      const int t = rev ? (len - 1 - s) : s;
      const float gi = pq[u][0], gj = pq[u][1], gf = pq[u][2], go = pq[u][3], cn = pq[u][4], cp = s > 0 ? pq[u][5] : 0.f, dho = pq[u][6];
      float dh_direct = 0.f;
      float dzv[4] = {0.f, 0.f, 0.f, 0.f};            // this step's d z: stored between the MFMA blocks below (r5)
      {
        // (zoneout decisions: branch-free, ahead of the divergent cell branch)
        const uint32_t idx = ((uint32_t)b * (uint32_t)T + (uint32_t)t) * (uint32_t)H + (uint32_t)j;
        const bool keep_c = (satt_hash(seed, a.sc[d], idx) >= a.zct) | (a.zct == 0);
        const bool keep_h = (satt_hash(seed, a.sh[d], idx) >= a.zht) | (a.zht == 0);
        if (mine) {
          float kc, kh, pc, ph;  // d(state)/d(new), d(state)/d(prev)
          if (a.training) {
            kc = keep_c ? 1.f : 0.f; pc = 1.f - kc;
            kh = keep_h ? 1.f : 0.f; ph = 1.f - kh;
          } else {
            kc = 1.f - a.zc; pc = a.zc; kh = 1.f - a.zh; ph = a.zh;
          }
          const float dhn = dho + kh * dh_state;
          dh_direct = ph * dh_state;
          const float tc = tanhf_(cn);
          const float dcn = dhn * go * (1.f - tc * tc) + kc * dc_state;
          const float d_o = dhn * tc;
          dzv[0] = dcn * gj * gi * (1.f - gi);
          dzv[1] = dcn * gi * (1.f - gj * gj);
          dzv[2] = dcn * cp * gf * (1.f - gf);
          dzv[3] = d_o * go * (1.f - go);
          dc_state = dcn * gf + pc * dc_state;
          uint16_t* zb = dzs[u & 1];
          xs_put(zb, DZS, j, dzv[0]); xs_put(zb, DZS, MH + j, dzv[1]);
          xs_put(zb, DZS, 2 * MH + j, dzv[2]); xs_put(zb, DZS, 3 * MH + j, dzv[3]);
        }
      }
      issue(s - LPD, pq[u]);          // behind the slot's last use (see lstm_fwd_mfma_k)
      lds_barrier();
      {
        f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const uint16_t* zrow = dzs[u & 1] + min(lane & 15, 3) * DZS + (lane >> 4) * 8;
        // r5: the A operands run EIGHT tiles ahead of the chain (an LDS read cannot move across an asm block: with the reads next
        // to their block every pair of K tiles waited for its own LDS round trip - eight exposed latencies per step), and the
        // four d z stores of the step are issued in the shadow of the first blocks instead of inside the cell phase.
        bf16x8_t za[16];
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) za[kt] = *reinterpret_cast<const bf16x8_t*>(zrow + kt * 32);
        float* dr = dxg + (size_t)t * G;
#pragma unroll
        for (int kt = 0; kt < 16; kt += 2) {
          // one chain: operand cover in front of the first block, result cover after the last (mfma_rec.h)
          if (kt == 0) mfma21_a<false>(acc, za[kt], za[kt + 1], w[kt], w[kt + 1]);
          else if (kt + 2 < 16) mfma21_a<false, false>(acc, za[kt], za[kt + 1], w[kt], w[kt + 1]);
          else mfma21_a<true, false>(acc, za[kt], za[kt + 1], w[kt], w[kt + 1]);
          if (kt + 8 < 16) {
            za[kt + 8] = *reinterpret_cast<const bf16x8_t*>(zrow + (kt + 8) * 32);
            za[kt + 9] = *reinterpret_cast<const bf16x8_t*>(zrow + (kt + 9) * 32);
          }
          if (kt == 0 && mine) { st_su(dr, (unsigned)j, dzv[0]); st_su(dr, (unsigned)(H + j), dzv[1]); }
          if (kt == 2 && mine) { st_su(dr, (unsigned)(2 * H + j), dzv[2]); st_su(dr, (unsigned)(3 * H + j), dzv[3]); }
        }
        if (mine) dh_state = acc[0] + acc[1] + acc[2] + dh_direct;
      }
    }
  }
  const int j = threadIdx.x;
  for (int t = len; t < T; ++t)
    for (int n = j; n < G; n += MNT) dxg[(size_t)t * G + n] = 0.f;
}

}  // namespace

#ifdef SATT_LSTM_PROF
extern "C" int satt_lstm_prof_read(unsigned long long* host8) {
  return hipMemcpyFromSymbol(host8, HIP_SYMBOL(satt_lstm_prof), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -3;
}
#endif

extern "C" int satt_lstm_fwd(const float* xg, const uint16_t* Wh, const int64_t* lengths, int ndir, int B, int T,
                             int H, int training, float zc, float zh, uint32_t zc_thresh, uint32_t zh_thresh,
                             const uint32_t* seed, const uint32_t* stream_c, const uint32_t* stream_h, float* hout,
                             int64_t ld_hout, float* gates, float* cnew, float* cstate, float* hstate,
                             void* stream) {
  if (ndir < 1 || ndir > 2 || B <= 0 || T <= 0 || H <= 0) return SATT_E_BADARG;
  if ((4 * H) % 8 != 0 || 4 * H > 8 * LNT || H > LNT) return SATT_E_UNSUPPORTED;
  LstmArgs a;
  a.xg = xg; a.Wh = Wh; a.lengths = lengths; a.B = B; a.T = T; a.H = H; a.training = training;
  a.zc = zc; a.zh = zh; a.zct = zc_thresh; a.zht = zh_thresh; a.seed = seed;
  for (int d = 0; d < 2; ++d) { a.sc[d] = stream_c ? stream_c[d < ndir ? d : 0] : 0; a.sh[d] = stream_h ? stream_h[d < ndir ? d : 0] : 0; }
  a.hout = hout; a.ld = ld_hout; a.gates = gates; a.cnew = cnew; a.cstate = cstate; a.hstate = hstate;
  if (H <= MH) {
    hipLaunchKernelGGL(lstm_fwd_mfma_k, dim3(B, ndir), dim3(MNT), 0, (hipStream_t)stream, a);
    SATT_LAUNCH_CHECK();
    return SATT_OK;
  }
  const size_t smem = sizeof(float) * ((size_t)H + 4 * H + (size_t)LNT * 8);
  hipLaunchKernelGGL(lstm_fwd_k, dim3(B, ndir), dim3(LNT), smem, (hipStream_t)stream, a);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

extern "C" int satt_lstm_bwd(const float* dhout, int64_t ld_dhout, const uint16_t* WhT, const int64_t* lengths,
                             int ndir, int B, int T, int H, int training, float zc, float zh, uint32_t zc_thresh,
                             uint32_t zh_thresh, const uint32_t* seed, const uint32_t* stream_c,
                             const uint32_t* stream_h, const float* gates, const float* cnew, const float* cstate,
                             float* dxg, void* stream) {
  if (ndir < 1 || ndir > 2 || B <= 0 || T <= 0 || H <= 0) return SATT_E_BADARG;
  if (H % 8 != 0 || 4 * H > 8 * LNT || H > LNT) return SATT_E_UNSUPPORTED;
  LstmBwdArgs a;
  a.dhout = dhout; a.ld = ld_dhout; a.WhT = WhT; a.lengths = lengths; a.B = B; a.T = T; a.H = H;
  a.training = training; a.zc = zc; a.zh = zh; a.zct = zc_thresh; a.zht = zh_thresh; a.seed = seed;
  for (int d = 0; d < 2; ++d) { a.sc[d] = stream_c ? stream_c[d < ndir ? d : 0] : 0; a.sh[d] = stream_h ? stream_h[d < ndir ? d : 0] : 0; }
  a.gates = gates; a.cnew = cnew; a.cstate = cstate; a.dxg = dxg;
  if (H <= MH) {
    hipLaunchKernelGGL(lstm_bwd_mfma_k, dim3(B, ndir), dim3(MNT), 0, (hipStream_t)stream, a);
    SATT_LAUNCH_CHECK();
    return SATT_OK;
  }
  const size_t smem = sizeof(float) * ((size_t)4 * H + H + (size_t)LNT * 8);
  hipLaunchKernelGGL(lstm_bwd_k, dim3(B, ndir), dim3(LNT), smem, (hipStream_t)stream, a);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}
