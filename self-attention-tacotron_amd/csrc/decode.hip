// Autoregressive decode step (BASELINE config 5; reference modules/module.py:762-778, modules/rnn_wrappers.py:47-124,
// modules/helpers.py:58-166 mirrors): small-batch kernels whose time index lives in DEVICE memory, so that one
// captured hipGraph of a decoder step (or of several) is replayed for every step without touching a kernel argument.
//   dec_linear_k    y = act([x0 | x1 | x2] W + b) (+ residual): batch rows times a weight matrix read once per
//                   launch (fp32 master or its bf16 plain-cast shadow), every operand addressed as
//                   base + b * batch_stride + step * step_stride - rows of per-step histories are read and written in place
//                   LSTM form: the product is the gate pre-activation and the ZoneoutLSTMCell (inference mode,
//                   interpolating zoneout) runs in the epilogue on double-buffered states
//   dec_attn_energy_k / dec_attn_context_k  the attention step in two launches of many small workgroups: query layer +
//                   location features + energies per slice of memory rows, then masked softmax + forward recursion (or
//                   location-sensitive / forced alignments) + a slice of the context columns
//   dec_self_attn_k the new query row of the causal self-attention over the K|V cache (== the reference's re-run over
//                   the whole history: the mask is causal and there is no padding mask)
//   step counter + stop rule of StopTokenBasedInferenceHelper: evaluated on the device by workgroup (0,0) of launches
//                   that exist anyway (dec_linear_k epilogue)
// The training kernels are built for B * C >= 128 workgroups and 400 steps per launch; relaunching them per step costs
// three prologues (register-resident weight slices) and ~25 host launches per step - the decode is host bound there.
// Here a step is 9 dependent launches; measured on MI355X each dependent launch costs >= 4.7 us start to start however
// little it does (profiles/r02_decode_timeline.txt), so the step time is set by the launch count, not by the arithmetic.
#include "common.h"

namespace {

// Workgroup barriers in this file are lds_barrier() (s_waitcnt lgkmcnt(0); s_barrier): every hand-off between the threads of
// a workgroup goes through LDS, and __syncthreads() would also drain the outstanding global loads - e.g. make the weight rows
// wait before the input rows are even requested (one more L2 round trip per launch).
constexpr int DL_NT = 256;      // threads of dec_linear_k: 8 column groups (4 columns each) x 32 k lanes
constexpr int DL_COLS = 32;     // output columns per workgroup
constexpr int DL_KMAX = 1024;   // staged input features per row

constexpr int DL_KI = DL_KMAX / 32;   // weight rows per thread

__device__ __forceinline__ float dec_act(float s, int a) {
  if (a == SATT_ACT_RELU) return fmaxf(s, 0.f);
  if (a == SATT_ACT_TANH) return tanhf_(s);
  if (a == SATT_ACT_SOFTSIGN) return s / (1.f + fabsf(s));
  return s;
}

// Dropout that stays ON while synthesising (apply_dropout_on_inference; modules/module.py:564-577 passes the flag to the plain
// PreNet layers): the same stateless keep mask as the training kernels (common.h), element index
// ((b * drop_T + step) * N + column) - the C-order index of a [B, drop_T, N] activation, so a teacher-fed decode of drop_T
// steps draws the masks of the batched evaluation pass.  Applied after the activation, before the residual.
__device__ __forceinline__ float dec_drop(float s, const satt_dec_linear_params& p, int b, int64_t step, int col) {
  if (p.drop_thresh == 0u) return s;
  const uint32_t idx = ((uint32_t)b * (uint32_t)p.drop_T + (uint32_t)step) * (uint32_t)p.N + (uint32_t)col;
  return satt_keep(*p.drop_seed, p.drop_stream, idx, p.drop_thresh) ? s * p.drop_scale : 0.f;
}

// step bookkeeping of one parameter block (see dec_linear_k); called by the first wave of workgroup (0,0)
__device__ __forceinline__ void dec_bookkeeping(const satt_dec_linear_params& p, int t, int lane) {
  if (p.stop && t >= 1) {
    bool ok = true;
    for (int b = lane; b < p.B; b += 64) {
      const float sgm = 1.f / (1.f + __expf(-p.stop[(int64_t)b * p.stop_bs + (int64_t)(t - 1) * p.stop_ss]));
      ok = ok && (sgm > p.stop_threshold);
    }
    const bool all = __ballot(!ok) == 0ull;
    if (lane == 0 && all && (t - 1) > p.min_steps && *p.flag == 0) *p.flag = t;
  }
  if (p.step_out && lane == 0) *p.step_out = t + p.step_add;
}

// Every phase of these kernels starts with global loads whose latency (~1 us: L2 / MALL after the previous kernel's
// write-back) is the cost that matters, so the loads that do not depend on earlier results are issued first and all at
// once: here the thread's whole weight column slice goes to registers before the step index is even read.
template <int NB, bool BF16W, bool VEC>
__global__ __launch_bounds__(DL_NT) void dec_linear_k(const satt_dec_linear_params p) {
  __shared__ float xs[NB * DL_KMAX];
  __shared__ float red[32 * NB * DL_COLS];
  const int tid = threadIdx.x, cg = tid & 7, kl = tid >> 3;
  const int n0 = blockIdx.x * DL_COLS, b0 = blockIdx.y * NB;
  // LSTM form (lstm_H > 0, N = 4 H): the workgroup owns 8 units and all four of their gate columns, so that the cell
  // runs in the epilogue.  The caller stores W with its columns regrouped as (unit block, gate, unit in block): the 32
  // columns of a workgroup are then contiguous (gathering them from the i | j | f | o layout touched four cache lines
  // per weight row for 16 bytes each and made the kernel line-traffic bound).
  const int H = p.lstm_H;
  const int n = n0 + 4 * cg;
  int K = p.k[0];
  if (p.nseg > 1) K += p.k[1];
  if (p.nseg > 2) K += p.k[2];
  const int64_t step = p.step ? (int64_t)*p.step : 0;
  const int64_t par = step & 1;        // recurrent states are double-buffered by step parity (read par, write par ^ 1)
  float w[DL_KI][4];
  // BRANCH-FREE: every weight load is issued unconditionally at a clamped (always valid) address and masked afterwards; a
  // load inside `if (k < K)` is waited for at the end of its branch - one L2 latency per weight row instead of one in all
  const int nc = VEC ? min(n, (int)p.ldw - 4) : n;
#pragma unroll
  for (int i = 0; i < DL_KI; ++i) {
    const int k = kl + 32 * i, kc = min(k, K - 1);
    if (VEC) {
      float v0, v1, v2, v3;
      if (BF16W) {
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
        const float v = BF16W ? bf2f(p.Wb[(int64_t)kc * p.ldw + nj]) : p.W[(int64_t)kc * p.ldw + nj];
        w[i][j] = (k < K && n + j < p.N) ? v : 0.f;
      }
    }
  }
  // Input rows and epilogue operands: requested in ONE go, right behind the weights, into registers - a staging loop per
  // segment with its LDS store inside waits for its own loads before the next segment's are even issued (one L2 / MALL round
  // trip per segment, ~1 us each), and so does a bias load in the epilogue.  Thread tid owns the input features
  // k = tid + 256 j (j < 4) of every sample: segment, base and offset are resolved once per j.
  float xin[NB][4];
  bool xok[4];
  {
    const int k0 = p.k[0], k01 = k0 + (p.nseg > 1 ? p.k[1] : 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = tid + DL_NT * j, kc = min(k, K - 1);
      xok[j] = k < K;
      const int sg = kc < k0 ? 0 : (kc < k01 ? 1 : 2);
      const float* base = sg == 0 ? p.x[0] : (sg == 1 ? p.x[1] : p.x[2]);
      const int64_t bs = sg == 0 ? p.x_bs[0] : (sg == 1 ? p.x_bs[1] : p.x_bs[2]);
      const int64_t off = step * (sg == 0 ? p.x_ss[0] : (sg == 1 ? p.x_ss[1] : p.x_ss[2])) +
                          par * (sg == 0 ? p.x_ps[0] : (sg == 1 ? p.x_ps[1] : p.x_ps[2])) + (kc - (sg == 0 ? 0 : (sg == 1 ? k0 : k01)));
#pragma unroll
      for (int b = 0; b < NB; ++b) xin[b][j] = base[(int64_t)min(b0 + b, p.B - 1) * bs + off];
    }
  }
  // LSTM form: previous cell / output state and the four gate biases of this thread's (sample, unit)
  float c_old = 0.f, h_old = 0.f, eb4[4] = {0.f, 0.f, 0.f, 0.f};
  const int eb = tid >> 3, eu = 8 * (int)blockIdx.x + (tid & 7);
  const bool cell = H && tid < NB * 8 && b0 + eb < p.B;
  if (cell) {
    c_old = p.c_state[par * p.B * H + (int64_t)(b0 + eb) * H + eu];
    h_old = p.h_state[par * p.B * H + (int64_t)(b0 + eb) * H + eu];
    if (p.bias) {
#pragma unroll
      for (int g = 0; g < 4; ++g) eb4[g] = p.bias[g * H + eu];
    }
  }
  // plain form: bias and residual of this thread's output (sample tid / 32, column tid % 32)
  float pbias = 0.f, pres = 0.f;
  const int ob = tid / DL_COLS, oc = tid - ob * DL_COLS;
  const bool outp = !H && tid < NB * DL_COLS && b0 + ob < p.B && n0 + oc < p.N;
  if (outp) {
    if (p.bias) pbias = p.bias[n0 + oc];
    if (p.res) pres = p.res[(int64_t)(b0 + ob) * p.res_bs + step * p.res_ss + n0 + oc];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int b = 0; b < NB; ++b) xs[b * DL_KMAX + tid + DL_NT * j] = (xok[j] && b0 + b < p.B) ? xin[b][j] : 0.f;
  lds_barrier();
  float acc[NB][4];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[b][j] = 0.f;
#pragma unroll
  for (int i0 = 0; i0 < DL_KI; i0 += 8) {
    if (32 * i0 < K) {          // uniform: whole groups of 8 weight rows beyond K are skipped (no loads inside the branch)
#pragma unroll
      for (int i = i0; i < i0 + 8; ++i) {
        const int k = kl + 32 * i, kc = min(k, DL_KMAX - 1);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float xv = k < K ? xs[b * DL_KMAX + kc] : 0.f;       // (w is zero there as well; xs may hold anything)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[b][j] += xv * w[i][j];
        }
      }
    }
  }
#pragma unroll
  for (int b = 0; b < NB; ++b)
    *reinterpret_cast<float4*>(red + (kl * NB + b) * DL_COLS + 4 * cg) = make_float4(acc[b][0], acc[b][1], acc[b][2], acc[b][3]);
  lds_barrier();
  if (H) {      // ZoneoutLSTMCell, inference mode: gates i | j | f | o, forget bias 1, interpolating zoneout
    if (cell) {
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float s = 0.f;
#pragma unroll 8
        for (int q = 0; q < 32; ++q) s += red[(q * NB + eb) * DL_COLS + g * 8 + (tid & 7)];
        z[g] = s + eb4[g];
      }
      const float cn = sigmoidf_(z[2] + 1.f) * c_old + sigmoidf_(z[0]) * tanhf_(z[1]);
      const float hn = sigmoidf_(z[3]) * tanhf_(cn);
      const int64_t o = (par ^ 1) * p.B * H + (int64_t)(b0 + eb) * H + eu;
      p.c_state[o] = (1.f - p.zc) * cn + p.zc * c_old;
      p.h_state[o] = (1.f - p.zh) * hn + p.zh * h_old;
      p.y[(int64_t)(b0 + eb) * p.y_bs + step * p.y_ss + eu] = hn;       // the cell output BEFORE zoneout
    }
    return;
  }
  if (outp) {
    float s = 0.f;
#pragma unroll 8
    for (int q = 0; q < 32; ++q) s += red[(q * NB + ob) * DL_COLS + oc];
    s = dec_drop(dec_act(s + pbias, p.act), p, b0 + ob, step, n0 + oc) + pres;
    p.y[(int64_t)(b0 + ob) * p.y_bs + step * p.y_ss + n0 + oc] = s;
  }
  // Step bookkeeping rides on launches that exist anyway (a separate 1-thread launch costs as much as any other: ~4.5 us
  // of launch-to-launch latency).  Workgroup (0,0) may publish a counter derived from the one it read - into a word that
  // no workgroup of THIS launch reads - and evaluate the stop rule of the PREVIOUS step (StopTokenBasedInferenceHelper,
  // modules/helpers.py:103-107 mirrors): finished once sigmoid(stop) > threshold for every sample and time > min_steps.
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid < 64) {
    const int t = (int)step;
    if (p.stop && t >= 1) {
      bool ok = true;
      for (int b = tid; b < p.B; b += 64) {
        const float sgm = 1.f / (1.f + __expf(-p.stop[(int64_t)b * p.stop_bs + (int64_t)(t - 1) * p.stop_ss]));
        ok = ok && (sgm > p.stop_threshold);
      }
      const bool all = __ballot(!ok) == 0ull;
      if (tid == 0 && all && (t - 1) > p.min_steps && *p.flag == 0) *p.flag = t;       // t steps were taken
    }
    if (p.step_out && tid == 0) *p.step_out = t + p.step_add;
  }
}

// ---- two Dense layers in ONE launch: y = act2(act1(x W1 + b1) [+ res1]) W2 + b2) [+ res2], one workgroup per sample.
// A dependent launch costs ~5 us whatever it does, so the two short layers at either end of the step (pre-net 0 -> pre-net 1,
// output transform -> mel | stop projection) share one: every weight row of BOTH layers is requested up front (branch-free,
// clamped), the intermediate vector lives in LDS only.  N1, N2 <= 256, K1 <= 512 (one segment), ldw % 4 == 0.
constexpr int D2_NT = 512, D2_KG = D2_NT / 64;       // 64 column groups (4 columns each) x 8 row groups
constexpr int D2_I = 256 / D2_KG;                    // weight rows per thread and layer (K1, N1 <= 256)

// bf16 weights only (the benchmark precision): the 2 x 32 rows of a thread stay packed in 128 registers
__global__ __launch_bounds__(D2_NT) void dec_linear2_k(const satt_dec_linear_params p1, const satt_dec_linear_params p2) {
  __shared__ float xs[256];
  __shared__ float hs[256];
  __shared__ __attribute__((aligned(16))) float red[D2_KG * 256];
  const int tid = threadIdx.x, cgp = tid & 63, kq = tid >> 6, b = blockIdx.x, n = 4 * cgp;
  const int K1 = p1.k[0], N1 = p1.N, N2 = p2.N;
  uint2 w1[D2_I], w2[D2_I];
  {   // branch-free: clamped (always valid) rows and columns; rows beyond K meet x = 0, columns beyond N are never stored
    const int n1 = min(n, (int)p1.ldw - 4), n2 = min(n, (int)p2.ldw - 4);
#pragma unroll
    for (int i = 0; i < D2_I; ++i) w1[i] = *reinterpret_cast<const uint2*>(p1.Wb + (int64_t)min(kq + D2_KG * i, K1 - 1) * p1.ldw + n1);
#pragma unroll
    for (int i = 0; i < D2_I; ++i) w2[i] = *reinterpret_cast<const uint2*>(p2.Wb + (int64_t)min(kq + D2_KG * i, N1 - 1) * p2.ldw + n2);
  }
  const int64_t step1 = p1.step ? (int64_t)*p1.step : 0, step2 = p2.step ? (int64_t)*p2.step : 0;
  float pb1 = 0.f, pr1 = 0.f, pb2 = 0.f, pr2 = 0.f;        // epilogue operands of thread n = tid
  if (tid < N1) {
    if (p1.bias) pb1 = p1.bias[tid];
    if (p1.res) pr1 = p1.res[(int64_t)b * p1.res_bs + step1 * p1.res_ss + tid];
  }
  if (tid < N2) {
    if (p2.bias) pb2 = p2.bias[tid];
    if (p2.res) pr2 = p2.res[(int64_t)b * p2.res_bs + step2 * p2.res_ss + tid];
  }
  for (int k = tid; k < K1; k += D2_NT) xs[k] = p1.x[0][(int64_t)b * p1.x_bs[0] + step1 * p1.x_ss[0] + (step1 & 1) * p1.x_ps[0] + k];
  lds_barrier();
  auto layer = [&](const float* xin, int K, const uint2 (&w)[D2_I]) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i0 = 0; i0 < D2_I; i0 += 8) {
      if (D2_KG * i0 < K) {
#pragma unroll
        for (int i = i0; i < i0 + 8; ++i) {
          const int k = kq + D2_KG * i;
          const float xv = k < K ? xin[min(k, 255)] : 0.f;
          acc[0] += xv * __uint_as_float(w[i].x << 16); acc[1] += xv * __uint_as_float(w[i].x & 0xFFFF0000u);
          acc[2] += xv * __uint_as_float(w[i].y << 16); acc[3] += xv * __uint_as_float(w[i].y & 0xFFFF0000u);
        }
      }
    }
    *reinterpret_cast<float4*>(red + kq * 256 + n) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  };
  auto act = [](float s, int a) {
    if (a == SATT_ACT_RELU) return fmaxf(s, 0.f);
    if (a == SATT_ACT_TANH) return tanhf_(s);
    if (a == SATT_ACT_SOFTSIGN) return s / (1.f + fabsf(s));
    return s;
  };
  layer(xs, K1, w1);
  lds_barrier();
  if (tid < N1) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < D2_KG; ++q) s += red[q * 256 + tid];
    hs[tid] = dec_drop(act(s + pb1, p1.act), p1, b, step1, tid) + pr1;
  }
  lds_barrier();
  layer(hs, N1, w2);
  lds_barrier();
  if (tid < N2) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < D2_KG; ++q) s += red[q * 256 + tid];
    p2.y[(int64_t)b * p2.y_bs + step2 * p2.y_ss + tid] = dec_drop(act(s + pb2, p2.act), p2, b, step2, tid) + pr2;
  }
  // step bookkeeping of either layer (see dec_linear_k), by workgroup 0
  if (blockIdx.x == 0 && tid < 64) {
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const satt_dec_linear_params& p = which ? p2 : p1;
      const int t = (int)(which ? step2 : step1);
      if (p.stop && t >= 1) {
        bool ok = true;
        for (int bb = tid; bb < p.B; bb += 64) {
          const float sgm = 1.f / (1.f + __expf(-p.stop[(int64_t)bb * p.stop_bs + (int64_t)(t - 1) * p.stop_ss]));
          ok = ok && (sgm > p.stop_threshold);
        }
        const bool all = __ballot(!ok) == 0ull;
        if (tid == 0 && all && (t - 1) > p.min_steps && *p.flag == 0) *p.flag = t;
      }
      if (p.step_out && tid == 0) *p.step_out = t + p.step_add;
    }
  }
}

// ---- a chain of layers in ONE launch: up to two short Dense layers (pre-net 0 -> pre-net 1; the folded output transform)
// computed REDUNDANTLY by every workgroup in front of its slice of the main layer (the attention LSTM gate product and cell;
// the mel | stop projection).  A dependent launch costs ~5 us start to start; a [256, 256] bf16 layer costs a workgroup of 512
// threads ~1 us (128 KB from L2, 64 weight rows per thread) - the chain trades two or three launches for that.  Every weight
// row of ALL layers is requested up front (branch-free, clamped); the intermediate vectors live in LDS only (workgroup 0 also
// writes them to the layers' y for inspection).  bf16 weights, 8-byte weight loads, pre-layers: one segment, K, N <= 256.
constexpr int DCH_NT = 512;
constexpr int DCH_KL = DCH_NT / 8;          // main layer: 8 column groups (4 columns each) x 64 k lanes
constexpr int DCH_KI = DL_KMAX / DCH_KL;    // main weight rows per thread
constexpr int DCH_KQ = DCH_NT / 64;         // pre-layers: 64 column groups x 8 k groups
constexpr int DCH_PI = 256 / DCH_KQ;        // pre-layer weight rows per thread

template <int NB, int NPRE>
__global__ __launch_bounds__(DCH_NT) void dec_chain_k(const satt_dec_linear_params q0, const satt_dec_linear_params q1,
                                                      const satt_dec_linear_params p) {
  __shared__ float xs[NB * DL_KMAX];
  __shared__ float hs[2][NB * 256];
  __shared__ __attribute__((aligned(16))) float red[DCH_KL * NB * DL_COLS];      // == DCH_KQ * NB * 256 floats
  const int tid = threadIdx.x, cg = tid & 7, kl = tid >> 3, cgp = tid & 63, kq = tid >> 6;
  const int n0 = blockIdx.x * DL_COLS, b0 = blockIdx.y * NB, n = n0 + 4 * cg, H = p.lstm_H;
  int K = p.k[0];
  if (p.nseg > 1) K += p.k[1];
  if (p.nseg > 2) K += p.k[2];
  const int64_t step = p.step ? (int64_t)*p.step : 0, par = step & 1;
  const int64_t step0 = q0.step ? (int64_t)*q0.step : 0, step1 = (NPRE > 1 && q1.step) ? (int64_t)*q1.step : 0;
  // ---- every global operand of every layer is requested here, in the order of use, before the first wait: weights (packed
  // bf16 pairs stay packed: 64 + 64 + 32 registers), input rows, epilogue operands (see dec_linear_k)
  uint2 wa[DCH_PI], wb[NPRE > 1 ? DCH_PI : 1], wm[DCH_KI];
  {
    const int na = min(4 * cgp, (int)q0.ldw - 4);
#pragma unroll
    for (int i = 0; i < DCH_PI; ++i) wa[i] = *reinterpret_cast<const uint2*>(q0.Wb + (int64_t)min(kq + DCH_KQ * i, q0.k[0] - 1) * q0.ldw + na);
  }
  // the first pre-layer's input rows: sample e / 256, feature e % 256 for e = tid + 512 u
  constexpr int HU = (NB * 256 + DCH_NT - 1) / DCH_NT;
  float hin[HU];
  const int Ka = q0.k[0];
#pragma unroll
  for (int u = 0; u < HU; ++u) {
    const int e = tid + DCH_NT * u, b = e >> 8, k = e & 255;
    hin[u] = q0.x[0][(int64_t)min(b0 + b, p.B - 1) * q0.x_bs[0] + step0 * q0.x_ss[0] + (step0 & 1) * q0.x_ps[0] + min(k, Ka - 1)];
  }
  {
    if constexpr (NPRE > 1) {
      const int nq = min(4 * cgp, (int)q1.ldw - 4);
#pragma unroll
      for (int i = 0; i < DCH_PI; ++i) wb[i] = *reinterpret_cast<const uint2*>(q1.Wb + (int64_t)min(kq + DCH_KQ * i, q1.k[0] - 1) * q1.ldw + nq);
    }
    const int nc = min(n, (int)p.ldw - 4);
#pragma unroll
    for (int i = 0; i < DCH_KI; ++i) wm[i] = *reinterpret_cast<const uint2*>(p.Wb + (int64_t)min(kl + DCH_KL * i, K - 1) * p.ldw + nc);
  }
  // epilogue operands of the pre-layers: output e = tid + 512 u -> (sample e / N, column e % N)
  float pqb[NPRE][HU], pqr[NPRE][HU];
#pragma unroll
  for (int j = 0; j < NPRE; ++j) {
    const satt_dec_linear_params& q = j ? q1 : q0;
    const int64_t sq = j ? step1 : step0;
#pragma unroll
    for (int u = 0; u < HU; ++u) {
      const int e = min(tid + DCH_NT * u, NB * q.N - 1), b = e / q.N, c = e - b * q.N;
      pqb[j][u] = q.bias ? q.bias[c] : 0.f;
      pqr[j][u] = q.res ? q.res[(int64_t)min(b0 + b, p.B - 1) * q.res_bs + sq * q.res_ss + c] : 0.f;
    }
  }
  // the main layer's segments behind the chained one: features k = tid + 512 j (j < 2) of every sample
  float xin[NB][2];
  bool xok[2];
  {
    const int k0 = p.k[0], k01 = k0 + (p.nseg > 1 ? p.k[1] : 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = tid + DCH_NT * j, kc = min(max(k, k0), K - 1);
      xok[j] = k >= k0 && k < K;
      const bool s1 = kc < k01;
      const float* base = s1 ? p.x[1] : p.x[2];
      const int64_t bs = s1 ? p.x_bs[1] : p.x_bs[2];
      const int64_t off = step * (s1 ? p.x_ss[1] : p.x_ss[2]) + par * (s1 ? p.x_ps[1] : p.x_ps[2]) + (kc - (s1 ? k0 : k01));
#pragma unroll
      for (int b = 0; b < NB; ++b) xin[b][j] = (p.nseg > 1) ? base[(int64_t)min(b0 + b, p.B - 1) * bs + off] : 0.f;
    }
  }
  float c_old = 0.f, h_old = 0.f, eb4[4] = {0.f, 0.f, 0.f, 0.f};
  const int eb = tid >> 3, eu = 8 * (int)blockIdx.x + (tid & 7);
  const bool cell = H && tid < NB * 8 && b0 + eb < p.B;
  if (cell) {
    c_old = p.c_state[par * p.B * H + (int64_t)(b0 + eb) * H + eu];
    h_old = p.h_state[par * p.B * H + (int64_t)(b0 + eb) * H + eu];
    if (p.bias) {
#pragma unroll
      for (int g = 0; g < 4; ++g) eb4[g] = p.bias[g * H + eu];
    }
  }
  float pbias = 0.f, pres = 0.f;
  const int ob = tid / DL_COLS, oc = tid - ob * DL_COLS;
  const bool outp = !H && tid < NB * DL_COLS && b0 + ob < p.B && n0 + oc < p.N;
  if (outp) {
    if (p.bias) pbias = p.bias[n0 + oc];
    if (p.res) pres = p.res[(int64_t)(b0 + ob) * p.res_bs + step * p.res_ss + n0 + oc];
  }
#pragma unroll
  for (int u = 0; u < HU; ++u) {
    const int e = tid + DCH_NT * u, b = e >> 8, k = e & 255;
    if (e < NB * 256) hs[0][e] = (k < Ka && b0 + b < p.B) ? hin[u] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int b = 0; b < NB; ++b)
      if (xok[j]) xs[b * DL_KMAX + tid + DCH_NT * j] = (b0 + b < p.B) ? xin[b][j] : 0.f;
  lds_barrier();
  // ---- pre-layers: 4 columns x DCH_PI rows per thread; partials through `red`, epilogue into the next layer's input rows
  auto pre = [&](const satt_dec_linear_params& q, int64_t stepq, const uint2 (&w)[DCH_PI], const float (&qb)[HU], const float (&qr)[HU],
                 const float* xrow, float* dst, int dst_ld) {
    const int Kq = q.k[0], Nq = q.N;
    float acc[NB][4];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[b][j] = 0.f;
#pragma unroll
    for (int i0 = 0; i0 < DCH_PI; i0 += 8) {
      if (DCH_KQ * i0 < Kq) {
#pragma unroll
        for (int i = i0; i < i0 + 8; ++i) {
          const int k = kq + DCH_KQ * i;
          const float w0 = __uint_as_float(w[i].x << 16), w1 = __uint_as_float(w[i].x & 0xFFFF0000u);
          const float w2 = __uint_as_float(w[i].y << 16), w3 = __uint_as_float(w[i].y & 0xFFFF0000u);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const float xv = k < Kq ? xrow[b * 256 + min(k, 255)] : 0.f;
            acc[b][0] += xv * w0; acc[b][1] += xv * w1; acc[b][2] += xv * w2; acc[b][3] += xv * w3;
          }
        }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
      *reinterpret_cast<float4*>(red + (kq * NB + b) * 256 + 4 * cgp) = make_float4(acc[b][0], acc[b][1], acc[b][2], acc[b][3]);
    lds_barrier();
#pragma unroll
    for (int u = 0; u < HU; ++u) {
      const int e = tid + DCH_NT * u;
      if (e < NB * Nq) {
        const int b = e / Nq, c = e - b * Nq;
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < DCH_KQ; ++g) s += red[(g * NB + b) * 256 + c];
        s = dec_drop(dec_act(s + qb[u], q.act), q, b0 + b, stepq, c) + qr[u];
        dst[b * dst_ld + c] = s;
        if (blockIdx.x == 0 && b0 + b < p.B && q.y) q.y[(int64_t)(b0 + b) * q.y_bs + stepq * q.y_ss + c] = s;
      }
    }
    lds_barrier();
  };
  if constexpr (NPRE > 1) {
    pre(q0, step0, wa, pqb[0], pqr[0], hs[0], hs[1], 256);
    pre(q1, step1, wb, pqb[1], pqr[1], hs[1], xs, DL_KMAX);
  } else {
    pre(q0, step0, wa, pqb[0], pqr[0], hs[0], xs, DL_KMAX);
  }
  // ---- main layer (dec_linear_k with 64 k lanes)
  float acc[NB][4];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[b][j] = 0.f;
#pragma unroll
  for (int i0 = 0; i0 < DCH_KI; i0 += 4) {
    if (DCH_KL * i0 < K) {
#pragma unroll
      for (int i = i0; i < i0 + 4; ++i) {
        const int k = kl + DCH_KL * i, kc = min(k, DL_KMAX - 1);
        const bool ok = k < K && n < p.N;
        const float w0 = ok ? __uint_as_float(wm[i].x << 16) : 0.f, w1 = ok ? __uint_as_float(wm[i].x & 0xFFFF0000u) : 0.f;
        const float w2 = ok ? __uint_as_float(wm[i].y << 16) : 0.f, w3 = ok ? __uint_as_float(wm[i].y & 0xFFFF0000u) : 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float xv = k < K ? xs[b * DL_KMAX + kc] : 0.f;
          acc[b][0] += xv * w0; acc[b][1] += xv * w1; acc[b][2] += xv * w2; acc[b][3] += xv * w3;
        }
      }
    }
  }
#pragma unroll
  for (int b = 0; b < NB; ++b)
    *reinterpret_cast<float4*>(red + (kl * NB + b) * DL_COLS + 4 * cg) = make_float4(acc[b][0], acc[b][1], acc[b][2], acc[b][3]);
  lds_barrier();
  if (H) {
    if (cell) {
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float s = 0.f;
#pragma unroll 8
        for (int q = 0; q < DCH_KL; ++q) s += red[(q * NB + eb) * DL_COLS + g * 8 + (tid & 7)];
        z[g] = s + eb4[g];
      }
      const float cn = sigmoidf_(z[2] + 1.f) * c_old + sigmoidf_(z[0]) * tanhf_(z[1]);
      const float hn = sigmoidf_(z[3]) * tanhf_(cn);
      const int64_t o = (par ^ 1) * p.B * H + (int64_t)(b0 + eb) * H + eu;
      p.c_state[o] = (1.f - p.zc) * cn + p.zc * c_old;
      p.h_state[o] = (1.f - p.zh) * hn + p.zh * h_old;
      p.y[(int64_t)(b0 + eb) * p.y_bs + step * p.y_ss + eu] = hn;
    }
  } else {
    if (outp) {
      float s = 0.f;
#pragma unroll 8
      for (int q = 0; q < DCH_KL; ++q) s += red[(q * NB + ob) * DL_COLS + oc];
      s = dec_drop(dec_act(s + pbias, p.act), p, b0 + ob, step, n0 + oc) + pres;
      p.y[(int64_t)(b0 + ob) * p.y_bs + step * p.y_ss + n0 + oc] = s;
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid < 64) {
    dec_bookkeeping(q0, (int)step0, tid);
    if (NPRE > 1) dec_bookkeeping(q1, (int)step1, tid);
    dec_bookkeeping(p, (int)step, tid);
  }
}

constexpr int DA_NT = 1024, DA_NW = DA_NT / 64;

__device__ __forceinline__ float block_max(float v, float* sm, int tid) {
  v = wave_max(v);
  lds_barrier();
  if ((tid & 63) == 0) sm[tid >> 6] = v;
  lds_barrier();
  float m = sm[0];
#pragma unroll
  for (int w = 1; w < DA_NW; ++w) m = fmaxf(m, sm[w]);
  return m;
}
__device__ __forceinline__ float block_sum(float v, float* sm, int tid) {
  v = wave_sum(v);
  lds_barrier();
  if ((tid & 63) == 0) sm[tid >> 6] = v;
  lds_barrier();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < DA_NW; ++w) s += sm[w];
  return s;
}

// ---- attention step, spread over the chip.  One workgroup per sample is bound by the VALU / load throughput of ONE
// CU (Ti * U1 tanh evaluations plus 330 KB of keys, values and query-layer weights: 28 us per step at B = 1), so the step
// is two launches of many small workgroups:
//   dec_attn_energy_k  grid (B, NS): a slice of the memory rows - processed query pq = h W_q (recomputed per workgroup:
//                      the query layer is fused, no separate launch), location features, energies of both mechanisms
//   dec_attn_context_k grid (B, NC): masked softmax + forward recursion (recomputed per workgroup from the Ti energies:
//                      cheap) and a slice of the context columns; workgroup 0 writes the state and the alignment rows
// The recurrent attention state (location-conv input, previous forward variable) is double-buffered by step parity:
// every workgroup of a launch reads [t & 1] while workgroup 0 writes [(t & 1) ^ 1].
constexpr int DE_NT = 512, DE_NW = DE_NT / 64;      // energy kernel: 8 waves, one memory row each per pass
constexpr int DC_NT = 256;                          // context kernel: 8 column groups (float4) x 32 row groups
constexpr int DC_CG = 8;

__host__ __device__ inline int de_rows(int Ti, int NS) { return (Ti + NS - 1) / NS; }

// dynamic LDS (floats): hq [A] | part [DE_NW * UQ] | pq [UQ] | ftab [KW * F + F] | aw [R + KW] | fl [R * F]
template <int F, bool BF16W>
__global__ __launch_bounds__(DE_NT) void dec_attn_energy_k(const satt_dec_attention_params p, int NS) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Ti = p.Ti, U1 = p.U1, U2 = p.U2, UQ = U1 + U2, KW = p.kernel, A = p.A, PL = (KW - 1) / 2;
  const int R = de_rows(Ti, NS), r0 = sl * R;
  float* hqs = smem; float* part = hqs + A; float* pq = part + DE_NW * UQ; float* ftab = pq + UQ;
  float* aw = ftab + KW * F + F; float* fl = aw + R + KW;
  // query-layer weights of this thread: 4 columns x A / 8 rows (rows wave, wave + 8, ...), every load issued at once
  constexpr int QR = 32;                       // A <= 256
  // BRANCH-FREE (see dec_linear_k): clamped, always valid addresses, masked afterwards - a load inside a divergent `if` is
  // waited for at the end of its branch, i.e. 32 serial L2 round trips here
  float wq[QR][4];
  {
    const int nq = min(4 * lane, UQ - 4);
    const bool okc = 4 * lane < UQ;
#pragma unroll
    for (int i = 0; i < QR; ++i) {
      const int k = wave + DE_NW * i, kc = min(k, A - 1);
      const bool ok = okc && k < A;
      if (BF16W) {
        const uint2 v = *reinterpret_cast<const uint2*>(p.Wqb + (int64_t)kc * UQ + nq);
        wq[i][0] = ok ? __uint_as_float(v.x << 16) : 0.f; wq[i][1] = ok ? __uint_as_float(v.x & 0xFFFF0000u) : 0.f;
        wq[i][2] = ok ? __uint_as_float(v.y << 16) : 0.f; wq[i][3] = ok ? __uint_as_float(v.y & 0xFFFF0000u) : 0.f;
      } else {
        const float4 v = *reinterpret_cast<const float4*>(p.Wq + (int64_t)kc * UQ + nq);
        wq[i][0] = ok ? v.x : 0.f; wq[i][1] = ok ? v.y : 0.f; wq[i][2] = ok ? v.z : 0.f; wq[i][3] = ok ? v.w : 0.f;
      }
    }
  }
  const int t = p.step ? *p.step : 0;
  const int len = (int)p.lengths[b];
  const float* ga = p.a_state + ((int64_t)(t & 1) * p.B + b) * Ti;
  // staged operands: one element of each per thread (R + KW, A, KW * F + F <= 512), stored to LDS after EVERY other load of
  // the kernel has been issued (the store waits for all earlier loads: they return in order)
  float st_a, st_h, st_f;
  {
    const int tt = r0 + tid - PL;
    const float av = ga[min(max(tt, 0), Ti - 1)];
    const float hv = p.hq[(int64_t)b * A + min(tid, A - 1)];
    const float fv = tid < KW * F ? p.locF[min(tid, KW * F - 1)] : p.locFb[min(max(tid - KW * F, 0), F - 1)];
    st_a = (tt >= 0 && tt < Ti) ? av : 0.f; st_h = hv; st_f = fv;
  }
  // lane constants of the energy phase and the key rows of this wave (independent of pq): requested now
  const int d0 = lane * 4;
  float v1r[4], b1r[4], Ur[F][4];
  {
    const int dc = min(d0, U1 - 4);             // U1 % 4 == 0: a lane's four units are all inside or all outside
    const bool ok = d0 < U1;                    // (scalar loads: the parameter views are only 4-byte aligned)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float vv = p.v1[dc + q], bb = p.b1[dc + q];
      v1r[q] = ok ? vv : 0.f; b1r[q] = ok ? bb : 0.f;
#pragma unroll
      for (int f = 0; f < F; ++f) { const float uu = p.locU[f * U1 + dc + q]; Ur[f][q] = ok ? uu : 0.f; }
    }
  }
  const float v2r = U2 ? (lane < U2 ? p.v2[min(lane, U2 - 1)] : 0.f) : 0.f;
  const float* k1 = p.keys1 + (int64_t)b * Ti * U1;
  const float* k2 = U2 ? p.keys2 + (int64_t)b * Ti * U2 : nullptr;
  constexpr int RP = 2;                        // row passes held in registers (R <= RP * DE_NW rows per slice)
  float4 kk[RP]; float kk2[RP];
#pragma unroll
  for (int u = 0; u < RP; ++u) {     // branch-free: clamped row / unit (rows beyond the slice are never stored, units beyond U carry zero weights)
    const int tt = min(r0 + wave + DE_NW * u, Ti - 1);
    kk[u] = *reinterpret_cast<const float4*>(k1 + (int64_t)tt * U1 + min(d0, U1 - 4));
    kk2[u] = U2 ? k2[(int64_t)tt * U2 + min(lane, U2 - 1)] : 0.f;
  }
  if (tid < R + KW) aw[tid] = st_a;
  if (tid < A) hqs[tid] = st_h;
  if (tid < KW * F + F) ftab[tid] = st_f;
  lds_barrier();
  {   // partial processed query of this wave's rows of W_q
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < QR; ++i) {
      const int k = wave + DE_NW * i;
      const float h = k < A ? hqs[k] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) a4[j] += h * wq[i][j];
    }
    if (4 * lane < UQ) *reinterpret_cast<float4*>(part + wave * UQ + 4 * lane) = make_float4(a4[0], a4[1], a4[2], a4[3]);
  }
  // location features of the slice (conv1d SAME of the previous alignments, 1 -> F channels; forward_attention.py:98-100)
  for (int i = tid; i < R * F; i += DE_NT) {
    const int rr = i / F, f = i - rr * F;
    float s = ftab[KW * F + f];
    for (int j = 0; j < KW; ++j) s += aw[rr + j] * ftab[j * F + f];
    fl[i] = s;
  }
  lds_barrier();
  for (int i = tid; i < UQ; i += DE_NT) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < DE_NW; ++w) s += part[w * UQ + i];
    pq[i] = s;
    if (p.pq_out && sl == 0) p.pq_out[((int64_t)(t & 1) * p.B + b) * UQ + i] = s;      // double-buffered by step parity
  }
  lds_barrier();
  float c1r[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) c1r[q] = d0 + q < U1 ? b1r[q] + pq[d0 + q] : 0.f;
  const float pq2 = lane < U2 ? pq[U1 + lane] : 0.f;
  float acc[RP], acc2[RP];
#pragma unroll
  for (int u = 0; u < RP; ++u) {
    const int i = min(wave + DE_NW * u, R - 1);
    const float kq[4] = {kk[u].x, kk[u].y, kk[u].z, kk[u].w};
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float x = kq[q] + c1r[q];
#pragma unroll
      for (int f = 0; f < F; ++f) x += fl[i * F + f] * Ur[f][q];
      a += v1r[q] * tanhf_(x);              // lanes beyond U1: zero weight, finite argument
    }
    acc[u] = a;
    acc2[u] = v2r * tanhf_(kk2[u] + pq2);
  }
  wave_sum_multi<RP>(acc);
  wave_sum_multi<RP>(acc2);
  if (lane == 0) {
#pragma unroll
    for (int u = 0; u < RP; ++u) {
      const int i = wave + DE_NW * u, tt = r0 + i;
      if (i < R && tt < len) { p.e1[(int64_t)b * Ti + tt] = acc[u]; if (U2) p.e2[(int64_t)b * Ti + tt] = acc2[u]; }
    }
  }
}

// dynamic LDS (floats): a1 [Ti] | a2 [Ti] | alphap [Ti] | aold [Ti] | part [32 * 32] | sm [4]
__global__ __launch_bounds__(DC_NT) void dec_attn_context_k(const satt_dec_attention_params p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, cs = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Ti = p.Ti, V1 = p.V1, V2 = p.V2, CT = V1 + V2;
  float* a1 = smem; float* a2 = a1 + Ti; float* alphap = a2 + Ti; float* aold = alphap + Ti;
  float* part = aold + Ti; float* sm = part + 32 * 4 * DC_CG;
  const int t = p.step ? *p.step : 0;
  const int par = t & 1;
  const int len = (int)p.lengths[b];
  const bool forced = p.teach1 != nullptr, dual = V2 > 0;
  const int64_t row = ((int64_t)b * p.Td + t) * Ti;
  // this thread's slice of the values: float4 column group (cs * 8 + tid % 8) of [values1 | values2], rows tid / 8 + 32 i
  const int cg = cs * DC_CG + (tid & 7), rg = tid >> 3, col = 4 * cg;
  const bool s1c = col < V1, cok = col < CT;
  const float* vv = s1c ? p.values1 + (int64_t)b * Ti * V1 + col : (cok ? p.values2 + (int64_t)b * Ti * V2 + (col - V1) : nullptr);
  const int ld = s1c ? V1 : V2;
  constexpr int NR = 8;                        // rows per thread in flight: 256 rows per pass over the workgroup
  float4 x[NR];
  const float* vs = cok ? vv : p.values1 + (int64_t)b * Ti * V1;       // column groups beyond CT: any valid address, never stored
  const int lds_ = cok ? ld : V1;
#pragma unroll
  for (int u = 0; u < NR; ++u)       // branch-free: rows clamped into the memory, their weight is zero below
    x[u] = *reinterpret_cast<const float4*>(vs + (int64_t)min(rg + 32 * u, Ti - 1) * lds_);
  if (!forced) {
    for (int i = tid; i < Ti; i += DC_NT) {
      a1[i] = i < len ? p.e1[(int64_t)b * Ti + i] : -INFINITY;
      a2[i] = (dual && i < len) ? p.e2[(int64_t)b * Ti + i] : -INFINITY;
      alphap[i] = p.alpha_state[((int64_t)par * p.B + b) * Ti + i];
      aold[i] = p.a_state[((int64_t)par * p.B + b) * Ti + i];
    }
  } else {    // forced alignments (modules/teacher_forcing_attention.py:31-38): the step's rows of the given histories
    for (int i = tid; i < Ti; i += DC_NT) { a1[i] = p.teach1[row + i]; a2[i] = p.teach2 ? p.teach2[row + i] : 0.f; }
  }
  lds_barrier();
  auto bmax = [&](float v) {
    v = wave_max(v);
    lds_barrier();
    if (lane == 0) sm[wave] = v;
    lds_barrier();
    return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  };
  auto bsum = [&](float v) {
    v = wave_sum(v);
    lds_barrier();
    if (lane == 0) sm[wave] = v;
    lds_barrier();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
  };
  float* ga_n = p.a_state + ((int64_t)(par ^ 1) * p.B + b) * Ti;
  float* gal_n = p.alpha_state + ((int64_t)(par ^ 1) * p.B + b) * Ti;
  // transition probability of this step's recursion: 0.5, or - with the transition agent (modules/forward_attention.py:111-116)
  // - predicted from the PREVIOUS step's [context 1 | processed query 1] (both double-buffered by step parity: this launch
  // writes buffer `par` while every workgroup reads buffer `par ^ 1`); recomputed by every workgroup (V1 + U1 products)
  float ut = 0.5f;
  if (!forced && p.agentW && t > 0) {
    const float* cprev = p.ctx + ((int64_t)(par ^ 1) * p.B + b) * CT;
    const float* qprev = p.pq_out + ((int64_t)(par ^ 1) * p.B + b) * (p.U1 + p.U2);
    float z = 0.f;
    for (int i = tid; i < V1 + p.U1; i += DC_NT) z += (i < V1 ? cprev[i] : qprev[i - V1]) * p.agentW[i];
    z = bsum(z);
    ut = 1.f / (1.f + __expf(-(z + p.agentb[0])));
  }
  if (!forced) {
    // masked softmax of both energy rows (TF _maybe_mask_score(-inf) + softmax)
    float m1 = -INFINITY, m2 = -INFINITY;
    for (int i = tid; i < len; i += DC_NT) { m1 = fmaxf(m1, a1[i]); m2 = fmaxf(m2, a2[i]); }
    m1 = bmax(m1); m2 = bmax(m2);
    float s1 = 0.f, s2 = 0.f;
    for (int i = tid; i < Ti; i += DC_NT) {
      const float x1 = i < len ? __expf(a1[i] - m1) : 0.f, x2 = (dual && i < len) ? __expf(a2[i] - m2) : 0.f;
      a1[i] = x1; a2[i] = x2; s1 += x1; s2 += x2;
    }
    s1 = bsum(s1); s2 = bsum(s2);
    const float r1 = 1.f / s1, r2 = dual ? 1.f / s2 : 0.f;
    float sa = 0.f;
    for (int i = tid; i < Ti; i += DC_NT) {
      const float a = a1[i] * r1;
      a2[i] *= r2;
      // next location-conv input: the softmax alignments, or their running sum (forward_attention.py:118-121)
      if (cs == 0) ga_n[i] = p.cumulative ? a + aold[i] : a;
      float al = a;
      if (p.att1_mode == 0) {     // forward recursion (:104-110) with the transition probability ut
        al = ((1.f - ut) * alphap[i] + ut * (i > 0 ? alphap[i - 1] : 0.f) + 1e-7f) * a;
        sa += al;
      }
      a1[i] = al;
    }
    if (p.att1_mode == 0) {
      sa = bsum(sa);
      const float rs = 1.f / sa;
      for (int i = tid; i < Ti; i += DC_NT) a1[i] *= rs;
    }
    lds_barrier();
  }
  if (cs == 0) {
    for (int i = tid; i < Ti; i += DC_NT) {
      gal_n[i] = a1[i];
      if (forced) ga_n[i] = a1[i];
      p.align1[row + i] = a1[i];
      if (p.align2) p.align2[row + i] = a2[i];
    }
  }
  // context slice
  const float* al = s1c ? a1 : a2;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int tt = rg + 32 * u;
    const float w = tt < len ? al[tt] : 0.f;
    acc.x += w * x[u].x; acc.y += w * x[u].y; acc.z += w * x[u].z; acc.w += w * x[u].w;
  }
  for (int t0 = 32 * NR; t0 < len; t0 += 32 * NR) {      // memories longer than 256 rows
#pragma unroll
    for (int u = 0; u < NR; ++u)
      x[u] = *reinterpret_cast<const float4*>(vs + (int64_t)min(t0 + rg + 32 * u, Ti - 1) * lds_);
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int tt = t0 + rg + 32 * u;
      const float w = tt < len ? al[tt] : 0.f;
      acc.x += w * x[u].x; acc.y += w * x[u].y; acc.z += w * x[u].z; acc.w += w * x[u].w;
    }
  }
  *reinterpret_cast<float4*>(part + (rg * DC_CG + (tid & 7)) * 4) = acc;
  lds_barrier();
  if (tid < 4 * DC_CG) {
    const int c = cs * 4 * DC_CG + tid;
    if (c < CT) {
      float s = 0.f;
#pragma unroll 8
      for (int g = 0; g < 32; ++g) s += part[g * 4 * DC_CG + tid];
      p.ctx[((int64_t)par * p.B + b) * CT + c] = s;       // double-buffered by step parity
    }
  }
}

constexpr int DS_NT = 1024, DS_NW = DS_NT / 64;
constexpr int DS_RB = 4;        // cache rows per 16-lane group in flight
// One workgroup per (sample, head): the query of step t against cache rows 0..t.  HD = head depth (a multiple of 64:
// a row is covered by 16 lanes x HD/16 dims, four rows per wave at a time); dynamic LDS: s [Td] | part [32 * HD] | sm [16]
template <int HD>
__global__ __launch_bounds__(DS_NT) void dec_self_attn_k(const float* __restrict__ kvq, float* __restrict__ out,
                                                        const int* __restrict__ stepp, int Td, int D, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int DPL = HD / 16;                 // dims per lane (a multiple of 4)
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* s = smem;
  float* part = s + ((Td + 3) & ~3);
  float* sm = part + 32 * HD;
  const int t = *stepp;
  const float* base = kvq + (int64_t)b * Td * 3 * D + h * HD;       // K at +0, V at +D, Q at +2D of every row
  const int rsub = lane >> 4, dl = lane & 15;
  float q[DPL];
#pragma unroll
  for (int i = 0; i < DPL; i += 4) {
    const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)t * 3 * D + 2 * D + dl * DPL + i);
    q[i] = v.x; q[i + 1] = v.y; q[i + 2] = v.z; q[i + 3] = v.w;
  }
  // scores: 64 rows per pass over the workgroup, DS_RB passes in flight
  for (int j0 = wave * 4 + rsub; j0 <= t; j0 += 64 * DS_RB) {
    float4 kk[DS_RB][DPL / 4];
#pragma unroll
    for (int u = 0; u < DS_RB; ++u) {
      const int j = j0 + 64 * u;
#pragma unroll
      for (int i = 0; i < DPL / 4; ++i)       // branch-free: row clamped to t, the score of such a row is never stored
        kk[u][i] = *reinterpret_cast<const float4*>(base + (int64_t)min(j, t) * 3 * D + dl * DPL + 4 * i);
    }
#pragma unroll
    for (int u = 0; u < DS_RB; ++u) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < DPL / 4; ++i)
        acc += q[4 * i] * kk[u][i].x + q[4 * i + 1] * kk[u][i].y + q[4 * i + 2] * kk[u][i].z + q[4 * i + 3] * kk[u][i].w;
      SATT_DPP_ADD(acc, 0xB1); SATT_DPP_ADD(acc, 0x4E); SATT_DPP_ADD(acc, 0x141); SATT_DPP_ADD(acc, 0x140);   // 16-lane row sum
      const int j = j0 + 64 * u;
      if (dl == 0 && j <= t) s[j] = acc * scale;
    }
  }
  lds_barrier();
  float m = -INFINITY;
  for (int j = tid; j <= t; j += DS_NT) m = fmaxf(m, s[j]);
  m = block_max(m, sm, tid);
  float z = 0.f;
  for (int j = tid; j <= t; j += DS_NT) { const float e = __expf(s[j] - m); s[j] = e; z += e; }
  z = block_sum(z, sm, tid);
  const float rz = 1.f / z;
  // o = P V: thread = 4 dims x one of 1024 / (HD / 4) row groups
  constexpr int NC = HD / 4, NG = DS_NT / NC;
  const int c = tid % NC, g = tid / NC;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = g; j0 <= t; j0 += NG * 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + NG * u;
      v[u] = *reinterpret_cast<const float4*>(base + (int64_t)min(j, t) * 3 * D + D + 4 * c);       // weight pj is zero beyond t
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + NG * u;
      const float pj = j <= t ? s[j] : 0.f;
      acc.x += pj * v[u].x; acc.y += pj * v[u].y; acc.z += pj * v[u].z; acc.w += pj * v[u].w;
    }
  }
  *reinterpret_cast<float4*>(part + g * HD + 4 * c) = acc;
  lds_barrier();
  if (tid < HD) {
    float o = 0.f;
#pragma unroll 8
    for (int gg = 0; gg < NG; ++gg) o += part[gg * HD + tid];
    out[(int64_t)b * D + h * HD + tid] = o * rz;
  }
}

// any head depth (parity configurations): a wave per cache row, then a thread per (dim, row group)
__global__ __launch_bounds__(DS_NT) void dec_self_attn_any_k(const float* __restrict__ kvq, float* __restrict__ out,
                                                            const int* __restrict__ stepp, int Td, int D, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int hd = D / heads;
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* q = smem;
  float* s = q + hd;
  float* part = s + Td;
  float* sm = part + DS_NT;
  const int t = *stepp;
  const float* base = kvq + (int64_t)b * Td * 3 * D + h * hd;
  for (int i = tid; i < hd; i += DS_NT) q[i] = base[(int64_t)t * 3 * D + 2 * D + i];
  lds_barrier();
  for (int j = wave; j <= t; j += DS_NW) {
    float acc = 0.f;
    for (int d = lane; d < hd; d += 64) acc += q[d] * base[(int64_t)j * 3 * D + d];
    acc = wave_sum(acc);
    if (lane == 0) s[j] = acc * scale;
  }
  lds_barrier();
  float m = -INFINITY;
  for (int j = tid; j <= t; j += DS_NT) m = fmaxf(m, s[j]);
  m = block_max(m, sm, tid);
  float z = 0.f;
  for (int j = tid; j <= t; j += DS_NT) { const float e = __expf(s[j] - m); s[j] = e; z += e; }
  z = block_sum(z, sm, tid);
  const float rz = 1.f / z;
  const int ng = DS_NT / hd, d = tid % hd, g = tid / hd;
  float acc = 0.f;
  if (g < ng)
    for (int j = g; j <= t; j += ng) acc += s[j] * base[(int64_t)j * 3 * D + D + d];
  part[tid] = acc;
  lds_barrier();
  if (tid < hd) {
    float o = 0.f;
    for (int gg = 0; gg < ng; ++gg) o += part[gg * hd + tid];
    out[(int64_t)b * D + h * hd + tid] = o * rz;
  }
}

}  // namespace

// dropout fields: off (thresh 0), or a seed word, a positive step count and the plain (non-LSTM) form
static bool drop_ok(const satt_dec_linear_params& p) {
  return p.drop_thresh == 0u || (p.drop_seed && p.drop_T > 0 && !p.lstm_H);
}
extern "C" int satt_dec_linear(const satt_dec_linear_params* pp, void* stream) {
  if (!pp || !drop_ok(*pp)) return SATT_E_BADARG;
  const satt_dec_linear_params& p = *pp;
  if (p.B <= 0 || p.N <= 0 || p.nseg < 1 || p.nseg > 3 || !p.y || (!p.W && !p.Wb)) return SATT_E_BADARG;
  int K = 0;
  for (int s = 0; s < p.nseg; ++s) { if (!p.x[s] || p.k[s] <= 0) return SATT_E_BADARG; K += p.k[s]; }
  if (K > DL_KMAX) return SATT_E_UNSUPPORTED;
  if (p.act != SATT_ACT_NONE && p.act != SATT_ACT_RELU && p.act != SATT_ACT_TANH && p.act != SATT_ACT_SOFTSIGN) return SATT_E_BADARG;
  const bool bf = p.Wb != nullptr;
  // (N may end inside a 4-column group when the caller pads the weight rows with zero columns: stores stay bounded by N)
  const bool vec = (p.ldw % 4 == 0) && (((uintptr_t)(bf ? (const void*)p.Wb : (const void*)p.W)) % (bf ? 8 : 16) == 0);
  const int nb = p.B >= 8 ? 8 : (p.B >= 4 ? 4 : (p.B >= 2 ? 2 : 1));
  if (p.lstm_H) {
    if (p.N != 4 * p.lstm_H || p.lstm_H % 8 || !p.c_state || !p.h_state || !vec || p.act != SATT_ACT_NONE || p.res)
      return SATT_E_BADARG;
  }
  const dim3 grid((p.N + DL_COLS - 1) / DL_COLS, (p.B + nb - 1) / nb);
  hipStream_t s = (hipStream_t)stream;
#define SATT_DL(NBV)                                                                                         \
  do {                                                                                                       \
    if (bf) { if (vec) hipLaunchKernelGGL((dec_linear_k<NBV, true, true>), grid, dim3(DL_NT), 0, s, p);      \
              else hipLaunchKernelGGL((dec_linear_k<NBV, true, false>), grid, dim3(DL_NT), 0, s, p); }       \
    else { if (vec) hipLaunchKernelGGL((dec_linear_k<NBV, false, true>), grid, dim3(DL_NT), 0, s, p);        \
           else hipLaunchKernelGGL((dec_linear_k<NBV, false, false>), grid, dim3(DL_NT), 0, s, p); }         \
  } while (0)
  if (nb == 8) SATT_DL(8); else if (nb == 4) SATT_DL(4); else if (nb == 2) SATT_DL(2); else SATT_DL(1);
#undef SATT_DL
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

extern "C" int satt_dec_linear2(const satt_dec_linear_params* pa, const satt_dec_linear_params* pb, void* stream) {
  if (!pa || !pb || !drop_ok(*pa) || !drop_ok(*pb)) return SATT_E_BADARG;
  const satt_dec_linear_params& p1 = *pa; const satt_dec_linear_params& p2 = *pb;
  if (p1.B <= 0 || p2.B != p1.B || p1.nseg != 1 || p2.nseg != 1 || !p1.x[0] || !p2.y) return SATT_E_BADARG;
  if (p1.lstm_H || p2.lstm_H || p2.k[0] != p1.N) return SATT_E_BADARG;
  if (!p1.Wb || !p2.Wb) return SATT_E_UNSUPPORTED;                  // bf16 weights only (see the kernel)
  if (p1.N > 256 || p2.N > 256 || p1.k[0] > 256 || p1.ldw % 4 || p2.ldw % 4 || p1.ldw < 4 || p2.ldw < 4) return SATT_E_UNSUPPORTED;
  if ((uintptr_t)p1.Wb % 8 || (uintptr_t)p2.Wb % 8) return SATT_E_UNSUPPORTED;
  hipLaunchKernelGGL(dec_linear2_k, dim3(p1.B), dim3(D2_NT), 0, (hipStream_t)stream, p1, p2);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

extern "C" int satt_dec_linear_chain(const satt_dec_linear_params* pre, int npre, const satt_dec_linear_params* mainp,
                                     void* stream) {
  if (!pre || !mainp || npre < 1 || npre > 2 || !drop_ok(*mainp)) return SATT_E_BADARG;
  for (int j = 0; j < npre; ++j) if (!drop_ok(pre[j])) return SATT_E_BADARG;
  const satt_dec_linear_params& p = *mainp;
  if (p.B <= 0 || p.N <= 0 || p.nseg < 1 || p.nseg > 3 || !p.y) return SATT_E_BADARG;
  int K = 0;
  for (int s = 0; s < p.nseg; ++s) { if (!p.x[s] || p.k[s] <= 0) return SATT_E_BADARG; K += p.k[s]; }
  if (K > DL_KMAX) return SATT_E_UNSUPPORTED;
  if (!p.Wb || p.ldw % 4 || p.ldw < 4 || (uintptr_t)p.Wb % 8) return SATT_E_UNSUPPORTED;        // bf16 weights, 8-byte loads
  if (p.lstm_H) {
    if (p.N != 4 * p.lstm_H || p.lstm_H % 8 || !p.c_state || !p.h_state || p.act != SATT_ACT_NONE || p.res) return SATT_E_BADARG;
    if (p.step_out || p.stop) return SATT_E_BADARG;
  }
  for (int j = 0; j < npre; ++j) {
    const satt_dec_linear_params& q = pre[j];
    if (q.B != p.B || q.nseg != 1 || q.lstm_H || q.k[0] <= 0 || q.N <= 0 || (j == 0 && !q.x[0])) return SATT_E_BADARG;
    if (!q.Wb || q.ldw % 4 || q.ldw < 4 || (uintptr_t)q.Wb % 8 || q.k[0] > 256 || q.N > 256) return SATT_E_UNSUPPORTED;
    const int next_k = j + 1 < npre ? pre[j + 1].k[0] : p.k[0];
    if (next_k != q.N) return SATT_E_BADARG;          // the chained input is the next layer's (first) segment
  }
  const int nb = p.B >= 4 ? 4 : (p.B >= 2 ? 2 : 1);
  const dim3 grid((p.N + DL_COLS - 1) / DL_COLS, (p.B + nb - 1) / nb);
  hipStream_t s = (hipStream_t)stream;
  const satt_dec_linear_params& q0 = pre[0];
  const satt_dec_linear_params& q1 = pre[npre - 1];
#define SATT_DCH(NBV)                                                                                              \
  do {                                                                                                             \
    if (npre == 2) hipLaunchKernelGGL((dec_chain_k<NBV, 2>), grid, dim3(DCH_NT), 0, s, q0, q1, p);                 \
    else hipLaunchKernelGGL((dec_chain_k<NBV, 1>), grid, dim3(DCH_NT), 0, s, q0, q1, p);                           \
  } while (0)
  if (nb == 4) SATT_DCH(4); else if (nb == 2) SATT_DCH(2); else SATT_DCH(1);
#undef SATT_DCH
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

extern "C" int satt_dec_attention(const satt_dec_attention_params* pp, void* stream) {
  if (!pp) return SATT_E_BADARG;
  const satt_dec_attention_params& p = *pp;
  const bool forced = p.teach1 != nullptr;
  if (p.B <= 0 || p.Ti <= 0 || p.Td <= 0 || p.kernel < 1 || !p.values1 || !p.ctx || !p.align1 || !p.a_state ||
      !p.alpha_state || !p.lengths || p.att1_mode < 0 || p.att1_mode > 1 || p.A <= 0)
    return SATT_E_BADARG;
  if (!forced && (!p.keys1 || !p.hq || (!p.Wq && !p.Wqb) || !p.e1 || (p.U2 > 0 && !p.e2))) return SATT_E_BADARG;
  if ((p.U2 > 0) != (p.V2 > 0) || (p.U2 > 0 && (!p.values2 || (!forced && (!p.keys2 || !p.v2))))) return SATT_E_BADARG;
  if (p.agentW && (!p.agentb || !p.pq_out)) return SATT_E_BADARG;
  if (p.filters != 5 || p.U1 > 256 || p.U2 > 64 || p.U1 + p.U2 > 256 || p.A > 256 || p.A % 4) return SATT_E_UNSUPPORTED;
  if (p.U1 % 4 || p.U2 % 4 || p.V1 % 4 || p.V2 % 4) return SATT_E_UNSUPPORTED;       // 16-byte rows
  hipStream_t s = (hipStream_t)stream;
  const int UQ = p.U1 + p.U2, CT = p.V1 + p.V2;
  if (!forced) {
    // slices of at most 2 * DE_NW rows (two row passes in registers), at least 8 rows: up to 64 workgroups per sample
    int NS = (p.Ti + 7) / 8;
    if (NS > 64) NS = 64;
    while (de_rows(p.Ti, NS) > 2 * DE_NW) ++NS;
    const int R = de_rows(p.Ti, NS);
    const size_t smem = sizeof(float) * ((size_t)p.A + DE_NW * UQ + UQ + p.kernel * 5 + 5 + R + p.kernel + R * 5 + 8);
    if (smem > 64 * 1024 || R + p.kernel > DE_NT || p.kernel * 5 + 5 > DE_NT || p.A > DE_NT) return SATT_E_UNSUPPORTED;
    if (p.Wqb) hipLaunchKernelGGL((dec_attn_energy_k<5, true>), dim3(p.B, NS), dim3(DE_NT), smem, s, p, NS);
    else hipLaunchKernelGGL((dec_attn_energy_k<5, false>), dim3(p.B, NS), dim3(DE_NT), smem, s, p, NS);
    SATT_LAUNCH_CHECK();
  }
  const int NC = (CT / 4 + DC_CG - 1) / DC_CG;
  const size_t smem = sizeof(float) * (4 * (size_t)p.Ti + 32 * 4 * DC_CG + 8);
  if (smem > 160 * 1024) return SATT_E_UNSUPPORTED;
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)dec_attn_context_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(dec_attn_context_k, dim3(p.B, NC), dim3(DC_NT), smem, s, p);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

extern "C" int satt_dec_self_attn(const float* kvq, float* out, const int* step, int B, int Td, int D, int heads, float scale,
                                  void* stream) {
  if (!kvq || !out || !step || B <= 0 || Td <= 0 || heads <= 0 || D <= 0 || D % heads) return SATT_E_BADARG;
  const int hd = D / heads;
  hipStream_t s = (hipStream_t)stream;
  if (hd == 128 && D % 4 == 0) {
    const size_t smem = sizeof(float) * ((size_t)((Td + 3) & ~3) + 32 * 128 + DS_NW);
    if (smem > 160 * 1024) return SATT_E_UNSUPPORTED;
    if (smem > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)dec_self_attn_k<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(dec_self_attn_k<128>, dim3(B * heads), dim3(DS_NT), smem, s, kvq, out, step, Td, D, heads, scale);
  } else {
    if (hd > DS_NT || DS_NT % hd) return SATT_E_UNSUPPORTED;
    const size_t smem = sizeof(float) * ((size_t)hd + Td + DS_NT + DS_NW);
    if (smem > 160 * 1024) return SATT_E_UNSUPPORTED;
    if (smem > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)dec_self_attn_any_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(dec_self_attn_any_k, dim3(B * heads), dim3(DS_NT), smem, s, kvq, out, step, Td, D, heads, scale);
  }
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}
