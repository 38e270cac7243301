// Cluster form of the dual-source attention RNN loop: C workgroups per sample (grid (B, C): with B % 8 == 0 the
// cluster of a sample sits on one XCD).  The two dominant costs of the single-workgroup kernel are split C ways,
// everything cheap stays REDUNDANT and bitwise identical in every member of the cluster, so only few exchanges
// per step are needed (8-byte {tag,value} granules, parity double-buffered, bounded spins — see lstm_cluster.hip):
//   forward : [ctx|h] x Wrec split by gate columns (REGISTER-resident MFMA operands, see below)
//             h'_own x Wq[own rows] = partial processed query -> X1: all-gather h_state (A) + partials (C x UQ)
//             energies split by memory rows (t' mod C) -> X2: all-gather (e1, e2)             (2 len floats)
//   backward: d alpha split by memory rows          -> Xb: all-gather (d alpha, d a2)         (2 Ti floats)
//             energy backward split by memory rows  -> Xd: all-reduce d pq (C partials x UQ) + all-gather dfl rows
//             dz x Wrec^T split by output columns   -> Xh: all-gather d[ctx|h]                (CT+A floats)
// Forward weight slices are pre-packed per cluster member (satt_attn_cluster_pack) in MFMA B-operand order and are
// loaded ONCE per launch into registers (128 VGPRs per lane for the [512 x 256] bf16 slice of the full model): the
// recurrent mat-vec is then 32 v_mfma_f32_16x16x32_bf16 per wave and step, with the fp32 input vector split
// exactly into three bf16 rows (hi/mid/lo) of the otherwise empty 16-row A operand, so the product is fp32-exact
// for bf16 weights.  No weight byte is re-read from L2 inside the time loop.
// The backward slice (Wrec^T) is streamed from L2 as a contiguous [G][NWP] bf16 matrix.
#include <type_traits>
#include "attn_common.h"
#include "mfma_rec.h"
#include "cluster_xchg.h"
#ifdef SATT_CLUSTER_JITTER
#define lds_barrier() do { lds_barrier(); cluster_jitter(); } while (0)
#endif
#ifdef SATT_PROFILE
static __device__ unsigned long long satt_prolog[8];
#define PLOG(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) satt_prolog[i] = wall_clock64(); } while (0)
#else
#define PLOG(i)
#endif
// backward time-stamp trace (scratch/prof_attn.py, SATT_TRACE_BWD=1): shares the forward trace buffer
#ifdef SATT_TRACE_BWD
#define BTRACE(step, slot) TRACE(step, slot)
#else
#define BTRACE(step, slot)
#endif

namespace {

// Outputs that OTHER streams consume while the kernel is still running (after a chunk signal) are written with agent-scope
// (write-through) stores, and every wave waits for its own stores before the chunk is counted.  The alternative - plain
// stores + __threadfence() at the chunk boundary - writes back the whole L2 of the XCD for each of the 128 workgroups:
// measured 0.32 ms per step for the 16 boundaries of the two launches.
#ifdef SATT_CHUNK_FENCE     // A/B switch: the fenced form
__device__ __forceinline__ void gst(float* p, float v) { *p = v; }
__device__ __forceinline__ void chunk_release() { __threadfence(); }
#else
__device__ __forceinline__ void gst(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void chunk_release() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }   // own stores have reached memory
#endif

// register-resident forward slice: a wave owns MNTW tiles of 16 gate columns (NL <= 16 * MNTW * AW) and all K tiles
// (32 rows each) of them; MNTW * MKT * 4 accumulation registers per lane hold it.
__host__ __device__ constexpr int mkt_of(int mntw) { return mntw == 1 ? 18 : 13; }
// K tiles of the forward slice kept in LDS (always an even count: they are consumed in pairs), and the row stride
// (in tiles) of the split input vector: every register tile and every LDS tile is multiplied unconditionally
__host__ __device__ constexpr int ktl_of(int KT, int mntw) { const int r = KT > mkt_of(mntw) ? KT - mkt_of(mntw) : 0; return (r + 1) & ~1; }
__host__ __device__ constexpr int xs_tiles(int KT, int mntw) { return mkt_of(mntw) + ktl_of(KT, mntw); }
__host__ __device__ constexpr int mntw_of(int NL) { return (NL + 16 * AW - 1) / (16 * AW); }
constexpr int MNTQ = 2;     // N tiles per wave of the partial processed query: UQ <= 16 * MNTQ * AW = 256
constexpr int RBF = 5;      // memory rows per wave iteration in the forward energies (one pass for <= 40 own rows)
constexpr float TS = 2.885390081777927f;   // 2 * log2(e)
typedef __attribute__((ext_vector_type(2))) float v2f;
__host__ __device__ constexpr int kt_of(int K) { return (K + 31) / 32; }

struct WsLayout {   // granule offsets (per sample, per parity) inside the workspace
  int x1, x2, x3, xb, xd, xh, xi, per_parity;
};
constexpr int NSC = 8;      // scalar slots appended to every member's partial context (m1, s1, sg1, m2, s2)
__host__ __device__ constexpr int nwp_of(int K, int C) { return (((K + C - 1) / C) + 7) & ~7; }
__host__ __device__ constexpr WsLayout ws_layout(int A, int Ti, int C, int UQ, int F, int K) {
  WsLayout w{}; int o = 0;
  w.x1 = o; o += A + C * UQ;
  w.x2 = o; o += 2 * Ti;
  w.x3 = o; o += C * (K - A + NSC);   // partial contexts (CT = K - A) + scalars of every member
  w.xb = o; o += 2 * Ti;
  w.xd = o; o += C * UQ + Ti * (F + 1);   // partial d pq | per row: F d fl values + the d w carry
  w.xh = o; o += C * K;
  w.xi = o; o += C;                    // XCC ids of the members (start-up handshake)
  w.per_parity = o;
  return w;
}

// The workspace is sized (and its sticky 64-byte tail placed) by the FULL layout - K = V1 + V2 + A rows of the recurrent input - at
// ws_ti(): the specialised bf16 kernels lay their granules out for Ti = 32 FKT whatever the launch's Ti (compile-time offsets, r5).
// Every kernel addresses the tail through ws_tail_words() - its own granule layout may be smaller (the folded forward kernel exchanges
// ctx2 | h only: until r5 it derived the tail from THAT layout and raised its error word where the host never looked).
// (the predicate is the SPECIALISATION alone, not the precision mode: an engine keeps its workspace when the precision is switched)
__host__ __device__ constexpr int ws_ti(int Ti, bool spec) { return (spec && Ti < 160) ? 160 : Ti; }
__host__ __device__ constexpr int64_t ws_tail_words(int B, int A, int Ti, int C, int UQ, int F, int CT, bool spec) {
  return (int64_t)2 * B * ws_layout(A, ws_ti(Ti, spec), C, UQ, F, CT + A).per_parity;
}

struct SmemCF {
  int xs, hs, gs, us, z, dpart, tab, aprev, alA, alB, u1, u2, eo1, eo2, eo3, fl, Fs, bFs, cg, dead, wl, kofs, vofs, als, wv, ls, las, total;
};
// FOLD (r3): the first source's context never enters the recurrent product as a vector.  gates += ctx1 Wc1 with ctx1 = alpha V1 is
// evaluated as alpha (V1 Wc1): the engine precomputes VW1 = values1 x Wrec[ctx1 rows] ([Ti, 4A] per sample, one GEMM per step), each
// member keeps its own gate columns of it as bf16 MFMA B tiles in LDS (FKT K tiles of 32 memory rows), and the normalised
// alignments - which every member holds after the exchange X2 anyway - are the A operand.  The partial-context MFMA of the
// first source, its 4 x 256 exchange granules and the context assembly in the normalisation leave the step's dependency chain
// (-0.5 us per step); ctx1 itself (an output: LSTM1's input, the backward pass) becomes one batched GEMM per pipeline chunk
// OUTSIDE the kernel (engine.py).  The backward kernel is untouched: it differentiates the same function in its unfolded form.
constexpr int FKT = 5;        // K tiles of the folded product: Ti <= 160
static_assert(ws_ti(1, true) == 32 * FKT, "the workspace is sized for the fixed layouts of the folded / saved-factor kernels");
// LOCM (r4, folded kernel): the location term of the energies, L[t', u] = sum_k fl[t', k] (TS U[k, u]), on the matrix cores.  It is
// formed for the own rows right behind the location convolution - inside the exchange window X1, where the waves otherwise
// poll - as [16 rows x 32] x [32 x 16 units] bf16 MFMAs whose 32 K slots carry the 5 filters three times (fl_hi U_hi, fl_lo U_hi,
// fl_hi U_lo: ~2^-16 relative), and kept in LDS as fp32 rows; the energy rows - the longest phase of the forward step, VALU
// bound - then read 4 values instead of issuing 5 LDS broadcasts + 10 packed FMAs per row and lane.
constexpr int LOC_ROWS = 48;  // 3 M tiles of own rows (Ti <= 160)
__host__ __device__ constexpr int loc_stride(int U1) { return U1 + 4; }     // (+4: the four row groups of a D tile hit distinct banks)
// KTL: K tiles of the forward slice kept in LDS (the ones that do not fit the accumulation registers)
__host__ __device__ constexpr SmemCF carve_cf(int A, int CT, int UQ, int Ti, int F, int KW, int NL, int nown, bool klds, int foldV1 = 0) {
  constexpr auto u = [](int x) constexpr { return (x + 3) & ~3; };
  const bool fold = foldV1 > 0;
  const int CTF = CT - foldV1;                    // context columns that enter the recurrent product / travel in the exchange
  const int C = 4 * A / NL, KT = kt_of(CTF + A), mntw = mntw_of(NL);
  const int KTL = ktl_of(KT, mntw), KTO = kt_of(nown);
  SmemCF s{}; int o = 0;
  s.xs = o; o += 4 * a_stride(xs_tiles(KT, mntw)) / 2;      // bf16 [4][XS]
  s.hs = o; o += 4 * a_stride(kt_of(A) < 2 ? 2 : kt_of(A)) / 2;   // bf16 [4][HS]
  s.gs = o; o += 4 * a_stride(KTO) / 2;           // bf16 [4][GS] split g = w * u1 of the own rows
  s.us = o; o += 4 * a_stride(KTO) / 2;           // bf16 [4][GS] split u2 of the own rows
  s.z = o; o += u(NL); s.dpart = o; o += u(C * UQ);
  s.tab = o; o += (2 + F) * 64 * NQ + 64 + 4;
  s.aprev = o; o += u(Ti + KW); s.alA = o; o += u(Ti); s.alB = o; o += u(Ti); s.u1 = o; o += u(Ti); s.u2 = o; o += u(Ti);
  s.eo1 = o; o += u(nown); s.eo2 = o; o += u(nown); s.eo3 = o; o += u(nown);
  s.fl = o; o += u(Ti * F); s.Fs = o; o += u(KW * F); s.bFs = o; o += u(F);
  s.cg = o; o += u(C * (CTF + NSC));
  s.dead = o; o += 12;                          // [0]: timeout flag; [4..11]: transition-agent scalars
  s.wl = o; o += AW * mntw * KTL * 64 * 4;       // [AW][MNTW][KTL][64 lanes][16 B]
  s.kofs = o; if (klds) o += u((nown * UQ + 1) / 2);
  s.vofs = o; if (klds) o += KTO * ((CTF + 15) / 16) * 64 * 4;  // own value rows as MFMA B tiles [KTO][NTV][64][16 B]
  s.als = o; if (fold) o += 4 * a_stride(FKT) / 2;              // bf16 [4][ALS] split normalised alignments of ALL rows
  s.wv = o;                                                     // (the VW1 tiles live in accumulation registers)
  s.ls = o; if (fold) o += u(LOC_ROWS * loc_stride(UQ - (CT - A)));   // LOCM rows (U1 = UQ - U2 and U2 == V2 in both specialisations)
  s.las = o; if (fold) o += LOC_ROWS * 32 / 2;                          // LOCM A operands: bf16 [LOC_ROWS][32 K slots]
  s.total = o;
  return s;
}

// SPEC: the model dimensions are the compile-time constants of SpecDimsOf<SPEC> (cluster of 4; 1 = the LJSpeech / VCTK
// self-attention Tacotron, 2 = the baseline Tacotron with its single attention source; 0 = run-time dimensions):
// LDS offsets, loop bounds and strides become immediates, which removes most of the scalar-register pressure
// (hundreds of spilled scalars were reloaded per step) and of the per-step address arithmetic.
template <int S> struct SpecDimsOf { static constexpr int C = 4, A = 256, V1 = 256, V2 = 32, U1 = 224, U2 = 32, KW = 10; };
template <> struct SpecDimsOf<2> { static constexpr int C = 4, A = 256, V1 = 256, V2 = 0, U1 = 256, U2 = 0, KW = 10; };
typedef SpecDimsOf<1> SpecDims;
template <int S> __host__ inline bool spec_dims_are(const satt_attn_rnn_params& p, int C) {
  typedef SpecDimsOf<S> D;
  return C == D::C && p.A == D::A && p.V1 == D::V1 && p.V2 == D::V2 && p.U1 == D::U1 && p.U2 == D::U2 && p.kernel == D::KW &&
         p.agentW == nullptr;
}
__host__ inline int spec_dims(const satt_attn_rnn_params& p, int C) {      // which specialisation fits (0: none)
  return spec_dims_are<1>(p, C) ? 1 : spec_dims_are<2>(p, C) ? 2 : 0;
}

template <int F, bool KLDS, int MNTW, int SPEC, bool FOLD = false>
__global__ __launch_bounds__(ANT) void attn_cluster_fwd_k(const satt_attn_cluster_params cp) {
  static_assert(!FOLD || (SPEC != 0 && KLDS && MNTW == 2), "the folded form exists for the specialised bf16 kernel");
  constexpr int MKT = FOLD ? (SpecDimsOf<SPEC>::V2 + SpecDimsOf<SPEC>::A + 31) / 32 : mkt_of(MNTW);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const satt_attn_rnn_params& p = cp.f;
  const int C = SPEC ? SpecDimsOf<SPEC>::C : cp.C;
  const int A = SPEC ? SpecDimsOf<SPEC>::A : p.A, G = 4 * A, V1 = SPEC ? SpecDimsOf<SPEC>::V1 : p.V1, V2 = SPEC ? SpecDimsOf<SPEC>::V2 : p.V2;
  const int CT = V1 + V2, U1 = SPEC ? SpecDimsOf<SPEC>::U1 : p.U1, U2 = SPEC ? SpecDimsOf<SPEC>::U2 : p.U2, UQ = U1 + U2;
  const int Ti = p.Ti, Td = p.Td, KW = SPEC ? SpecDimsOf<SPEC>::KW : p.kernel, PL = (KW - 1) / 2;
  // CTF: the context columns that enter the recurrent product and travel through the exchange (FOLD: the second source's only),
  // C0 = first of them within [ctx1 | ctx2]
  const int CTF = FOLD ? V2 : CT, C0 = CT - CTF;
  const int AU = A / C, NL = 4 * AU, KR = CTF + A;
  const int KT = kt_of(KR), XS = a_stride(FOLD ? MKT : xs_tiles(KT, MNTW)), KTQ = kt_of(AU), HS = a_stride(kt_of(A) < 2 ? 2 : kt_of(A));
  const int KTL = FOLD ? 0 : ktl_of(KT, MNTW);   // K tiles MKT.. of the slice live in LDS (even count, zero padded)
  const int b = blockIdx.x, c = blockIdx.y;
  // FOLD (r5): the LDS layout is the one of Ti = 32 FKT whatever the launch's Ti (<= 32 FKT) - every offset a compile-time constant,
  // i.e. an immediate of the LDS instruction.  With run-time offsets ~40 array bases lived in scalar registers, a hundred of
  // them spilled: a v_readlane (a VALU slot, plus its hazard nop) and a vector add in front of most LDS accesses of the step.
  typedef SpecDimsOf<SPEC> DL;
  constexpr int TIL = 32 * FKT, NOWNL = (TIL + DL::C - 1) / DL::C;
  constexpr SmemCF LFIX = carve_cf(DL::A, DL::V1 + DL::V2, DL::U1 + DL::U2, TIL, F, DL::KW, 4 * (DL::A / DL::C), NOWNL, KLDS, FOLD ? DL::V1 : 0);
  const int nown_max = FOLD ? NOWNL : (Ti + C - 1) / C, KTO = kt_of(nown_max), GS = a_stride(KTO), NTV = (CTF + 15) / 16;
  const int ALS = a_stride(FKT);
  const SmemCF L = FOLD ? LFIX : carve_cf(A, CT, UQ, Ti, F, KW, NL, nown_max, KLDS, FOLD ? V1 : 0);
  uint16_t* xs = reinterpret_cast<uint16_t*>(smem + L.xs);   // bf16 [4][XS]: split [ctx1 | ctx2 | h_state], row 3 = 0
  uint16_t* hs = reinterpret_cast<uint16_t*>(smem + L.hs);   // bf16 [4][HS]: split own h' units
  uint16_t* gs = reinterpret_cast<uint16_t*>(smem + L.gs);   // bf16 [4][GS]: split w*u1 of the own rows
  uint16_t* us = reinterpret_cast<uint16_t*>(smem + L.us);   // bf16 [4][GS]: split u2 of the own rows
  float* z = smem + L.z;            // [NL]      own gate pre-activations
  float* dpart = smem + L.dpart;    // [C][UQ]   partial processed queries of every member (full after X1)
  float* tab = smem + L.tab;        // v1[256] | b1[256] | U[F][256] | v2[64]  (per-lane attention parameters)
  float* aprev = smem + L.aprev + PL;   // [-PL, Ti + KW - PL): a1_{t-1} (input of the location conv), zero borders
  float* alA = smem + L.alA;        // [Ti] alignment ping-pong
  float* alB = smem + L.alB;
  float* u1 = smem + L.u1;          // [Ti] exp(e1 - m_member) of every row (full after X2)
  float* u2 = smem + L.u2;
  float* eo1 = smem + L.eo1;        // [nown] energies of the own rows
  float* eo2 = smem + L.eo2;
  float* eo3 = smem + L.eo3;
  float* fl = smem + L.fl;          // [Ti*F] (own rows only are valid)
  float* Fs = smem + L.Fs;
  float* bFs = smem + L.bFs;
  float* cg = smem + L.cg;          // [C][CT + NSC] partial contexts + scalars of every member (full after X2)
  int* dead = reinterpret_cast<int*>(smem + L.dead);
  i32x4_t* Wl = reinterpret_cast<i32x4_t*>(smem + L.wl);
  uint16_t* K1s = reinterpret_cast<uint16_t*>(smem + L.kofs);   // bf16 [nown][U1]  (local row i <-> t' = c + C*i)
  uint16_t* K2s = K1s + nown_max * U1;
  i32x4_t* Vt = reinterpret_cast<i32x4_t*>(smem + L.vofs);      // bf16 B tiles [KTO][NTV][64]
  uint16_t* als = reinterpret_cast<uint16_t*>(smem + L.als);    // FOLD: bf16 [4][ALS] split alpha_{t-1} of all memory rows
  float* lsm = smem + L.ls;                                      // FOLD: LOCM rows [LOC_ROWS][LSTR] (see LOC_ROWS)
  uint16_t* las = reinterpret_cast<uint16_t*>(smem + L.las);    // FOLD: LOCM A operands [LOC_ROWS][32] (slots >= 15 stay zero)
  const int LSTR = loc_stride(U1);

  const int len = (int)p.lengths[b];
  const uint32_t seed = p.seed ? *p.seed : 0u;
  const float* xg = p.xg + (size_t)b * Td * G;
  const float* keys1 = p.keys1 + (size_t)b * Ti * U1;
  const float* values1 = p.values1 + (size_t)b * Ti * V1;
  const float* keys2 = p.keys2 + (size_t)b * Ti * U2;
  const float* values2 = p.values2 + (size_t)b * Ti * V2;
  const int OW = A + CT;
  float* out = p.out + (size_t)b * Td * OW;
  uint16_t* const saf = reinterpret_cast<uint16_t*>(p.saf);    // fp16 [B,Td,Ti,UQ] derivative factors for the backward pass (FOLD)
  constexpr WsLayout WLFIX = ws_layout(DL::A, TIL, DL::C, DL::U1 + DL::U2, F, (FOLD ? DL::V2 : DL::V1 + DL::V2) + DL::A);
  const WsLayout WL = FOLD ? WLFIX : ws_layout(A, Ti, C, UQ, F, KR);
  u64* wsb = reinterpret_cast<u64*>(cp.ws);
  unsigned int* err_word = reinterpret_cast<unsigned int*>(wsb + ws_tail_words(p.B, A, Ti, C, UQ, F, CT, SPEC != 0));
  const int nown = len > c ? (len - c + C - 1) / C : 0;         // own memory rows: t' = c + C*i < len

  // 8 consecutive own-row values of context column `col` (rows io0.. of the own-row index), fp32
  auto load_v8 = [&](int io0, int col, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int tt = c + C * (io0 + i);
      float x = 0.f;
      const int cc = col + C0;                     // column within [ctx1 | ctx2]
      if (tt < len && col < CTF) x = cc < V1 ? values1[(size_t)tt * V1 + cc] : values2[(size_t)tt * V2 + (cc - V1)];
      v[i] = x;
    }
  };

  PLOG(0);
  // register-resident weights, pinned to the accumulation half of the register file.  B operand of tile
  // (nt = wave*MNTW + j, kt): lane l holds rows kt*32 + (l>>4)*8 .. +8 of local column nt*16 + (l&15).
  // Packed by satt_attn_cluster_pack as [C][AW][MNTW][KT][64][8] bf16; tiles kt >= MKT go to LDS.
  i32x4_t wreg[MNTW][MKT];
  i32x4_t wvr[MNTW][FKT];   // FOLD: own gate columns of VW1 (B tiles of the folded context product), accumulation registers too
  i32x4_t wq[MNTQ][2];           // own rows of Wq (rows c*AU .. +AU): tile (nt = wave*MNTQ + j, kt < 2)
  {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const i32x4_t* wsrc = reinterpret_cast<const i32x4_t*>(cp.WrecP) + (size_t)(c * AW + wave) * MNTW * KT * 64 + lane;
    // loads are issued in batches of WB tuples before they are pinned: a pin needs the value, i.e. one L2 round trip
    // per batch instead of one per tuple
    constexpr int WB = 8;
#pragma unroll
    for (int q0 = 0; q0 < MNTW * MKT; q0 += WB) {
      i32x4_t tmp[WB];
#pragma unroll
      for (int q = 0; q < WB; ++q) {
        const int idx = q0 + q, j = idx / MKT, kt = idx - j * MKT;
        tmp[q] = (idx < MNTW * MKT && kt < KT) ? wsrc[(size_t)(j * KT + kt) * 64] : (i32x4_t){0, 0, 0, 0};
      }
#pragma unroll
      for (int q = 0; q < WB; ++q) {
        const int idx = q0 + q, j = idx / MKT, kt = idx - j * MKT;
        if (idx < MNTW * MKT) { asm volatile("" : "+a"(tmp[q])); wreg[j][kt] = tmp[q]; }
      }
    }
    for (int q0 = 0; q0 < MNTW * KTL; q0 += WB) {            // LDS-resident tiles: WB loads in flight, then the stores
      i32x4_t tmp[WB];
#pragma unroll
      for (int q = 0; q < WB; ++q) {
        const int idx = q0 + q, j = idx / max(KTL, 1), kl = idx - j * max(KTL, 1);
        tmp[q] = (idx < MNTW * KTL && (MKT + kl) < KT) ? wsrc[(size_t)(j * KT + MKT + kl) * 64] : (i32x4_t){0, 0, 0, 0};
      }
#pragma unroll
      for (int q = 0; q < WB; ++q) {
        const int idx = q0 + q, j = idx / max(KTL, 1), kl = idx - j * max(KTL, 1);
        if (idx < MNTW * KTL) Wl[((wave * MNTW + j) * KTL + kl) * 64 + lane] = tmp[q];
      }
    }
    {   // query-layer slice: branch-free 2-byte loads, all in flight before the first use (`ok ? W[..] : 0` is a load inside
        // a divergent branch, waited for at the end of the branch: one L2 round trip per element pair)
      uint16_t raw[MNTQ][2][8];
#pragma unroll
      for (int j = 0; j < MNTQ; ++j)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const int n = min((wave * MNTQ + j) * 16 + (lane & 15), UQ - 1);
#pragma unroll
          for (int i = 0; i < 8; ++i) raw[j][kt][i] = p.Wq[(size_t)(c * AU + min(kt * 32 + (lane >> 4) * 8 + i, AU - 1)) * UQ + n];
        }
#pragma unroll
      for (int j = 0; j < MNTQ; ++j)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const int n = (wave * MNTQ + j) * 16 + (lane & 15);
          i32x4_t w = (i32x4_t){0, 0, 0, 0};
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int k = kt * 32 + (lane >> 4) * 8 + i;
            const uint32_t v = (k < AU && n < UQ) ? (uint32_t)raw[j][kt][i] : 0u;
            w[i >> 1] |= (int)(v << ((i & 1) * 16));
          }
          asm volatile("" : "+a"(w));
          wq[j][kt] = w;
        }
    }
    // energies use tanh(x) = 1 - 2 / (1 + exp2(TS * x)): v is stored as -2 v, U as TS * U, so the
    // inner loop is  x' = TS * key + pq' + sum_k f_k U'_k ;  acc += v' / (1 + exp2(x'))  and  e = sum(v) + acc
    PLOG(1);
    for (int i = tid; i < 64 * NQ; i += ANT) {
      tab[i] = i < U1 ? -2.f * p.v1[i] : 0.f;
      tab[64 * NQ + i] = i < U1 ? p.b1[i] : 0.f;
      for (int k = 0; k < F; ++k) tab[(2 + k) * 64 * NQ + i] = i < U1 ? TS * p.locU[k * U1 + i] : 0.f;
    }
    if (tid < 64) tab[(2 + F) * 64 * NQ + tid] = tid < U2 ? -2.f * p.v2[tid] : 0.f;
    if (tid < 64) {                     // sum(v), sum|v| of both mechanisms (wave 0)
      float s1 = 0.f, s2 = 0.f, a1 = 0.f, a2 = 0.f;
      for (int i = tid; i < U1; i += 64) { s1 += p.v1[i]; a1 += fabsf(p.v1[i]); }
      for (int i = tid; i < U2; i += 64) { s2 += p.v2[i]; a2 += fabsf(p.v2[i]); }
      s1 = wave_sum(s1); s2 = wave_sum(s2); a1 = wave_sum(a1); a2 = wave_sum(a2);
      if (tid == 0) {
        float* tc = tab + (2 + F) * 64 * NQ + 64;
        tc[0] = s1; tc[1] = s2; tc[2] = a1; tc[3] = a2;
      }
    }
    for (int i = tid; i < 4 * XS; i += ANT) xs[i] = 0;
    for (int i = tid; i < 4 * HS; i += ANT) hs[i] = 0;
    for (int i = tid; i < 4 * GS; i += ANT) { gs[i] = 0; us[i] = 0; }
    for (int i = tid; i < Ti; i += ANT) { alA[i] = (i == 0) ? 1.f : 0.f; alB[i] = 0.f; u1[i] = 0.f; u2[i] = 0.f; }
    for (int i = tid; i < Ti + KW; i += ANT) aprev[i - PL] = 0.f;
    for (int i = tid; i < KW * F; i += ANT) Fs[i] = p.locF[i];
    if (tid < F) bFs[tid] = p.locFb[tid];
    if (tid == 0) *dead = 0;
    if (FOLD) for (int i = tid; i < LOC_ROWS * 32; i += ANT) las[i] = 0;
    PLOG(2);
    if (KLDS) {
#pragma unroll 4
      for (int e = tid; e < nown * U1; e += ANT) { const int i = e / U1, d = e - i * U1; K1s[e] = f2bf(keys1[(size_t)(c + C * i) * U1 + d]); }
      for (int e = tid; e < nown * U2; e += ANT) { const int i = e / U2, d = e - i * U2; K2s[e] = f2bf(keys2[(size_t)(c + C * i) * U2 + d]); }
      for (int e = tid; e < KTO * NTV * 64; e += ANT) {          // own value rows as bf16 MFMA B tiles
        const int l = e & 63, tile = e >> 6, nt = tile % NTV, kt = tile / NTV;
        float v[8];
        load_v8(kt * 32 + (l >> 4) * 8, nt * 16 + (l & 15), v);
        i32x4_t w;
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = (int)((uint32_t)f2bf(v[2 * q]) | ((uint32_t)f2bf(v[2 * q + 1]) << 16));
        Vt[e] = w;
      }
    }
    if (FOLD) {
      for (int i = tid; i < 4 * ALS; i += ANT) als[i] = 0;
      // (all zero at t = 0: the reference starts from a zero context, not from alpha_{-1} V)
      // own gate columns of VW1 = values1 x Wrec[ctx1 rows] as bf16 B tiles in the accumulation registers the shorter recurrent
      // slice leaves free: lane l of tile (j, kt) holds memory rows kt*32 + (l>>4)*8 .. +8 of local column
      // (wave*MNTW + j)*16 + (l & 15); rows >= Ti are zero.  All 80 loads of the lane are in flight before the first pin.
      const float* vw = cp.vw1 + (size_t)b * Ti * G;
#pragma unroll
      for (int j = 0; j < MNTW; ++j) {
        const int n = (wave * MNTW + j) * 16 + (lane & 15), gq = n / AU, uq = n - gq * AU;
        const float* col = vw + gq * A + c * AU + uq;
#pragma unroll
        for (int kt = 0; kt < FKT; ++kt) {        // one tile at a time: 8 loads in flight (the prologue runs once per launch)
          float v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = col[(size_t)min(kt * 32 + (lane >> 4) * 8 + i, Ti - 1) * G];
          i32x4_t w;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int t0 = kt * 32 + (lane >> 4) * 8 + 2 * q;
            const uint32_t lo = t0 < Ti ? (uint32_t)f2bf(v[2 * q]) : 0u, hi = t0 + 1 < Ti ? (uint32_t)f2bf(v[2 * q + 1]) : 0u;
            w[q] = (int)(lo | (hi << 16));
          }
          asm volatile("" : "+a"(w));
          wvr[j][kt] = w;
        }
      }
    }
  }
  __syncthreads();
  PLOG(3);
  // start-up handshake: are all members of this cluster on one XCD?  (granules with a tag no step can produce)
  {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* flagp = dead + 1;
    if (tid == 0) gput(wsb + (size_t)b * WL.per_parity + WL.xi + c, XCC_TAG, __int_as_float(xcc_id()), false);
    if (wave == 0) {
      int mism = 0;
      const int mine = xcc_id();
      gather_span(wsb + (size_t)b * WL.per_parity + WL.xi, C, XCC_TAG, 0, 1, lane,
                  [&](int i, float v) { if (__float_as_int(v) != mine) mism = 1; }, err_word, dead);
      mism = __any(mism) || *dead;
      if (lane == 0) *flagp = mism ? 0 : 1;
    }
    __syncthreads();
  }
  PLOG(4);
  const bool same_xcd = dead[1] != 0;
  // words 1 / 2 behind the error word count the workgroup-launches that publish with plain (same-XCD) stores / with
  // write-through stores: tests read them through satt_attn_cluster_fastpath to prove which exchange path produced the
  // results they compare.  The 64-byte tail (error word + counters) is STICKY: launches zero the granules only
  if (threadIdx.x == 0) atomicAdd(err_word + (same_xcd ? 1 : 2), 1u);
  // |e| <= sum|v|: with both bounds <= 40 the softmax numerators exp(e - bound) cannot under/overflow, so the
  // member-local max pass (and one barrier) is skipped and every member uses the same constant shift
  const float VB1 = tab[(2 + F) * 64 * NQ + 66], VB2 = tab[(2 + F) * 64 * NQ + 67];
  const bool vsafe = VB1 <= 40.f && VB2 <= 40.f;
  // LAZY (r5, folded kernel): the normalisation leaves the step's dependency chain.  The forward variable is carried UN-normalised
  // (g = w u1 with its sum SG; the reference's recursion `((1-u) a + u shift(a) + 1e-7) / sum` - modules/forward_attention.py:108-110 -
  // is evaluated as w = ((1-u) g + u shift(g)) / SG_prev + 1e-7), the folded context product takes split(g) in A rows 4..6 so that
  // its result lands in lanes 16..31 of the SAME accumulators and is scaled by 1 / SG_prev behind the chain, the location
  // convolution scales its sum by 1 / S1_prev, and the second source's context is normalised by the wave that gathers its
  // partials.  Everything the next gate product needs is written by the gather callbacks of X2: one barrier behind the exchange
  // instead of scalars -> rows -> contexts -> barrier (0.9 us of 6.5 per step).  The normalised rows (saved for the backward pass:
  // a1, alpha, a2) are stored one step late by the waves the single-wave cell phase leaves idle.  Needs the constant softmax shift
  // with room for the bf16 split of g (bounds <= 30: u >= e^-60, the low part of w u stays a normal number); cumulative location
  // input keeps the in-chain form.
  const bool lazy = FOLD && VB1 <= 30.f && VB2 <= 30.f && p.cumulative == 0;
  float iS1p = 1.f, iSGp = 1.f, iS2p = 1.f;     // 1 / (S1, SG, S2) of the previous step (1: the carried rows are normalised)
  bool rows_pending = false;                    // the previous step's normalised rows are not stored yet
  // forced-alignment mode (see satt_hip.h).  Never in the folded (training) kernel - the launcher refuses the combination - and
  // there it must be a compile-time false: the conditional loads of the given alignments put an s_waitcnt vmcnt(0) at their join
  // in the tail of the energy rows, where every wave then waited for its factor-row stores and the next step's x-gate loads (r4)
  const bool forced = !FOLD && p.teach1 != nullptr && p.teach2 != nullptr;
  // location_sensitive: no alpha recursion - the forward-attention weight w is the constant 1, so alpha == softmax(e);
  // cumulative: the location-conv input accumulates the softmax alignments (satt_attn_rnn_params.att1_mode / cumulative)
  const bool unit_w = forced || p.att1_mode == 1;
  const bool cumul = p.cumulative != 0 && !forced;
  // transition agent (modules/forward_attention.py:111-116; generic instantiation only): the transition probability uc of
  // step t is predicted at the end of step t-1 from [ctx1 | processed query 1]; uc == 0.5 (a compile-time constant in the
  // specialised kernel) without it.  uw[3..7]: the per-wave partial dot products of the step, summed after its last barrier.
  const bool agent = !SPEC && p.agentW != nullptr && !unit_w;
  float* uw = smem + L.dead + 4;
  const float agent_b = agent ? p.agentb[0] : 0.f;
  float uc = 0.5f;
  if (agent && cp.t0 > 0) uc = p.ustate[(size_t)b * Td + cp.t0];
  float cst = 0.f, hst = 0.f;
  float* alp = alA;
  float* aln = alB;
  // LOCM B tiles of this wave (N tiles wave and wave + AW of the U1 / 16): lane l holds unit 16 nt + (l & 15), K slots
  // (l >> 4) * 8 + e: slots 0..4 = bf16(TS U[k]) (x fl_hi), 5..9 the same (x fl_lo), 10..14 = the bf16 residual (x fl_hi)
  constexpr int NT1 = SPEC ? SpecDimsOf<SPEC>::U1 / 16 : 1;
  i32x4_t lub[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  if constexpr (FOLD) {
    const int lane_ = (int)threadIdx.x & 63, wave_ = (int)threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nt = wave_ + j * AW, uu = min(nt, NT1 - 1) * 16 + (lane_ & 15);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ks = (lane_ >> 4) * 8 + e, k = ks % F;
        const float x = TS * p.locU[k * U1 + uu];
        const uint16_t hi = f2bf(x), lo = f2bf(x - bf2f(hi));
        const uint32_t v = (nt < NT1 && ks < 3 * F) ? (uint32_t)(ks < 2 * F ? hi : lo) : 0u;
        lub[j][e >> 1] |= (int)(v << ((e & 1) * 16));
      }
    }
  }
  if (cp.t0 > 0) {        // chunked launch: restart from the tensors saved by the previous chunk at step t0-1
    const int tid = threadIdx.x;
    const size_t bp = (size_t)b * Td + cp.t0 - 1;
    __syncthreads();
    for (int i = tid; i < CTF; i += ANT) xs_put(xs, XS, i, out[(size_t)(cp.t0 - 1) * OW + A + C0 + i]);
    for (int i = tid; i < A; i += ANT) xs_put(xs, XS, CTF + i, p.hstate[bp * A + i]);
    if (FOLD) for (int i = tid; i < Ti; i += ANT) xs_put(als, ALS, i, p.align1[bp * Ti + i]);
    for (int i = tid; i < Ti; i += ANT) { aprev[i] = cumul ? p.acum[bp * Ti + i] : p.a1[bp * Ti + i]; alA[i] = p.align1[bp * Ti + i]; }
    if (tid < AU) { cst = p.cstate[bp * A + c * AU + tid]; hst = p.hstate[bp * A + c * AU + tid]; }
  }
  __syncthreads();

  int bidx = 0;                          // chunk signalling (see satt_attn_cluster_params)
  while (bidx < cp.nbound && cp.bound[bidx] <= cp.t0) ++bidx;
  int next_bound = (cp.progress && bidx < cp.nbound) ? cp.bound[bidx] : -1;
  PLOG(5);
  PROF_DECL;
  float nxg[4];
  {
    const float* xr = xg + (size_t)cp.t0 * G + c * AU + min((int)threadIdx.x, AU - 1);
    nxg[0] = xr[0]; nxg[1] = xr[A]; nxg[2] = xr[2 * A]; nxg[3] = xr[3 * A];
  }
  for (int t = cp.t0, t_end = cp.t1; t < t_end; ++t) {
    PROF(0);
    // kernel arguments are re-read from the kernarg segment inside every step (see the backward kernel)
    typedef const __attribute__((address_space(4))) satt_attn_cluster_params KArgsF;
    KArgsF* kq = (KArgsF*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kq));
    const auto& cp = *kq;
    const auto& p = cp.f;
    // an opaque per-step zero keeps every thread-index expression INSIDE the step: nothing index-like is hoisted
    // out of the time loop, so the only long-lived registers are the weights and the recurrent state
    int oz = 0;
    asm volatile("" : "+v"(oz));
    const int tid = (int)threadIdx.x + oz, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(tid >= 0 && tid < ANT && wave >= 0 && wave < AW);   // ranges lost through the opaque zero
    const int d0 = lane * NQ;
    const bool actU = d0 < U1;
    const size_t bt = (size_t)b * Td + t;
    const uint32_t tag = (uint32_t)(t + 1);
    u64* wp = wsb + ((size_t)(t & 1) * p.B + b) * WL.per_parity;
    // input contributions of the own units: requested ONE STEP AHEAD (xg was written by a GEMM and comes from the MALL / HBM:
    // ~1 us, more than the gate product below that used to cover it); branch-free, clamped unit / step
    const float xi = nxg[0], xj = nxg[1], xf = nxg[2], xo = nxg[3];
    {
      const float* xr = xg + (size_t)min(t + 1, t_end - 1) * G + c * AU + min(tid, AU - 1);
      nxg[0] = xr[0]; nxg[1] = xr[A]; nxg[2] = xr[2 * A]; nxg[3] = xr[3 * A];
    }
    bool zkeep_c, zkeep_h;
    // (1) own gate columns: [ctx_{t-1} | h_{t-1}] x Wrec[:, own]  (A rows 0..2 = hi/mid/lo of x).  Straight-line:
    //     every register tile is multiplied (tiles beyond KT hold zeros), K tiles are consumed in pairs.
    {
      f32x4_t acc[2];
      acc[0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      const uint16_t* xrow = xs + min(lane & 15, 3) * XS + (lane >> 4) * 8;
      // EVERY A operand of the register tiles is requested before the first MFMA, and with them the operands of the first pair
      // of LDS-resident tiles (the second pair's go out behind the register chain): left to itself the compiler reuses two operand
      // registers and issues each block's ds_reads AFTER the previous block's MFMAs - nine exposed LDS latencies per step in
      // front of a chain that needs one (gate phase 0.76 us per step with 0.23 us of MFMA issue in it).
      bf16x8_t av[MKT];
#pragma unroll
      for (int kt = 0; kt < MKT; ++kt) av[kt] = *reinterpret_cast<const bf16x8_t*>(xrow + kt * 32);
      if constexpr (FOLD && SpecDimsOf<SPEC>::V2 == 32) {
        // LAZY: K tile 0 is the second source's context, carried UN-normalised: A rows 8..10 -> lanes 32..47, scaled by 1 / S2 below
        const int m16 = lane & 15, r8 = lazy ? ((m16 >= 8 && m16 < 11) ? m16 - 8 : 3) : min(m16, 3);
        av[0] = *reinterpret_cast<const bf16x8_t*>(xs + r8 * XS + (lane >> 4) * 8);
      }
      auto lds_pair = [&](int kl, bf16x8_t& a0, bf16x8_t& a1, i32x4_t (&b)[4]) {
        a0 = *reinterpret_cast<const bf16x8_t*>(xrow + (MKT + kl) * 32);
        a1 = *reinterpret_cast<const bf16x8_t*>(xrow + (MKT + kl + 1) * 32);
        const i32x4_t* w0 = Wl + ((wave * MNTW) * KTL + kl) * 64 + lane;
        b[0] = w0[0]; b[2] = w0[64];
        if (MNTW == 2) { b[1] = w0[KTL * 64]; b[3] = w0[KTL * 64 + 64]; }
      };
      bf16x8_t pa0 = av[0], pa1 = av[0], qa0 = av[0], qa1 = av[0];
      i32x4_t pb[4] = {}, qb[4] = {};
      if (KTL > 0) lds_pair(0, pa0, pa1, pb);
#pragma unroll
      for (int kt = 0; kt + 1 < MKT; kt += 2) {
        if (MNTW == 2) mfma22_a<false>(acc[0], acc[1], av[kt], av[kt + 1], wreg[0][kt], wreg[MNTW - 1][kt], wreg[0][kt + 1], wreg[MNTW - 1][kt + 1]);
        else mfma21_a<false>(acc[0], av[kt], av[kt + 1], wreg[0][kt], wreg[0][kt + 1]);
      }
      if (KTL > 2) lds_pair(2, qa0, qa1, qb);
      if (MKT & 1) {
        if (MNTW == 2) mfma12_a<false>(acc[0], acc[1], av[MKT - 1], wreg[0][MKT - 1], wreg[MNTW - 1][MKT - 1]);
        else mfma_bf16_areg<false>(acc[0], av[MKT - 1], wreg[0][MKT - 1]);
      }
      // The LDS-resident tiles continue the chain only where their count is a compile-time constant (SPEC): behind a run-time
      // branch the compiler copies the accumulators at the join (v_mov of a result the hardware has not finished - the static
      // check of tools/mfma_hazard_check.py caught exactly that), so there the register chain is covered first and every
      // conditional block carries its own cover.
      constexpr bool CHL = !SPEC;      // LAST flag of the conditional blocks
      if (!SPEC) mfma_cover(acc[0], acc[1]);
      if (KTL > 0) {
        if (MNTW == 2) mfma22_v<CHL>(acc[0], acc[1], pa0, pa1, pb[0], pb[1], pb[2], pb[3]);
        else mfma21_v<CHL>(acc[0], pa0, pa1, pb[0], pb[2]);
      }
      if (KTL > 2) {
        if (MNTW == 2) mfma22_v<CHL>(acc[0], acc[1], qa0, qa1, qb[0], qb[1], qb[2], qb[3]);
        else mfma21_v<CHL>(acc[0], qa0, qa1, qb[0], qb[2]);
      }
      for (int kl = 4; kl < KTL; kl += 2) {
        bf16x8_t a0, a1; i32x4_t b[4] = {};
        lds_pair(kl, a0, a1, b);
        if (MNTW == 2) mfma22_v(acc[0], acc[1], a0, a1, b[0], b[1], b[2], b[3]);
        else mfma21_v(acc[0], a0, a1, b[0], b[2]);
      }
      if (FOLD) {
        // ... continued by the folded first-source context: alpha_{t-1} (split, all memory rows) x own columns of VW1 (LDS tiles)
        // (LAZY: split(g) in A rows 4..6 - the product lands in lanes 16..31 of the accumulators, scaled by 1 / SG below)
        const int m16 = lane & 15;
        const int frow = lazy ? ((m16 >= 4 && m16 < 7) ? m16 - 4 : 3) : min(m16, 3);
        const uint16_t* arow = als + frow * ALS + (lane >> 4) * 8;
        bf16x8_t fa[FKT];
#pragma unroll
        for (int kt = 0; kt < FKT; ++kt) fa[kt] = *reinterpret_cast<const bf16x8_t*>(arow + kt * 32);
        static_assert(FKT == 5 && MNTW == 2 || !FOLD, "block structure below: 2 + 2 + 1 K tiles x 2 N tiles");
        mfma22_a<false>(acc[0], acc[1], fa[0], fa[1], wvr[0][0], wvr[MNTW - 1][0], wvr[0][1], wvr[MNTW - 1][1]);
        mfma22_a<false>(acc[0], acc[1], fa[2], fa[3], wvr[0][2], wvr[MNTW - 1][2], wvr[0][3], wvr[MNTW - 1][3]);
        mfma12_a<false>(acc[0], acc[1], fa[4], wvr[0][4], wvr[MNTW - 1][4]);
      }
      // the zoneout masks of the cell phase depend on (seed, step, unit) only: formed HERE, while the matrix pipe drains the chain
      // (the cell is one wave's instruction stream: twenty integer instructions less on it)
      {
        const uint32_t idxz = (uint32_t)bt * (uint32_t)A + (uint32_t)(c * AU + min(tid, AU - 1));
        const uint32_t zct = p.zc_thresh, zht = p.zh_thresh;
        zkeep_c = (satt_hash(seed, p.stream_c, idxz) >= zct) | (zct == 0);
        zkeep_h = (satt_hash(seed, p.stream_h, idxz) >= zht) | (zht == 0);
      }
      // SPEC: the blocks above are ONE accumulator chain (chained forms, mfma_rec.h): the result cover is paid once, here
      if (SPEC) mfma_cover(acc[0], acc[1]);
      float zs[MNTW];
#pragma unroll
      for (int j = 0; j < MNTW; ++j) zs[j] = acc[j][0] + acc[j][1] + acc[j][2];
      if constexpr (FOLD) {      // rows 4..6 (lanes 16..31): the folded context product of the un-normalised g (zero unless LAZY)
        const float sc = (lane & 32) ? iS2p : (lane & 16) ? iSGp : 1.f;
#pragma unroll
        for (int j = 0; j < MNTW; ++j) {
          const float sv_ = zs[j] * sc;
          // lanes 0..15 <- lanes 16..31 and 32..47: the gfx950 row / half swaps (one VALU instruction each, no trip through the LDS pipe)
          const unsigned su = __float_as_uint(sv_);
          const auto r16 = __builtin_amdgcn_permlane16_swap(su, su, false, false);
          const auto r32 = __builtin_amdgcn_permlane32_swap(su, su, false, false);
          zs[j] = (sv_ + __uint_as_float(r16[1])) + __uint_as_float(r32[1]);
        }
      }
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < MNTW; ++j) {
          const int n = (wave * MNTW + j) * 16 + lane;
          if (n < NL) z[n] = zs[j];
        }
      }
    }
    lds_barrier();
    PROF(1);
    auto conv_phase = [&](int ct) {
      float* flg = p.fl + bt * Ti * F;
      auto conv_elem = [&](int e) {
        const int i = e / F, k = e - i * F, tt = c + C * i;
        float s = bFs[k];
        if (lazy) {        // aprev holds the un-normalised numerators u1 of the previous step (or normalised rows and a unit scale)
          float sa = 0.f;
          for (int jj = 0; jj < KW; ++jj) sa += aprev[tt + jj - PL] * Fs[jj * F + k];
          s += iS1p * sa;
        } else
        for (int jj = 0; jj < KW; ++jj) s += aprev[tt + jj - PL] * Fs[jj * F + k];   // zero borders: no bounds test
        fl[tt * F + k] = s; pst_s(flg, (unsigned)(tt * F + k), s);
        if constexpr (FOLD) {      // LOCM A operand of own row i: K slots k | F + k | 2 F + k = hi | lo | hi (see LOC_ROWS)
          const uint16_t hi = f2bf(s), lo = f2bf(s - bf2f(hi));
          uint16_t* ar = las + i * 32 + k;
          ar[0] = hi; ar[F] = lo; ar[2 * F] = hi;
        }
      };
#ifdef SATT_EXP_NOCONV      // (timing experiment only: is the convolution beside the cell on the chain?  tools/build_variant.sh)
      if (ct >= 0) return;
#endif
      if constexpr (FOLD) {
        // (Ti <= 32 FKT = 160: one element per thread at most, the padding rows in two.  As LOOPS these stores made the wait-count
        // pass flush the vector-memory counter at the loop header - an s_waitcnt vmcnt(0) on the next step's x-gate loads and on
        // every pending output store, once per step, in the middle of the exchange window X1 (r4, found in the ISA listing))
        // The rows beyond the sequence length (never read back; zeroed to keep the saved tensor defined) are split like the live
        // rows - t' = c + C i of the member - and take the threads behind the convolution's: as ONE member's job (r4) that member
        // ran ~20 instructions more per step on every wave, was the last to publish in both exchanges of every step, and the
        // other three waited for it (trace of the members' publish times, any sample: 0.15-0.2 us behind)
        const int npad = (Ti - c + C - 1) / C;
        if (ct >= 0 && ct < nown * F) conv_elem(ct);
        else if (ct >= 0 && ct < npad * F) { const int i = ct / F, k = ct - i * F; pst_s(flg, (unsigned)((c + C * i) * F + k), 0.f); }
      } else {
        for (int e = ct; e < nown * F; e += ANT) conv_elem(e);
        // rows beyond the sequence length are never read back, but keep the saved tensor defined
        if (c == 2 % C) for (int e = ct + len * F; e < Ti * F; e += ANT) flg[e] = 0.f;
      }
    };
    // LAZY: the normalised rows of step ts (saved for the backward pass; alpha is also an output) from the carried numerators
    auto rows_out = [&](int r, size_t bts) {
      if (r < Ti) {          // (each member forms only the row it stores: this runs beside the cell phase and is not free)
        const bool ok = r < len;
        if (c == 0) pst_s(p.a1 + bts * Ti, (unsigned)r, ok ? aprev[r] * iS1p : 0.f);
        if (c == 1 % C) gst_s(p.align1 + bts * Ti, (unsigned)r, ok ? alp[r] * iSGp : 0.f);
        if (c == 2 % C) pst_s(p.align2 + bts * Ti, (unsigned)r, ok ? u2[r] * iS2p : 0.f);
      }
      // ctx2 of the step (an output; normalised by the sum the wave that gathers u2 formed)
      if (c == 3 % C && r < CTF) gst_s(out + (bts - (size_t)b * Td) * OW + A + C0, (unsigned)r, cg[8 + r] * iS2p);
    };
    // (2) LSTM cell for own units, publish h_state (consumed by the NEXT step), stage h' for the partial query
    float sv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (tid < AU) {
      const int j = c * AU + tid;
      const float gi = sigmoidf_(xi + z[tid]);
      const float gj = tanhf_(xj + z[AU + tid]);
      const float gf = sigmoidf_(xf + z[2 * AU + tid] + 1.0f);
      const float go = sigmoidf_(xo + z[3 * AU + tid]);
      const float cn = gf * cst + gi * gj;
      const float hn = go * tanhf_(cn);
      // (both hashes are computed unconditionally - behind `if (training)` / `thresh == 0 ||` each argument was its own kernarg load
      // with its own wait on the chain to the publication of h - and ahead of the phase, see (1))
      const int ztr = p.training;
      const bool keep_c = zkeep_c, keep_h = zkeep_h;
      if (ztr) {
        if (keep_c) cst = cn;
        if (keep_h) hst = hn;
      } else {
        cst = (1.f - p.zc) * cn + p.zc * cst;
        hst = (1.f - p.zh) * hn + p.zh * hst;
      }
      gput_s(wp + WL.x1 + c * AU, (unsigned)tid, tag, hst, same_xcd);
      xs_put(hs, HS, tid, hn);
      // r5: the eight result stores (with their address arithmetic a third of this single-wave, issue-bound instruction stream)
      // wait until the partial query is published: here they stood between the staging of h' and the barrier every wave waits at
      sv[0] = gi; sv[1] = gj; sv[2] = gf; sv[3] = go; sv[4] = cn; sv[5] = hn;
    } else if (FOLD) {
      // location features of the own rows (they need a_{t-1} only) on the waves the single-wave cell phase leaves idle; the
      // barrier below also hands the LOCM operands they stage to the product behind the publication of the partial query
      // (r6, tried and NOT kept: the convolution's eight elements beyond waves 1..3 moved off wave 4 - which shares SIMD 0 with the
      //  cell's wave - onto the upper half of wave 7 or into a second pass of wave 1: forward launch 2.07 -> 2.15 ms either way.  The
      //  convolution waves, not the cell, are what the barrier below waits for: one element per thread is already their floor)
      conv_phase(tid - AU);
      // LAZY: the previous step's rows on the waves neither the cell nor the convolution uses (Ti <= 160 <= ANT - AU - 256)
#ifndef SATT_EXP_NOROWSOUT
      if (rows_pending && tid - AU >= 256) rows_out(tid - AU - 256, bt - 1);
#endif
    }
    lds_barrier();
    // (3) partial processed query of the own units: h'_own x Wq[own rows, :]  -> published per column
    {
      f32x4_t acc0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      const uint16_t* hrow = hs + min(lane & 15, 3) * HS + (lane >> 4) * 8;
      const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(hrow), a1 = *reinterpret_cast<const bf16x8_t*>(hrow + 32);
      mfma22_a(acc0, acc1, a0, a1, wq[0][0], wq[1][0], wq[0][1], wq[1][1]);
      if (lane < 16) {
        const int n0 = wave * MNTQ * 16 + lane;
        if (n0 < UQ) gput_s(wp + WL.x1 + A + c * UQ, (unsigned)n0, tag, acc0[0] + acc0[1] + acc0[2], same_xcd);
        if (n0 + 16 < UQ) gput_s(wp + WL.x1 + A + c * UQ, (unsigned)(n0 + 16), tag, acc1[0] + acc1[1] + acc1[2], same_xcd);
      }
    }
    if (tid < AU) {          // the cell's results (see (2)): issued inside the exchange window X1
      const int j = c * AU + tid;
      float* gr = p.gates + bt * G;      // (scalar bases + 32-bit indices: see pst_s)
      pst_s(gr, (unsigned)j, sv[0]); pst_s(gr, (unsigned)(A + j), sv[1]); pst_s(gr, (unsigned)(2 * A + j), sv[2]); pst_s(gr, (unsigned)(3 * A + j), sv[3]);
      pst_s(p.cnew + bt * A, (unsigned)j, sv[4]);
      pst_s(p.cstate + bt * A, (unsigned)j, cst);
      pst_s(p.hstate + bt * A, (unsigned)j, hst);
      gst_s(out + (size_t)t * OW, (unsigned)j, sv[5]);
    }
    PROF(2); TRACE(t - cp.t0, 0);
    // (4) location features for own rows: the unfolded kernels compute them here (needs only a_{t-1}: hides the exchange
    //     latency); the folded kernel did so beside the cell phase and spends the window on the LOCM product instead
    if constexpr (!FOLD) conv_phase(tid);
    if constexpr (FOLD) {
      // LOCM: location term of the own rows (see LOC_ROWS).  A tile of M tile mt: lane l holds row 16 mt + (l & 15), K slots
      // (l >> 4) * 8 + e of the staging rows the conv threads filled (fl_hi[0..4], fl_lo[0..4], fl_hi[0..4], zeros).
      const int g4 = lane >> 4;                                       // (the conv results are behind the cell phase's barrier)
#pragma unroll
      for (int mt = 0; mt < LOC_ROWS / 16; ++mt) {
        const bf16x8_t a8 = *reinterpret_cast<const bf16x8_t*>(las + (mt * 16 + (lane & 15)) * 32 + g4 * 8);
        f32x4_t d0v, d1v;
        mfma12z_v(d0v, d1v, a8, lub[0], lub[1]);
        float* dst = lsm + (mt * 16 + g4 * 4) * LSTR + wave * 16 + (lane & 15);     // D[m = 4 (l >> 4) + r][n = l & 15]
        dst[0] = d0v[0]; dst[LSTR] = d0v[1]; dst[2 * LSTR] = d0v[2]; dst[3 * LSTR] = d0v[3];
        if (wave + AW < NT1) {
          float* dst1 = dst + AW * 16;
          dst1[0] = d1v[0]; dst1[LSTR] = d1v[1]; dst1[2 * LSTR] = d1v[2]; dst1[3 * LSTR] = d1v[3];
        }
      }
    }
    PROF(3);
    // per-lane attention parameters for (5): loaded before the gather so that their LDS latency overlaps it
    v2f vp01, vp23, Us01[F], Us23[F];
    float4 tb;
    {
      const float4 tv = *reinterpret_cast<const float4*>(tab + d0);
      tb = *reinterpret_cast<const float4*>(tab + 64 * NQ + d0);
      vp01 = (v2f){tv.x, tv.y}; vp23 = (v2f){tv.z, tv.w};
#pragma unroll
      for (int k = 0; k < F; ++k) {
        const float4 tu = *reinterpret_cast<const float4*>(tab + (2 + k) * 64 * NQ + d0);
        Us01[k] = (v2f){tu.x, tu.y}; Us23[k] = (v2f){tu.z, tu.w};
      }
    }
    const float v2p = tab[(2 + F) * 64 * NQ + lane];
    const float vs1 = tab[(2 + F) * 64 * NQ + 64], vs2 = tab[(2 + F) * 64 * NQ + 65];
    // X1: gather the partial processed queries of every member
    gather_span(wp + WL.x1 + A, C * UQ, tag, wave, AW, lane, [&](int i, float v) { dpart[i] = v; }, err_word, dead);
    lds_barrier();
    if constexpr (SPEC != 0) {       // (the processed query, saved for the backward pass: a quarter per member - see the padding rows)
      static_assert((SpecDimsOf<SPEC>::U1 + SpecDimsOf<SPEC>::U2) % SpecDimsOf<SPEC>::C == 0, "");
      if (tid < UQ / C) {
        const int col = c * (UQ / C) + tid;
        float s = 0.f;
        for (int k = 0; k < C; ++k) s += dpart[k * UQ + col];
        pst_s(p.pq + bt * UQ, (unsigned)col, s);
      }
    } else
    if (c == 1 % C && tid < UQ) {
      float s = 0.f;
      for (int k = 0; k < C; ++k) s += dpart[k * UQ + tid];
      pst_s(p.pq + bt * UQ, (unsigned)tid, s);
    }
    PROF(4); TRACE(t - cp.t0, 1);
    // (5) energies of own rows -> eo1 / eo2   (packed fp32 math; see the table setup for the scaling)
    {
      v2f pqs01, pqs23;
      {
        float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int dq = min(d0, UQ - NQ);          // clamped: lanes beyond U1 carry zero weights (their result is unused)
        for (int k = 0; k < C; ++k) {
          const float4 q4 = *reinterpret_cast<const float4*>(dpart + k * UQ + dq);   // UQ % 8 == 0, d0 % 4 == 0
          sacc.x += q4.x; sacc.y += q4.y; sacc.z += q4.z; sacc.w += q4.w;
        }
        pqs01 = (v2f){TS * (sacc.x + tb.x), TS * (sacc.y + tb.y)};
        pqs23 = (v2f){TS * (sacc.z + tb.z), TS * (sacc.w + tb.w)};
      }
      float pq2 = 0.f;
      for (int k = 0; k < C; ++k) pq2 += dpart[k * UQ + U1 + min(lane, U2 - 1)];   // lanes beyond U2: result unused
      pq2 *= TS;
      const v2f ts2 = (v2f){TS, TS}, one2 = (v2f){1.f, 1.f};
      uint16_t* const saf_rows = saf + ((size_t)bt * Ti + c + C * wave) * UQ;       // own row `wave` of this step (FOLD + saf)
      PROF(9);
      for (int i0 = wave; i0 < nown; i0 += RBF * AW) {
        float red[2 * RBF];
        // forward-attention weight of the row this lane finishes after the reduction (lanes < RBF): requested now so
        // that its LDS latency hides behind the row arithmetic
        float wrow;
        {
          const int tw = min(c + C * (i0 + min(lane, RBF - 1) * AW), Ti - 1);
          // (LAZY: alp holds g of the previous step; the scale 1 / SG rides on the two transition weights)
          wrow = ((1.f - uc) * iSGp) * alp[tw] + (tw > 0 ? uc * iSGp : 0.f) * alp[max(tw - 1, 0)] + 1e-7f;
        }
        // r5: a wave whose RBF rows all exist (and, folded kernel, with the factor rows to save) runs them without the per-row tests -
        // six scalar / copy instructions per row in the step's largest phase (issue bound).  Same arithmetic either way.
        auto rows_pass = [&](auto full_tag) {
        constexpr bool FULLW = decltype(full_tag)::value;
#pragma unroll
        for (int u = 0; u < RBF; ++u) {
          const int i = i0 + u * AW, tt = c + C * i;
          float acc = 0.f, acc2 = 0.f;
          if (FULLW || i < nown) {
            float kk[NQ];
            load_key4u<KLDS>(keys1 + (KLDS ? 0 : (size_t)tt * U1), K1s, KLDS ? i : 0, U1, d0, kk);
            const float k2 = load_key1u<KLDS>(keys2 + (KLDS ? 0 : (size_t)tt * U2), K2s, KLDS ? i : 0, U2, lane);
            v2f x01 = (v2f){kk[0], kk[1]} * ts2 + pqs01, x23 = (v2f){kk[2], kk[3]} * ts2 + pqs23;
            if constexpr (FOLD) {          // LOCM: the location term comes from the matrix cores (one 16-byte LDS read)
              const float4 l4 = *reinterpret_cast<const float4*>(lsm + i * LSTR + min(d0, U1 - NQ));
              x01 += (v2f){l4.x, l4.y}; x23 += (v2f){l4.z, l4.w};
            } else {
#pragma unroll
              for (int k = 0; k < F; ++k) {
                const float fk = fl[tt * F + k];
                const v2f f2 = (v2f){fk, fk};
                x01 = f2 * Us01[k] + x01; x23 = f2 * Us23[k] + x23;
              }
            }
            const v2f e01 = (v2f){exp2f_(x01.x), exp2f_(x01.y)} + one2, e23 = (v2f){exp2f_(x23.x), exp2f_(x23.y)} + one2;
            const v2f r01 = (v2f){__builtin_amdgcn_rcpf(e01.x), __builtin_amdgcn_rcpf(e01.y)};
            const v2f r23 = (v2f){__builtin_amdgcn_rcpf(e23.x), __builtin_amdgcn_rcpf(e23.y)};
            const v2f a2 = vp01 * r01 + vp23 * r23;
            acc = a2.x + a2.y;
            const float r2 = __builtin_amdgcn_rcpf(1.f + exp2f_(TS * k2 + pq2));
            acc2 = lane < U2 ? v2p * r2 : 0.f;
            if (FOLD && (FULLW || saf)) {     // s = r - 1/2 of this row for the backward pass (satt_attn_rnn_params.saf; tanh = -2 s,
              //                      r (1 - r) = 1/4 - s^2: both consumers get what they need).  Stored right here: holding the
              //                      values until after the exchange X2, or until the next step's recurrent product, measured
              //                      slower (2.66 ms per launch against 2.62; without the stores 2.57, without any of it 2.46)
              const v2f half2 = (v2f){0.5f, 0.5f};
              const v2f q01 = r01 - half2, q23 = r23 - half2;
              typedef __attribute__((ext_vector_type(2))) __fp16 h2;
              union { h2 h[2]; uint2 u; } pk;
              pk.h[0] = __builtin_amdgcn_cvt_pkrtz(q01.x, q01.y); pk.h[1] = __builtin_amdgcn_cvt_pkrtz(q23.x, q23.y);
              // (row pointers from ONE 64-bit base per step: own rows are C * AW * UQ halfs apart)
              // (r5: scalar row base + 32-bit lane offset, written as the instruction: the compiler forms a 64-bit vector address
              //  per store - three VALU instructions each in the phase that is issue bound; byte offsets 2 d0 / 2 (U1 + lane))
              uint16_t* row = saf_rows + (size_t)(u + (i0 - wave) / AW) * (size_t)(C * AW * UQ);
              const __fp16 h2v = (__fp16)(r2 - 0.5f);
              if (d0 < U1) asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"(2 * d0), "v"(pk.u), "s"(row) : "memory");
              if (lane < U2) asm volatile("global_store_short %0, %1, %2" :: "v"(2 * (U1 + lane)), "v"(h2v), "s"(row) : "memory");
            }
          }
          red[u] = acc; red[RBF + u] = acc2;
        }
        };
        if (FOLD && saf != nullptr && i0 + (RBF - 1) * AW < nown) rows_pass(std::true_type{}); else rows_pass(std::false_type{});
        // transposing reduction: lane u (< RBF) ends up with the two energies of row i0 + u*AW
        float red16[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) red16[q] = 0.f;
#pragma unroll
        for (int u = 0; u < RBF; ++u) { red16[u] = red[u]; red16[8 + u] = red[RBF + u]; }
        const float tot = wave_sum_transpose<16>(red16);          // lane l: total of slot l & 15
        const float tot2 = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + 8) & 63) << 2, __float_as_int(tot)));
        if (lane < RBF) {
          const int i = i0 + lane * AW;
          const float r1 = tot, r2 = tot2;
          if (i < nown) {
            const float e1v = vs1 + r1, e2v = vs2 + r2;
            if (vsafe) {       // numerators with the constant shift; see (6)
              const int tt = c + C * i;
              float uu1 = exp2f_(1.4426950408889634f * (e1v - VB1)), uu2 = exp2f_(1.4426950408889634f * (e2v - VB2));
              float wg = unit_w ? 1.f : wrow;
              if (forced) {      // forced-alignment mode: the given alignments take the place of the numerators
                uu1 = p.teach1[bt * Ti + tt]; uu2 = p.teach2[bt * Ti + tt];
              }
              const float g = wg * uu1;
              if (!FOLD) xs_put(gs, GS, i, g);
              xs_put(us, GS, i, uu2);
              gput_s(wp + WL.x2, (unsigned)tt, tag, uu1, same_xcd); gput_s(wp + WL.x2 + Ti, (unsigned)tt, tag, uu2, same_xcd);
              eo1[i] = uu1; eo2[i] = g; eo3[i] = uu2;
            } else {
              eo1[i] = e1v; eo2[i] = e2v;
            }
          }
        }
      }
    }
    PROF(10);
    lds_barrier();
    PROF(5);
    // (6) member-local softmax numerators: u = exp(e - m_member); for the forward attention also g = w * u with
    //     w = 0.5 alpha_{t-1}[t'] + 0.5 alpha_{t-1}[t'-1] + 1e-7.  Normalisation happens after the exchange:
    //     a1 = u1 f / S1, alpha = g f / SG, ctx1 = sum_members f * (sum_own g v) / SG   with f = exp(m_member - M).
    if (!vsafe) {
      if (wave == 0) {
        float m = -INFINITY;
        for (int i = lane; i < nown; i += 64) m = fmaxf(m, eo1[i]);
        m = forced ? 0.f : wave_max(m);
        float s = 0.f, sg = 0.f;
        for (int i = lane; i < nown; i += 64) {
          const int tt = c + C * i;
          const float uu = forced ? p.teach1[bt * Ti + tt] : exp2f_(1.4426950408889634f * (eo1[i] - m));
          const float w = unit_w ? 1.f : (1.f - uc) * alp[tt] + uc * (tt > 0 ? alp[tt - 1] : 0.f) + 1e-7f;
          const float g = w * uu;
          s += uu; sg += g;
          xs_put(gs, GS, i, g);
          gput(wp + WL.x2 + tt, tag, uu, same_xcd);
        }
        s = wave_sum(s); sg = wave_sum(sg);
        if (lane == 0) {
          u64* sc = wp + WL.x3 + c * (CTF + NSC) + CTF;
          gput(sc + 0, tag, m, same_xcd); gput(sc + 1, tag, s, same_xcd); gput(sc + 2, tag, sg, same_xcd);
        }
      } else if (wave == 1) {
        float m = -INFINITY;
        for (int i = lane; i < nown; i += 64) m = fmaxf(m, eo2[i]);
        m = forced ? 0.f : wave_max(m);
        float s = 0.f;
        for (int i = lane; i < nown; i += 64) {
          const int tt = c + C * i;
          const float uu = forced ? p.teach2[bt * Ti + tt] : exp2f_(1.4426950408889634f * (eo2[i] - m));
          s += uu;
          xs_put(us, GS, i, uu);
          gput(wp + WL.x2 + Ti + tt, tag, uu, same_xcd);
        }
        s = wave_sum(s);
        if (lane == 0) {
          u64* sc = wp + WL.x3 + c * (CTF + NSC) + CTF;
          gput(sc + 3, tag, m, same_xcd); gput(sc + 4, tag, s, same_xcd);
          for (int q = 5; q < NSC; ++q) gput(sc + q, tag, 0.f, same_xcd);
        }
      }
      lds_barrier();
    } else if (wave == AW - 1 && !lazy) {   // sums of the numerators written by (5); shift = the common bounds (LAZY: the gathering waves sum)
      float s1 = 0.f, sg = 0.f, s2 = 0.f;
      for (int i = lane; i < nown; i += 64) { s1 += eo1[i]; sg += eo2[i]; s2 += eo3[i]; }
      s1 = wave_sum(s1); sg = wave_sum(sg); s2 = wave_sum(s2);
      if (lane == 0) {
        u64* sc = wp + WL.x3 + c * (CTF + NSC) + CTF;
        gput(sc + 0, tag, VB1, same_xcd); gput(sc + 1, tag, s1, same_xcd); gput(sc + 2, tag, sg, same_xcd);
        gput(sc + 3, tag, VB2, same_xcd); gput(sc + 4, tag, s2, same_xcd);
        for (int q = 5; q < NSC; ++q) gput(sc + q, tag, 0.f, same_xcd);
      }
    }
    PROF(6);
    // (7) unnormalised partial contexts of the own rows by MFMA: [g | u2] (3-way split rows) x own value tiles
    if (FOLD) {
      // the second source's context only (NTV = V2 / 16 tiles): u2 rows x own value rows of source 2
      // (LAZY: on waves 5.. - waves 0..4 go straight to their gathers, wave AW - 1 forms the sums)
      const int wv0 = lazy ? 5 : 0;
      for (int nt = wave - wv0; nt >= 0 && nt < NTV; nt += AW) {
        const uint16_t* arow = us + min(lane & 15, 3) * GS + (lane >> 4) * 8;
        f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (KTO == 2) {      // r5: all four operands requested before the chain (LDS reads do not move across an asm block)
          const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(arow), a1 = *reinterpret_cast<const bf16x8_t*>(arow + 32);
          const i32x4_t b0 = Vt[nt * 64 + lane], b1 = Vt[(NTV + nt) * 64 + lane];
          mfma_chain2_z(acc, a0, a1, b0, b1);
        } else
        for (int kt = 0; kt < KTO; ++kt)
          mfma_bf16_vreg(acc, *reinterpret_cast<const bf16x8_t*>(arow + kt * 32), Vt[(kt * NTV + nt) * 64 + lane]);
        if (lane < 16) {
          const int col = nt * 16 + lane;
          if (col < CTF) gput(wp + WL.x3 + c * (CTF + NSC) + col, tag, acc[0] + acc[1] + acc[2], same_xcd);
        }
      }
    } else if (KLDS && KTO == 2 && NTV > AW && NTV <= 3 * AW) {
      // common case (<= 64 own rows, 9..24 context tiles): the wave's two (three) tiles in one (two) asm blocks
      const int lrow = min(lane & 15, 3) * GS + (lane >> 4) * 8;
      const int nt0 = wave, nt1 = wave + AW, nt2 = wave + 2 * AW;
      const uint16_t* ar0 = ((nt0 * 16 < V1) ? gs : us) + lrow;
      const uint16_t* ar1 = ((nt1 * 16 < V1) ? gs : us) + lrow;
      const bool two = nt1 < NTV;                    // wave-uniform
      const int nt1c = two ? nt1 : nt0;
      f32x4_t q0, q1;
      mfma_2chains_z(q0, q1, *reinterpret_cast<const bf16x8_t*>(ar0), *reinterpret_cast<const bf16x8_t*>(ar0 + 32),
                     *reinterpret_cast<const bf16x8_t*>(ar1), *reinterpret_cast<const bf16x8_t*>(ar1 + 32),
                     Vt[nt0 * 64 + lane], Vt[(NTV + nt0) * 64 + lane], Vt[nt1c * 64 + lane], Vt[(NTV + nt1c) * 64 + lane]);
      u64* x3 = wp + WL.x3 + c * (CT + NSC);
      if (lane < 16) {
        const int col0 = nt0 * 16 + lane, col1 = nt1 * 16 + lane;
        if (col0 < CT) gput(x3 + col0, tag, q0[0] + q0[1] + q0[2], same_xcd);
        if (two && col1 < CT) gput(x3 + col1, tag, q1[0] + q1[1] + q1[2], same_xcd);
      }
      if (nt2 < NTV) {
        const uint16_t* ar2 = ((nt2 * 16 < V1) ? gs : us) + lrow;
        f32x4_t q2;
        mfma_chain2_z(q2, *reinterpret_cast<const bf16x8_t*>(ar2), *reinterpret_cast<const bf16x8_t*>(ar2 + 32),
                      Vt[nt2 * 64 + lane], Vt[(NTV + nt2) * 64 + lane]);
        const int col2 = nt2 * 16 + lane;
        if (lane < 16 && col2 < CT) gput(x3 + col2, tag, q2[0] + q2[1] + q2[2], same_xcd);
      }
    } else
    for (int nt = wave; nt < NTV; nt += AW) {
      f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      const uint16_t* arow = ((nt * 16 + C0 < V1) ? gs : us) + min(lane & 15, 3) * GS + (lane >> 4) * 8;
      for (int kt = 0; kt < KTO; ++kt) {
        const bf16x8_t av = *reinterpret_cast<const bf16x8_t*>(arow + kt * 32);
        if (KLDS) {
          mfma_bf16_vreg(acc, av, Vt[(kt * NTV + nt) * 64 + lane]);
        } else {       // exact mode: fp32 values from L2, split 3-way on the fly
          float v[8];
          load_v8(kt * 32 + (lane >> 4) * 8, nt * 16 + (lane & 15), v);
          i32x4_t bh, bm, bl;
          split8(v, bh, bm, bl);
          mfma_bf16_vreg(acc, av, bh); mfma_bf16_vreg(acc, av, bm); mfma_bf16_vreg(acc, av, bl);
        }
      }
      if (lane < 16) {
        const int col = nt * 16 + lane;
        if (col < CTF) gput(wp + WL.x3 + c * (CTF + NSC) + col, tag, acc[0] + acc[1] + acc[2], same_xcd);
      }
    }
    TRACE(t - cp.t0, 2);
    // X2: one exchange for everything the normalisation needs: u1, u2 (rows), partial contexts + scalars, h_state
    if (FOLD && lazy) {
      // LAZY: the callbacks leave everything the next gate product reads - split(g) of all rows, [ctx2 | h] - in LDS
      if (wave == 0 || wave == AW - 1) {   // u1 -> location-conv input, g = w u1 -> alpha carry + A operand of the folded product, and
        //                                     their sums S1, SG: rows 0..127 on wave 0, the rest on the last wave
        constexpr int NP = 2;
        const int r0 = wave == 0 ? 0 : 128, cnt = min(len - r0, 128);
        float s1 = 0.f, sg = 0.f;
        if (cnt > 0) {
          const gu64* g[NP]; u64 x[NP]; float wr[NP];
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            const int i = r0 + min(lane + 64 * q, cnt - 1);
            g[q] = (const gu64*)(wp + WL.x2 + i); x[q] = 0;
            wr[q] = unit_w ? 1.f : ((1.f - uc) * iSGp) * alp[i] + (i > 0 ? uc * iSGp : 0.f) * alp[max(i - 1, 0)] + 1e-7f;
          }
          poll_or_die<NP>(g, tag, x, lane, err_word, dead);
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            const int i = r0 + lane + 64 * q;
            if (lane + 64 * q < cnt) {
              const float v = __uint_as_float((uint32_t)x[q]), gq = wr[q] * v;
              aprev[i] = v; aln[i] = gq; s1 += v; sg += gq;
              xs_put(als, ALS, i, gq);
            }
          }
        }
        s1 = wave_sum(s1); sg = wave_sum(sg);
        if (lane == 0) { cg[wave == 0 ? 0 : 2] = s1; cg[wave == 0 ? 1 : 3] = sg; }
      } else if (wave == 1) {    // u2 of all rows and their sum S2
        constexpr int NP = 3;    // Ti <= 32 FKT = 160 < 64 NP
        const gu64* g[NP]; u64 x[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) { g[q] = (const gu64*)(wp + WL.x2 + Ti + min(lane + 64 * q, len - 1)); x[q] = 0; }
        poll_or_die<NP>(g, tag, x, lane, err_word, dead);
        float s2 = 0.f;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const int i = lane + 64 * q;
          if (i < len) { const float v = __uint_as_float((uint32_t)x[q]); u2[i] = v; s2 += v; }
        }
        s2 = wave_sum(s2);
        if (lane == 0) cg[4] = s2;
      }
      else if (wave < 4) gather_span(wp + WL.x1, A, tag, wave - 2, 2, lane, [&](int i, float v) { xs_put(xs, XS, CTF + i, v); }, err_word, dead);
      else if (wave == 4 && CTF > 0) {      // ctx2 partials of the C members, column per lane: summed here, normalised behind the product
        constexpr int NC = SPEC ? SpecDimsOf<SPEC>::C : 1;
        const gu64* g[NC]; u64 x[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) { g[k] = (const gu64*)(wp + WL.x3 + k * (CTF + NSC) + min(lane, max(CTF - 1, 0))); x[k] = 0; }
        // (r5: a sleep of 3..12 x 64 clocks in front of this poll - the partials it waits for are published by waves 5 / 6 AFTER the
        //  barrier above - measured flat, 12 slower: not kept)
        poll_or_die<NC>(g, tag, x, lane, err_word, dead);
        float sm = 0.f;
#pragma unroll
        for (int k = 0; k < NC; ++k) sm += __uint_as_float((uint32_t)x[k]);
        if (lane < CTF) { xs_put(xs, XS, lane, sm); cg[8 + lane] = sm; }
      }
    } else
    if (wave == 0) gather_span(wp + WL.x2, len, tag, 0, 1, lane, [&](int i, float v) { u1[i] = v; }, err_word, dead);
    else if (wave == 1) gather_span(wp + WL.x2 + Ti, len, tag, 0, 1, lane, [&](int i, float v) { u2[i] = v; }, err_word, dead);
    else if (FOLD) {       // little is left of the context exchange: h over two waves, the compact [ctx2 | scalars] block by one
      if (wave < 4) gather_span(wp + WL.x1, A, tag, wave - 2, 2, lane, [&](int i, float v) { xs_put(xs, XS, CTF + i, v); }, err_word, dead);
      else if (wave == 4) gather_span(wp + WL.x3, C * (CTF + NSC), tag, 0, 1, lane, [&](int i, float v) { cg[i] = v; }, err_word, dead);
    }
    else if (wave == 2) gather_span(wp + WL.x1, A, tag, 0, 1, lane, [&](int i, float v) { xs_put(xs, XS, CTF + i, v); }, err_word, dead);
    else gather_span(wp + WL.x3, C * (CTF + NSC), tag, wave - 3, AW - 3, lane, [&](int i, float v) { cg[i] = v; }, err_word, dead);
    lds_barrier();
    PROF(7); TRACE(t - cp.t0, 3);
    // (8) normalisation (redundant, bitwise identical in every member).  The member scalars are read once into
    //     registers (one LDS latency); f = exp(m_member - M) per member, selected per row / summed per column.
    if (FOLD && lazy) {      // LAZY: nothing is left of it on the chain but the three reciprocals the next step applies
      {
        const float4 q = *reinterpret_cast<const float4*>(cg);
        iS1p = __builtin_amdgcn_rcpf(q.x + q.z); iSGp = __builtin_amdgcn_rcpf(q.y + q.w); iS2p = __builtin_amdgcn_rcpf(cg[4]);
      }
      rows_pending = true;
    } else {
      constexpr int MC = 8;                         // C <= 8
      constexpr float L2E = 1.4426950408889634f;
      float f1[MC], f2[MC];
      float M1 = -INFINITY, M2 = -INFINITY, S1 = 0.f, SG = 0.f, S2 = 0.f;
      if (vsafe) {       // every member used the same constant shift: all rescaling factors are exactly 1
#pragma unroll
        for (int k = 0; k < MC; ++k) {
          f1[k] = k < C ? 1.f : 0.f; f2[k] = f1[k];
          if (k < C) {
            const float4 sc = *reinterpret_cast<const float4*>(cg + k * (CTF + NSC) + CTF);   // m1 s1 sg m2
            S1 += sc.y; SG += sc.z; S2 += cg[k * (CTF + NSC) + CTF + 4];
          }
        }
      } else {
        float m1[MC], m2[MC], s1[MC], sg[MC], s2[MC];
#pragma unroll
        for (int k = 0; k < MC; ++k) {
          const float* sc = cg + (k < C ? k : 0) * (CTF + NSC) + CTF;
          m1[k] = k < C ? sc[0] : -INFINITY; s1[k] = sc[1]; sg[k] = sc[2]; m2[k] = k < C ? sc[3] : -INFINITY; s2[k] = sc[4];
        }
#pragma unroll
        for (int k = 0; k < MC; ++k) { M1 = fmaxf(M1, m1[k]); M2 = fmaxf(M2, m2[k]); }
#pragma unroll
        for (int k = 0; k < MC; ++k) {
          f1[k] = exp2f_(L2E * (m1[k] - M1)); f2[k] = exp2f_(L2E * (m2[k] - M2));     // 0 for k >= C
          S1 += f1[k] * s1[k]; SG += f1[k] * sg[k]; S2 += f2[k] * s2[k];
        }
      }
      const float iS1 = __builtin_amdgcn_rcpf(S1), iSG = __builtin_amdgcn_rcpf(SG), iS2 = __builtin_amdgcn_rcpf(S2);
      TRACE(t - cp.t0, 5);
      if (wave < 3) for (int tt = tid; tt < Ti; tt += 192) {
        // unconditional loads (rows >= len hold stale but finite numerators), masked afterwards
        const float uu = u1[tt], u2v = u2[tt], ap = alp[tt], am = alp[max(tt - 1, 0)];
        const int cm = tt % C;
        float g1 = f1[0], g2 = f2[0];
        if (!vsafe) {
#pragma unroll
          for (int k = 1; k < MC; ++k) { g1 = (cm == k) ? f1[k] : g1; g2 = (cm == k) ? f2[k] : g2; }
        }
        const float w = unit_w ? 1.f : (1.f - uc) * ap + (tt > 0 ? uc : 0.f) * am + 1e-7f;
        const bool ok = tt < len;
        const float a = ok ? uu * g1 * iS1 : 0.f;
        const float al = ok ? (w * uu) * g1 * iSG : 0.f;
        const float a2 = ok ? u2v * g2 * iS2 : 0.f;
        const float an = cumul ? aprev[tt] + a : a;      // next step's location-conv input
        aprev[tt] = an; aln[tt] = al;
        if (FOLD) xs_put(als, ALS, tt, al);           // A operand of the next step's folded context product
        if (cumul && c == 3 % C) p.acum[bt * Ti + tt] = an;
        if (c == 0) p.a1[bt * Ti + tt] = a;
        if (c == 1 % C) gst(p.align1 + bt * Ti + tt, al);
        if (c == 2 % C) p.align2[bt * Ti + tt] = a2;
      }
      TRACE(t - cp.t0, 6);
      float ua = 0.f;                       // transition agent: this thread's share of [ctx1 | pq1] . agentW
      if (wave >= 3) for (int i = tid - 192; i < CTF; i += ANT - 192) {
        const bool first = i + C0 < V1;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < MC; ++k)
          if (k < C) s += (first ? f1[k] : f2[k]) * cg[k * (CTF + NSC) + i];
        s *= first ? iSG : iS2;
        xs_put(xs, XS, i, s);
        if (c == 3 % C) gst(out + (size_t)t * OW + A + C0 + i, s);
        if (agent && first) ua += s * p.agentW[i];
      }
      if (agent && wave >= 3) {
        for (int d = tid - 192; d < U1; d += ANT - 192) {       // processed query 1 = the sum of the members' partials
          float q = 0.f;
          for (int k = 0; k < C; ++k) q += dpart[k * UQ + d];
          ua += q * p.agentW[V1 + d];
        }
        ua = wave_sum(ua);
        if (lane == 0) uw[wave] = ua;
      }
    }
    TRACE(t - cp.t0, 7);
    { float* tmp = alp; alp = aln; aln = tmp; }
    if (!(FOLD && lazy)) lds_barrier();
    if (agent) {         // u of the next step (redundant and identical in every thread); saved for the backward pass
      uc = sigmoidf_(((uw[3] + uw[4]) + (uw[5] + uw[6])) + (uw[7] + agent_b));
      if (c == 0 && threadIdx.x == 0 && t + 1 < Td) p.ustate[(size_t)b * Td + t + 1] = uc;
    }
    PROF(8); TRACE(t - cp.t0, 4);
    if (next_bound == t + 1) {           // end of a pipeline chunk: make the step's outputs visible, then count
      if (rows_pending) { rows_out(tid, bt); rows_pending = false; }      // LAZY: this step's rows belong to the chunk
      chunk_release();
      __syncthreads();
      if (threadIdx.x == 0) atomicAdd(cp.progress + bidx, 1u);   // one word per chunk: samples run at different speeds
      ++bidx;
      next_bound = bidx < cp.nbound ? cp.bound[bidx] : -1;
    }
  }
  if (rows_pending) {        // LAZY: the last step's rows (the state every thread needs is live behind the loop)
    const int len_ = len, Ti_ = Ti, r = (int)threadIdx.x;
    if (r < Ti_) {
      const size_t bts = (size_t)b * Td + cp.t1 - 1;
      const bool ok = r < len_;
      const float a = ok ? aprev[r] * iS1p : 0.f, al = ok ? alp[r] * iSGp : 0.f, a2 = ok ? u2[r] * iS2p : 0.f;
      if (c == 0) p.a1[bts * Ti_ + r] = a;
      if (c == 1 % C) gst(p.align1 + bts * Ti_ + r, al);
      if (c == 2 % C) p.align2[bts * Ti_ + r] = a2;
    }
    if (c == 3 % C && r < CTF) gst(out + (size_t)(cp.t1 - 1) * OW + A + C0 + r, cg[8 + r] * iS2p);
  }
  PROF_STORE(0);
}

constexpr int MNTB = 26;    // N tiles of the backward slice held in accumulation registers (the rest lives in LDS)
constexpr int RBB = 3;      // memory rows per wave iteration in the backward energy phase
// N-split form of the backward recurrent product (phase (h)): every wave owns 4 of the N tiles (16 outputs each) for ALL
// 8 K tiles and accumulates over K inside the MFMA accumulators, so the result leaves the registers straight into the
// exchange (no per-K-tile partials in LDS, no cross-wave reduction, one barrier less).  Needs 4 * AU == 256 (8 K tiles)
// and 32..34 N tiles; tiles 32.. are extra tiles of the last waves.  Slots per wave: kt * 4 + j (j-th own N tile), then
// 8 slots of the extra tile; the first MNTB slots live in accumulation registers, the rest in LDS.
constexpr int NS_SLOTS = 40;
__host__ __device__ constexpr bool nsplit_of(int K, int A, int C) {
  const int NTK = (K + 15) / 16;
  return C > 0 && 4 * (A / C) == 256 && NTK >= 4 * AW && NTK <= 4 * AW + 2;
}
constexpr int RBV = 5;      // memory rows per wave iteration in the backward d-alpha phase (<= 8)

__host__ __device__ constexpr int ntl_of(int NTK) { const int r = NTK > MNTB ? NTK - MNTB : 0; return (r + 3) & ~3; }
struct SmemCB {
  int dzs, dps, cgx, hpart, dqp, dpq, pqv, dctx, alprev, a, al, a2, dal, da2, de1, dac, dalc, draw, scal, fl, dfl, Fs, dpart,
      partial, tab, dead, wl, kofs, dcs, vs1, vs2, ext, ub, nl, total;
};
// VMF (r4): phase (b) of the backward step - raw d alpha of the own memory rows = value rows x d ctx - on the matrix cores.
// The own value rows of both sources stay in LDS as bf16 MFMA B tiles for the whole launch (VMF_ROWS rows, zero beyond the own
// count; row strides padded by 8 elements against bank conflicts), d ctx is split 3-way into A rows at the top of the step.
constexpr int VMF_ROWS = 48;      // 3 N tiles of 16 own rows: Ti <= 4 * 40
__host__ __device__ constexpr int vmf_stride(int V) { return V + 8; }
__host__ __device__ constexpr SmemCB carve_cb(int A, int CT, int UQ, int Ti, int F, int KW, int C, int nown, bool klds, bool vmf = false) {
  constexpr auto u = [](int x) constexpr { return (x + 3) & ~3; };
  const int KR = CT + A, NL = 4 * (A / C), NTK = (KR + 15) / 16, NTL = ntl_of(NTK), KRP = (MNTB + NTL) * 16;
  SmemCB s{}; int o = 0;
  s.dzs = o; o += 4 * a_stride(kt_of(NL)) / 2;   // bf16 [4][DZS] split own dz
  s.dps = o; o += 4 * a_stride(kt_of(UQ)) / 2;   // bf16 [4][DPS] split d pq
  s.cgx = o; o += u(C * KR);                     // [C][KR] partial d[ctx|h] of every member
  // [AW][KRP] per-K-tile partials of the own d[ctx|h]: K-split form only (the N-split form accumulates over K inside the MFMA
  // accumulators and publishes from them)
  s.hpart = o; if (!nsplit_of(KR, A, C)) o += AW * KRP;
  s.dqp = o; o += AW * 64;                       // [AW][64] per-K-tile partials of the own d query
  s.dpq = o; o += u(UQ); s.pqv = o; o += u(UQ); s.dctx = o; o += u(CT);
  const int T4 = u(Ti);
  s.alprev = o; o += T4; s.a = o; o += T4; s.al = o; o += T4; s.a2 = o; o += T4; s.dal = o; o += T4; s.da2 = o; o += T4;
  s.de1 = o; o += T4; s.dac = o; o += 3 * T4; s.dalc = o; o += T4;   // dac: 3 partial sums over filter-tap groups
  s.draw = o; o += 2 * T4; s.scal = o; o += 4 * AW + 8;                   // raw d alpha / d a2 of the own rows; per-wave partial sums
  s.ext = o; o += 2 * T4;                                              // external d alignment rows of the step (both sources)
  s.fl = o; if (!vmf) o += u(Ti * F);                          // (the saved-factor kernel never reads the location features)
  s.dfl = o; o += u((Ti + KW) * F); s.Fs = o; o += u(KW * F);   // dfl: zero rows around [0, Ti)
  s.dpart = o; o += u(C * UQ);
  s.partial = o; o += AW * u(UQ);
  s.tab = o; o += (2 + F) * 64 * NQ + 64;
  s.dead = o; o += 12;                          // [0]: timeout flag; [4..11]: transition-agent scalars
  s.wl = o; o += AW * NTL * 64 * 4;              // [AW][NTL][64 lanes][16 B]
  s.kofs = o; if (klds && !vmf) o += u((nown * UQ + 1) / 2);        // (the saved-factor kernel never reads the keys)
  s.dcs = o; s.vs1 = o; s.vs2 = o; s.ub = o; s.nl = o;
  if (vmf) {
    const int V2 = CT - A;                                           // (V1 == A in every specialisation that takes this path)
    o += 4 * a_stride(kt_of(CT)) / 2;                                // bf16 [4][DCS] split d ctx
    s.vs1 = o; o += u(VMF_ROWS * vmf_stride(A) / 2);
    s.vs2 = o; o += u(VMF_ROWS * vmf_stride(V2 > 0 ? V2 : 8) / 2);
    const int U1 = UQ - (V2 > 0 ? 32 : 0);                           // (the two specialisations: 224 + 32 and 256 + 0)
    s.ub = o; o += kt_of(U1) * 64 * 8 / 2;                           // fp16 B tiles of the location-feature map (see NLOC)
    s.nl = o; o += VMF_ROWS * 16;                                    // [own row][16]: N rows of the current step (hi | lo column pairs)
  }
  s.total = o;
  return s;
}

// Backward of the loop.  The recurrent products are K-SPLIT so that they use exactly the member's own slice:
//   d query_own = d pq x Wq^T[:, own units]                     (4 B tiles per wave, K tile = wave)
//   cell backward for the OWN units only -> dz_own (4 x AU values)
//   partial d[ctx|h] = dz_own x Wrec[:, own gate columns]^T     (K tile = wave, all N tiles: MNTB in registers)
//   Xh: all-reduce of the C partial d[ctx|h] vectors (summed in a fixed order -> identical in every member)
// SAF: the derivative factors r (1 - r) of the energy nonlinearity come from the forward pass (satt_attn_rnn_params.saf, fp16
// rows of U1 + U2 values per memory row and step) instead of being recomputed from keys + query + location features: the energy
// backward rows (d) - the largest phase of the step, VALU bound - lose their exp2 / rcp and the location term, phase (a) loses
// the location-feature and query rows.  The rows of a step are pulled into L2 one step ahead (one dummy load per wave) and read
// into registers one phase before their use.
template <int F, bool KLDS, int SPEC, bool NSPLIT, bool SAF = false>
__global__ __launch_bounds__(ANT) void attn_cluster_bwd_k(const satt_attn_cluster_bwd_params cb) {
  static_assert(!SAF || (SPEC != 0 && KLDS), "saved factors: specialised bf16 kernel only");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const satt_attn_rnn_bwd_params& pb = cb.b;
  const satt_attn_rnn_params& p = pb.f;
  const int C = SPEC ? SpecDimsOf<SPEC>::C : cb.C;
  const int A = SPEC ? SpecDimsOf<SPEC>::A : p.A, G = 4 * A, V1 = SPEC ? SpecDimsOf<SPEC>::V1 : p.V1, V2 = SPEC ? SpecDimsOf<SPEC>::V2 : p.V2;
  const int CT = V1 + V2, U1 = SPEC ? SpecDimsOf<SPEC>::U1 : p.U1, U2 = SPEC ? SpecDimsOf<SPEC>::U2 : p.U2, UQ = U1 + U2;
  const int Ti = p.Ti, Td = p.Td, KW = SPEC ? SpecDimsOf<SPEC>::KW : p.kernel, PL = (KW - 1) / 2;
  const int KR = CT + A, NWP = nwp_of(KR, C), AU = A / C, NL = 4 * AU;
  const int NTK = (KR + 15) / 16, NTL = ntl_of(NTK), KRP = (MNTB + NTL) * 16;   // every tile is multiplied unconditionally
  const int KTN = kt_of(NL), DZS = a_stride(KTN), KTU = kt_of(UQ), DPS = a_stride(KTU), NTA = (AU + 15) / 16;
  const int b = blockIdx.x, c = blockIdx.y;
  // SAF (r5): fixed LDS layout of Ti = 32 FKT, every offset an immediate (see the forward kernel)
  typedef SpecDimsOf<SPEC> DL;
  constexpr int TIL = 32 * FKT, NOWNL = (TIL + DL::C - 1) / DL::C;
  constexpr SmemCB LFIX = carve_cb(DL::A, DL::V1 + DL::V2, DL::U1 + DL::U2, TIL, F, DL::KW, DL::C, NOWNL, KLDS, SAF);
  const int nown_max = SAF ? NOWNL : (Ti + C - 1) / C;
  const SmemCB L = SAF ? LFIX : carve_cb(A, CT, UQ, Ti, F, KW, C, nown_max, KLDS, SAF);
  constexpr bool VMF = SAF;                    // value-row products of phase (b) on the matrix cores (specialised bf16 kernel)
  const int DCS = a_stride(kt_of(CT)), VS1 = vmf_stride(V1), VS2 = vmf_stride(V2 > 0 ? V2 : 8);
  uint16_t* dcs = reinterpret_cast<uint16_t*>(smem + L.dcs);
  uint16_t* vs1 = reinterpret_cast<uint16_t*>(smem + L.vs1);
  uint16_t* vs2 = reinterpret_cast<uint16_t*>(smem + L.vs2);
  // NLOC (r4): d fl[t', k] = sum_u g[t', u] U[k, u] with g = d e[t'] (4 v[u]) f[t', u] factorises as d e[t'] * N[t', k], and
  // N[t', k] = sum_u f[t', u] (4 v[u] U[k, u]) depends on the FORWARD pass only.  The rows N of the step processed next are formed
  // one step ahead, off the dependency chain (three otherwise idle waves during the single-wave cell phase): 16x16x32 fp16 MFMAs
  // with A = f = 1/4 - s^2 straight from the saved fp16 rows (loaded in operand layout, 4 packed FMAs per tile) and B = the
  // constant 4 v U split into fp16 hi + lo columns (resident in LDS).  The energy-backward rows (d) lose their five filter dot
  // products and both 16-value transposing wave reductions; phase (c) multiplies d e into the N row and publishes.
  typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
  constexpr int KT1 = SPEC ? (SpecDimsOf<SPEC>::U1 + 31) / 32 : 1;
  h8_t* ub = reinterpret_cast<h8_t*>(smem + L.ub);        // [KT1][64 lanes]: B operand tiles
  float* nl = smem + L.nl;                                 // [VMF_ROWS][16]: col 2k (+ 2k+1) = hi (+ lo) part of N[row][k]
  uint16_t* dzs = reinterpret_cast<uint16_t*>(smem + L.dzs);
  uint16_t* dps = reinterpret_cast<uint16_t*>(smem + L.dps);
  float* cgx = smem + L.cgx;        // [C][KR] gathered partial d[ctx|h]; their sum is the carried gradient
  float* hpart = smem + L.hpart;
  float* dqp = smem + L.dqp;
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
  float* dac = smem + L.dac;          // [3][T4] partial d a_{t-1} (tap groups), summed by the reader
  const int T4 = SAF ? ((TIL + 3) & ~3) : ((Ti + 3) & ~3);
  float* dalc = smem + L.dalc;
  float* draw = smem + L.draw;        // [2][T4] raw d alpha | d a2 of the own rows (own-row index)
  float* scal = smem + L.scal;        // [AW][4] per-wave partials of s1, s2, s3, S
  float* exts = smem + L.ext;         // [2][T4] external gradients wrt the two alignment rows of this step (zeros without)
  float* fl = smem + L.fl;
  float* dfl = smem + L.dfl + (KW - 1 - PL) * F;   // rows [-(KW-1-PL), Ti + PL]: the conv backward needs no bounds test
  float* Fs = smem + L.Fs;
  float* dpart = smem + L.dpart;    // [C][UQ] gathered d pq partials
  float* partial = smem + L.partial;
  float* tab = smem + L.tab;
  int* dead = reinterpret_cast<int*>(smem + L.dead);
  i32x4_t* Wl = reinterpret_cast<i32x4_t*>(smem + L.wl);
  uint16_t* K1s = reinterpret_cast<uint16_t*>(smem + L.kofs);
  uint16_t* K2s = K1s + nown_max * U1;
  const int UQ4 = (UQ + 3) & ~3;

  const int len = (int)p.lengths[b];
  const uint32_t seed = p.seed ? *p.seed : 0u;
  const float* keys1 = p.keys1 + (size_t)b * Ti * U1;
  const float* values1 = p.values1 + (size_t)b * Ti * V1;
  const float* keys2 = p.keys2 + (size_t)b * Ti * U2;
  const float* values2 = p.values2 + (size_t)b * Ti * V2;
  const int OW = A + CT;
  const float* dout = pb.dout + (size_t)b * Td * OW;
  const float* fout = p.out + (size_t)b * Td * OW;          // forward outputs [h | ctx1 | ctx2] per step
  constexpr WsLayout WLFIX = ws_layout(DL::A, TIL, DL::C, DL::U1 + DL::U2, F, DL::V1 + DL::V2 + DL::A);
  const WsLayout WL = SAF ? WLFIX : ws_layout(A, Ti, C, UQ, F, KR);
  u64* wsb = reinterpret_cast<u64*>(cb.ws);
  unsigned int* err_word = reinterpret_cast<unsigned int*>(wsb + ws_tail_words(p.B, A, Ti, C, UQ, F, CT, SPEC != 0));
  const int nown = len > c ? (len - c + C - 1) / C : 0;

  // register-resident backward slice (accumulation registers): B operand of tile (kt = wave, nt): lane l holds own gate
  // columns wave*32 + (l>>4)*8 .. +8 (local order g*AU + u) of input row nt*16 + (l&15).
  // Packed by satt_attn_cluster_pack as [C][AW][NTK][64][8] bf16; tiles nt >= MNTB go to LDS.
  i32x4_t wregT[MNTB];
  i32x4_t wqT[4];                // Wq^T[pq rows wave*32.., own units nt*16..]
  {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NSRC = NSPLIT ? NS_SLOTS : NTK;    // tuples per (member, wave) in the packed buffer
    const i32x4_t* wsrc = reinterpret_cast<const i32x4_t*>(cb.WrecTP) + (size_t)(c * AW + wave) * NSRC * 64 + lane;
    constexpr int WB = 8;                        // batched loads before the pins (see the forward kernel)
#pragma unroll
    for (int q0 = 0; q0 < MNTB; q0 += WB) {
      i32x4_t tmp[WB];
#pragma unroll
      for (int q = 0; q < WB; ++q) tmp[q] = (q0 + q < MNTB && q0 + q < NSRC) ? wsrc[(size_t)(q0 + q) * 64] : (i32x4_t){0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < WB; ++q)
        if (q0 + q < MNTB) { asm volatile("" : "+a"(tmp[q])); wregT[q0 + q] = tmp[q]; }
    }
    if (NSPLIT) {      // LDS: 6 slots (26..31) of every wave, then the 8 slots of each extra tile (waves >= AW - NX)
      const int NX = NTK - 4 * AW;
      i32x4_t tmp[WB];
#pragma unroll
      for (int q = 0; q < 6; ++q) tmp[q] = wsrc[(size_t)(MNTB + q) * 64];
#pragma unroll
      for (int q = 0; q < 6; ++q) Wl[(wave * 6 + q) * 64 + lane] = tmp[q];
      if (wave >= AW - NX) {
#pragma unroll
        for (int q = 0; q < 8; ++q) tmp[q] = wsrc[(size_t)(32 + q) * 64];
#pragma unroll
        for (int q = 0; q < 8; ++q) Wl[(6 * AW + (wave - (AW - NX)) * 8 + q) * 64 + lane] = tmp[q];
      }
    } else
    for (int q0 = 0; q0 < NTL; q0 += WB) {
      i32x4_t tmp[WB];
#pragma unroll
      for (int q = 0; q < WB; ++q)
        tmp[q] = (q0 + q < NTL && (MNTB + q0 + q) < NTK) ? wsrc[(size_t)(MNTB + q0 + q) * 64] : (i32x4_t){0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < WB; ++q)
        if (q0 + q < NTL) Wl[(wave * NTL + q0 + q) * 64 + lane] = tmp[q];
    }
    {   // transposed query-layer slice: branch-free loads, all in flight before the first use (see the forward kernel)
      uint16_t raw[4][8];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          raw[nt][i] = pb.WqT[(size_t)min(wave * 32 + (lane >> 4) * 8 + i, UQ - 1) * A + c * AU + min(nt * 16 + (lane & 15), AU - 1)];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int n = nt * 16 + (lane & 15);
        i32x4_t w = (i32x4_t){0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = wave * 32 + (lane >> 4) * 8 + i;
          const uint32_t v = (k < UQ && n < AU) ? (uint32_t)raw[nt][i] : 0u;
          w[i >> 1] |= (int)(v << ((i & 1) * 16));
        }
        asm volatile("" : "+a"(w));
        wqT[nt] = w;
      }
    }
    // tanh(x) = 1 - 2r, 1 - tanh^2 = 4 r (1 - r) with r = 1 / (1 + exp2(TS x)): v is stored as 4 v and U as TS U; the
    // d location-feature sums are taken against TS U and rescaled by 1 / TS once per row
    for (int i = tid; i < 64 * NQ; i += ANT) {
      tab[i] = i < U1 ? 4.f * p.v1[i] : 0.f;
      tab[64 * NQ + i] = i < U1 ? p.b1[i] : 0.f;
      for (int k = 0; k < F; ++k) tab[(2 + k) * 64 * NQ + i] = i < U1 ? TS * p.locU[k * U1 + i] : 0.f;
    }
    if (tid < 64) tab[(2 + F) * 64 * NQ + tid] = tid < U2 ? 4.f * p.v2[tid] : 0.f;
    for (int i = tid; i < 4 * DZS; i += ANT) dzs[i] = 0;
    for (int i = tid; i < 4 * DPS; i += ANT) dps[i] = 0;
    for (int i = tid; i < C * KR; i += ANT) cgx[i] = 0.f;
    for (int i = tid; i < AW * 64; i += ANT) dqp[i] = 0.f;
    if (!NSPLIT) for (int i = tid; i < AW * KRP; i += ANT) hpart[i] = 0.f;
    for (int i = tid; i < Ti; i += ANT) { dac[i] = 0.f; dac[T4 + i] = 0.f; dac[2 * T4 + i] = 0.f; dalc[i] = 0.f; dal[i] = 0.f; da2[i] = 0.f; }
    for (int i = tid; i < (Ti + KW) * F; i += ANT) dfl[i - (KW - 1 - PL) * F] = 0.f;
    for (int i = tid; i < KW * F; i += ANT) Fs[i] = p.locF[i];
    if (tid == 0) *dead = 0;
    if constexpr (VMF) {         // own value rows as bf16 B tiles (the engine stores the memory at bf16 precision in this mode:
      //                            the conversion is exact); rows beyond the own count are zero
      static_assert(SpecDimsOf<SPEC>::V1 == SpecDimsOf<SPEC>::A, "the image sizes of carve_cb assume V1 == A");
      for (int i = tid; i < 4 * DCS; i += ANT) dcs[i] = 0;
      for (int e = tid; e < VMF_ROWS * (V1 / 4); e += ANT) {
        const int i = e / (V1 / 4), d = (e - i * (V1 / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nown) v = *reinterpret_cast<const float4*>(values1 + (size_t)(c + C * i) * V1 + d);
        uint2 w;
        w.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16); w.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
        *reinterpret_cast<uint2*>(vs1 + i * VS1 + d) = w;
      }
      for (int e = tid; e < VMF_ROWS * (VS2 - 8); e += ANT) {
        const int i = e / (VS2 - 8), d = e - i * (VS2 - 8);
        vs2[i * VS2 + d] = (V2 > 0 && i < nown && d < V2) ? f2bf(values2[(size_t)(c + C * i) * V2 + d]) : (uint16_t)0;
      }
      // B tiles of NLOC: lane l of tile kt holds column n = l & 15 for the units kt*32 + (l >> 4)*8 + e; column 2k / 2k+1 = the fp16
      // hi / lo part of 4 v[u] U[k, u] (columns 10..15 and units beyond U1: zero)
      _Float16* ubh = reinterpret_cast<_Float16*>(ub);
      for (int e = tid; e < KT1 * 512; e += ANT) {
        const int kt = e >> 9, l = (e >> 3) & 63, j = e & 7, n = l & 15, u = kt * 32 + (l >> 4) * 8 + j, k = n >> 1;
        const float x = (k < F && u < U1) ? 4.f * p.v1[u] * p.locU[k * U1 + u] : 0.f;
        const _Float16 hi = (_Float16)x;
        ubh[e] = (n & 1) ? (_Float16)(x - (float)hi) : hi;
      }
      for (int e = tid; e < VMF_ROWS * 16; e += ANT) nl[e] = 0.f;
    }
    if (KLDS && !SAF) {          // (SAF: the keys are only needed for the recomputation it replaces)
#pragma unroll 4
      for (int e = tid; e < nown * U1; e += ANT) { const int i = e / U1, d = e - i * U1; K1s[e] = f2bf(keys1[(size_t)(c + C * i) * U1 + d]); }
      for (int e = tid; e < nown * U2; e += ANT) { const int i = e / U2, d = e - i * U2; K2s[e] = f2bf(keys2[(size_t)(c + C * i) * U2 + d]); }
    }
  }
  __syncthreads();
  // start-up handshake: are all members of this cluster on one XCD?  (granules with a tag no step can produce)
  {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* flagp = dead + 1;
    if (tid == 0) gput(wsb + (size_t)b * WL.per_parity + WL.xi + c, XCC_TAG, __int_as_float(xcc_id()), false);
    if (wave == 0) {
      int mism = 0;
      const int mine = xcc_id();
      gather_span(wsb + (size_t)b * WL.per_parity + WL.xi, C, XCC_TAG, 0, 1, lane,
                  [&](int i, float v) { if (__float_as_int(v) != mine) mism = 1; }, err_word, dead);
      mism = __any(mism) || *dead;
      if (lane == 0) *flagp = mism ? 0 : 1;
    }
    __syncthreads();
  }
  const bool same_xcd = dead[1] != 0;
  // words 1 / 2 behind the error word count the workgroup-launches that publish with plain (same-XCD) stores / with
  // write-through stores: tests read them through satt_attn_cluster_fastpath to prove which exchange path produced the
  // results they compare.  The 64-byte tail (error word + counters) is STICKY: launches zero the granules only
  if (threadIdx.x == 0) atomicAdd(err_word + (same_xcd ? 1 : 2), 1u);
  const bool unit_w = p.att1_mode == 1;      // location_sensitive: w == 1, nothing flows back into alpha_{t-1}
  const bool cumul = p.cumulative != 0;      // the conv input of step t feeds every later step: its gradient accumulates
  // GSPLIT (r4, specialised kernels: AU == 64): the cell backward (g) runs on FOUR waves - wave w < 4 keeps the carried gradients of
  // all own units, computes the shared part (d c, d o) redundantly and then forms, stores and stages ONE gate's d z.  The phase is a
  // single-wave instruction stream (issue bound: a wave costs the same with 16 or 64 active lanes); the four per-gate tails
  // (product, address, store, three-way split into the A image: ~19 instructions each) were 40 % of it.
  constexpr bool GSPLIT = SPEC != 0;
  const int GL = GSPLIT ? 4 * AU : AU;                       // threads that take part in the cell phase
  float dc_state = 0.f, dh_state = 0.f;                    // own units (tid < GL: unit tid % AU)
  constexpr int PFL = 2;                                   // fl elements prefetched per thread (PFL*ANT >= Ti*F typically)
  float pf_alprev = 0.f, pf_a = 0.f, pf_al = 0.f, pf_a2 = 0.f, pf_pq = 0.f, pf_ctx = 0.f, pf_fl[PFL], pf_e1 = 0.f, pf_e2 = 0.f, pf_alm = 0.f;
  uint32_t pf_saf = 0u;                                    // SAF: result of the L2-prefetch load (kept alive, never used)
  const uint16_t* const safp = reinterpret_cast<const uint16_t*>(p.saf);
  float pf_g[4] = {0.f, 0.f, 0.f, 0.f}, pf_cn = 0.f, pf_cp = 0.f, pf_dh = 0.f, pf_dc = 0.f;   // cell inputs (tid < AU), d out
  // Loads of step tn, issued one step ahead and consumed from registers.  They are UNCONDITIONAL (indices clamped
  // into range, out-of-range lanes load a valid element they never use): the memory counter is in-order, and only
  // with branch-free issue can the compiler count exactly which loads a later wait has to cover.
  // element idx of a row whose pointer is wave-uniform: the byte offset is formed in 32 bits BEFORE it meets the pointer, so the load
  // takes the scalar base + a 32-bit vector offset (r5: as `base[(size_t)row * n + idx]` every one of the 17 prefetch loads of a step
  // carried its own 64-bit vector address arithmetic - a hundred instructions on the cell waves, in front of the barrier of (g))
  auto ldu = [](const float* base, unsigned idx) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + idx * 4u); };
  auto prefetch_rows = [&](const auto& p, int tn, int tid) {   // per-row state + d out (needed at the top of step tn)
    // (unconditional on purpose, also in waves that hold no consumer of a value: behind a branch - even a wave-uniform one -
    // the wait-count pass drains the loads at the join, measured: phase (a) 1.6 -> 2.4 us)
    const size_t bn = (size_t)b * Td + tn;
    const unsigned tr = (unsigned)min(tid, Ti - 1);
    const unsigned tcx = (unsigned)min(tid, CT - 1);
    pf_dc = ldu(dout + (size_t)tn * OW + A, tcx);
    pf_ctx = ldu(fout + (size_t)tn * OW + A, tcx);      // the forward's context of step tn
    const float* alp_row = p.align1 + (tn > 0 ? bn - 1 : bn) * Ti;
    pf_alprev = ldu(alp_row, tr);
    pf_alm = ldu(alp_row, (unsigned)max((int)tr - 1, 0));     // row tid - 1 (the sums of (a) run from registers)
    if (tn == 0) { pf_alprev = tid == 0 ? 1.f : 0.f; pf_alm = tid == 1 ? 1.f : 0.f; }
    pf_a = ldu(p.a1 + bn * Ti, tr); pf_al = ldu(p.align1 + bn * Ti, tr); pf_a2 = ldu(p.align2 + bn * Ti, tr);
    // external gradients wrt the alignments (tests; NULL in training): ALWAYS loaded - through a stand-in pointer and a zero
    // factor when absent.  A conditional load in the phases that consume them made the wait-count pass put s_waitcnt vmcnt(0)
    // at the join: phase (b) then waited ~0.45 us per step for the NEXT step's prefetch loads (r4, found in the ISA listing)
    {
      const float* e1p = pb.dalign1 ? pb.dalign1 : p.align1;
      const float* e2p = pb.dalign2 ? pb.dalign2 : p.align1;
      pf_e1 = ldu(e1p + bn * Ti, tr); pf_e2 = ldu(e2p + bn * Ti, tr);
    }
    if constexpr (SAF) {
      // pull the factor rows of step tn into L2: lane l of wave w touches 64-byte piece l & 7 of own row w + AW * (l >> 3)
      // (5 rows x 512 bytes per wave; clamped to valid rows - a hit costs nothing)
      const unsigned lane_ = (unsigned)tid & 63u, w_ = (unsigned)tid >> 6;
      const unsigned row = min((unsigned)c + (unsigned)C * (w_ + (unsigned)AW * (lane_ >> 3)), (unsigned)Ti - 1u);
      pf_saf = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(safp + bn * Ti * UQ) + (row * (unsigned)UQ + 32u * (lane_ & 7u)) * 2u);
    } else {
#pragma unroll
      for (int u = 0; u < PFL; ++u) pf_fl[u] = p.fl[bn * Ti * F + (unsigned)min(tid + u * ANT, Ti * F - 1)];
      pf_pq = p.pq[bn * UQ + (unsigned)min(tid, UQ - 1)];
    }
  };
  auto prefetch_cell = [&](const auto& p, int tn, int tid) {   // cell inputs of the own units (needed by phase (g) of step tn)
    const size_t bn = (size_t)b * Td + tn;
    const unsigned j = (unsigned)(c * AU + (GSPLIT ? (tid & (AU - 1)) : min(tid, AU - 1)));
    const float* gr = p.gates + bn * G;
    pf_g[0] = ldu(gr, j); pf_g[1] = ldu(gr, A + j); pf_g[2] = ldu(gr, 2 * A + j); pf_g[3] = ldu(gr, 3 * A + j);
    pf_cn = ldu(p.cnew + bn * A, j);
    pf_cp = ldu(p.cstate + (tn > 0 ? bn - 1 : bn) * A, j);
    if (tn == 0) pf_cp = 0.f;
    pf_dh = ldu(dout + (size_t)tn * OW, j);
  };
  // (e) location conv backward (redundant in every member): dac = carry for a_{t-1} from the gathered d fl rows.
  //     The KW filter taps are split into (up to) 3 groups handled by different threads: partial sums dac[part][s].
  //     Runs inside the wait of the step's last exchange (Xh): dac is first needed by the next step's (b) / (c).  (r3: running it
  //     beside the cell phase on the waves that hold no cell unit was measured - the cell phase grew by what the window lost.)
  // SAF build (registers to spare): the filter taps of this thread's tap group live in registers for the whole launch - the
  // group is a function of the thread index only - so the window work is one LDS stream (d fl rows) instead of two
  constexpr int CVT = 4;                          // taps per group at most (KW = 10 in three groups: 3, 3, 4)
  float Fr[CVT * F];
  if constexpr (SAF) {
    const int tix = threadIdx.x, np = min(3, ANT / Ti), jb1 = KW / np, jb2 = 2 * KW / np;
    const int part = tix >= 2 * Ti ? 2 : (tix >= Ti ? 1 : 0);
    const int j0 = part == 0 ? 0 : (part == 1 ? jb1 : jb2), j1 = part + 1 == np ? KW : (part == 0 ? jb1 : jb2);
#pragma unroll
    for (int q = 0; q < CVT; ++q)
#pragma unroll
      for (int k = 0; k < F; ++k) Fr[q * F + k] = (part < np && j0 + q < j1) ? p.locF[(j0 + q) * F + k] : 0.f;
  }
  auto conv_bwd = [&](int tix, int nth) {       // tix: thread index within the nth threads that run it
    const int np = min(3, nth / Ti);              // tap groups that fit the threads (nth >= Ti: see the check)
    const int jb1 = KW / np, jb2 = 2 * KW / np;
    const int part = tix >= 2 * Ti ? 2 : (tix >= Ti ? 1 : 0), s = tix - part * Ti;
    if (part < np && s < Ti) {
      const int j0 = part == 0 ? 0 : (part == 1 ? jb1 : jb2), j1 = part + 1 == np ? KW : (part == 0 ? jb1 : jb2);
      float g = 0.f;
      if constexpr (SAF) {                        // (called with nth == ANT: the groups match Fr; padded taps carry zero weights
        //                                           and read the zero rows that surround d fl)
#pragma unroll
        for (int q = 0; q < CVT; ++q) {
          const int tt = s - min(j0 + q, KW - 1) + PL;
#pragma unroll
          for (int k = 0; k < F; ++k) g += dfl[tt * F + k] * Fr[q * F + k];
        }
      } else
      for (int jj = j0; jj < j1; ++jj) {       // rows outside [0, len) are zero (never written): no bounds test
        const int tt = s - jj + PL;
#pragma unroll
        for (int k = 0; k < F; ++k) g += dfl[tt * F + k] * Fs[jj * F + k];
      }
      dac[part * T4 + s] = cumul ? dac[part * T4 + s] + g : g;
    }
  };
  // one launch over several pipeline chunks: wait (bounded) until the producer stream has published the incoming
  // gradients of chunk k, then drop whatever this CU may have cached of them
  auto wait_ready = [&](uint32_t need) {
    if (threadIdx.x == 0 && !*dead) {
      unsigned spins = 0;
      while (__hip_atomic_load((const gu32*)cb.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(20);
        if (++spins > (1u << 22)) {     // ~5 s
          __hip_atomic_store((gu32*)err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          *dead = 1;
          break;
        }
      }
    }
    // ONE wave invalidates (the vector L1 belongs to the CU, the L2 to the XCD: eight invalidates per workgroup bought nothing
    // and cost 0.04 ms per step), in front of the barrier that releases the others
#ifdef SATT_ACQ_ALL
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
    if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
#endif
  };
  // NLOC rows of step tn for the own rows of M tile mt (one wave): see the comment at `ub`
  // Three waves (5..7: no share in the d ctx sums of (a) nor in the value-row MFMAs of (b)) take one M tile each at the TOP of the
  // step: all K-tile pieces of the rows requested at once (L2 hits: pulled during the previous step), one MFMA chain, done before
  // the barrier in front of phase (c) that consumes the rows.
  // (both halves run behind the barrier of (a), beside the value-row MFMAs of waves 0..2 in (b): requested at the top of the step
  //  the 28 registers of the pieces spilled, and run as a whole there the chain delayed that barrier)
  h8_t nsv[KT1];
  auto nloc_load = [&](int tn, int mt, int lane) {
    if constexpr (VMF) {
      const size_t bn = (size_t)b * Td + tn;
      const unsigned tt = (unsigned)min(c + C * (mt * 16 + (lane & 15)), Ti - 1);       // clamped: rows >= nown are never read
      // (scalar base + 32-bit byte offset, see ldu; 4 h8 per K tile)
      const char* rowb = reinterpret_cast<const char*>(safp + bn * Ti * UQ);
      const unsigned off = (tt * (unsigned)UQ + (unsigned)(lane >> 4) * 8u) * 2u;
#pragma unroll
      for (int q = 0; q < KT1; ++q) nsv[q] = *reinterpret_cast<const h8_t*>(rowb + (off + (unsigned)q * 64u));
    }
  };
  auto nloc_mfma = [&](int mt, int lane) {
    if constexpr (VMF) {
      typedef float f4_t __attribute__((ext_vector_type(4)));
      const h8_t* ubk = ub + lane;
      const h8_t quarter = {0.25f16, 0.25f16, 0.25f16, 0.25f16, 0.25f16, 0.25f16, 0.25f16, 0.25f16};
      f4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < KT1; ++q) {
        const h8_t f = quarter - nsv[q] * nsv[q];                   // r (1 - r) from the saved s = r - 1/2
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f, ubk[q * 64], acc, 0, 0, 0);
      }
      float* dst = nl + (mt * 16 + (lane >> 4) * 4) * 16 + (lane & 15);                     // D[m = 4 (l >> 4) + r][n = l & 15]
      dst[0] = acc[0]; dst[16] = acc[1]; dst[32] = acc[2]; dst[48] = acc[3];
    }
  };
  int bidx = 0;
  int next_lo = -1;
  if (cb.ready) {
    // word 1 behind `ready`: workgroups of this launch that are RESIDENT (r6).  For batches whose LSTM cluster launches must share
    // CUs (Engine._layers_fit_side_by_side, B > 32) the LSTM stream waits for B * C here before its first launch: this kernel needs
    // whole CUs, and a workgroup of it that is still pending while LSTM workgroups spread over the empty CUs can wait in a circle
    // with them (the dispatcher holds CUs back for the pending workgroup, the LSTM launch cannot become resident as a whole)
    if (threadIdx.x == 0) atomicAdd((unsigned int*)cb.ready + 1, 1u);
    wait_ready(1u);
    next_lo = cb.nbound > 0 ? cb.bound[0] : -1;
  }
  prefetch_rows(p, cb.t1 - 1, threadIdx.x);
  prefetch_cell(p, cb.t1 - 1, threadIdx.x);
  // hand-off record between chunks: [C*NWP: d[ctx|h] (first KR used)] [A: dc_state] [A: dh_state] [Ti: dac] [Ti: dalc]
  // [4: d u carried by the transition agent]
  float* stb = cb.state ? cb.state + (size_t)b * (C * NWP + 2 * A + 2 * Ti + 4) : nullptr;
  // transition agent (generic instantiation only; see the forward kernel): du_s[0] = gradient wrt the transition
  // probability of the step processed last (step t+1), written by wave 0 in phase (c) and consumed at the top of step t
  const bool agent = !SPEC && p.agentW != nullptr && !unit_w;
  float* du_s = smem + L.dead + 4;
  if (threadIdx.x == 0) du_s[0] = (agent && cb.t1 < Td) ? stb[C * NWP + 2 * A + 2 * Ti] : 0.f;
  if (cb.t1 < Td) {        // continue from the chunk that processed steps >= t1
    const int tid = threadIdx.x;
    __syncthreads();
    for (int i = tid; i < KR; i += ANT) cgx[i] = stb[i];
    if (tid < GL) { const int u = tid % AU; dc_state = stb[C * NWP + c * AU + u]; dh_state = stb[C * NWP + A + c * AU + u]; }
    for (int i = tid; i < Ti; i += ANT) { dac[i] = stb[C * NWP + 2 * A + i]; dalc[i] = stb[C * NWP + 2 * A + Ti + i]; }
  }
  __syncthreads();

  PROF_DECL;
  const int t_first = cb.t1 - 1, t_last = cb.t0;
  for (int t = t_first; t >= t_last; --t) {
    PROF(0);
    // The kernel arguments are re-read from the kernarg segment inside every step (scalar loads next to their use)
    // instead of being held in scalar registers for the whole loop: ~60 pointers and sizes do not fit the 100 SGPRs
    // next to everything else, and a spilled scalar costs a v_readlane (a VALU slot) per use.  The opaque copy of the
    // segment pointer keeps the loads inside the loop body.
    typedef const __attribute__((address_space(4))) satt_attn_cluster_bwd_params KArgs;
    KArgs* kq = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kq));
    const auto& cb = *kq;
    const auto& pb = cb.b;
    const auto& p = pb.f;
    int oz = 0;                                            // opaque per-step zero (see the forward kernel)
    asm volatile("" : "+v"(oz));
    const int tid = (int)threadIdx.x + oz, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(tid >= 0 && tid < ANT && wave >= 0 && wave < AW);   // ranges lost through the opaque zero
    const int d0 = lane * NQ;
    const bool actU = d0 < U1, actV = d0 < V1;
    const size_t bt = (size_t)b * Td + t;
    const uint32_t tag = (uint32_t)(t + 1);
    u64* wp = wsb + ((size_t)(t & 1) * p.B + b) * WL.per_parity;
    if (t == next_lo && bidx + 1 < cb.nbound) wait_ready((uint32_t)(bidx + 2));   // last step of chunk bidx: its prefetches read the next chunk
    // (a) forward state of this step: prefetched into registers one step ahead (Ti <= ANT: see the check)
    const float ext1 = pb.dalign1 ? pf_e1 : 0.f, ext2 = pb.dalign2 ? pf_e2 : 0.f;      // (selects, no branch)
    if (tid < Ti) { alprev[tid] = pf_alprev; a[tid] = pf_a; al[tid] = pf_al; a2[tid] = pf_a2; exts[tid] = ext1; exts[T4 + tid] = ext2; }
    if constexpr (!SAF) {
#pragma unroll
      for (int u = 0; u < PFL; ++u) { const int e = tid + u * ANT; if (e < Ti * F) fl[e] = pf_fl[u]; }
      for (int e = tid + PFL * ANT; e < Ti * F; e += ANT) fl[e] = p.fl[bt * Ti * F + e];
      if (tid < UQ) pqv[tid] = pf_pq;
    } else {
      asm volatile("" :: "v"(pf_saf));                      // the prefetch load's destination stays reserved until here
    }
    const float cg0 = pf_g[0], cg1 = pf_g[1], cg2 = pf_g[2], cg3 = pf_g[3], ccn = pf_cn, ccp = pf_cp, cdh = pf_dh;
    const float ctxv = pf_ctx;                              // this step's forward context column (phase (b))
    // transition agent: u of this step (recursion), and d z of this step's prediction of u_{t+1}
    float ut = 0.5f, dz = 0.f;
    if (agent) {
      if (t > 0) ut = p.ustate[bt];
      if (t + 1 < Td) { const float un = p.ustate[bt + 1]; dz = du_s[0] * un * (1.f - un); }
      if (c == 0 && tid == 0) gst(pb.dz + bt, dz);
    }
    float dctx_own = 0.f;                                       // d ctx of column tid (the sums below)
    if (tid < CT) {
      float g = pf_dc;
      if constexpr (SPEC != 0) {       // the four partials requested together (as a loop: four exposed LDS latencies on the chain)
        static_assert(SpecDimsOf<SPEC>::C == 4, "");
        const float c0 = cgx[tid], c1 = cgx[KR + tid], c2 = cgx[2 * KR + tid], c3 = cgx[3 * KR + tid];
        g = (((g + c0) + c1) + c2) + c3;
      } else
      for (int k = 0; k < C; ++k) g += cgx[k * KR + tid];
      if (agent && tid < V1) g += dz * p.agentW[tid];          // d ctx1 through the agent's Dense
      dctx[tid] = g; dctx_own = g;
      if constexpr (VMF) xs_put(dcs, DCS, tid, g);              // A rows (hi / mid / lo) of the value-row product in (b)
      if (c == 1 % C) gst_s(pb.dctx + bt * CT, (unsigned)tid, g);
    }
    // value rows of the own memory rows i0 + u*AW for phase (b): they do not depend on the carried gradient, so they
    // are requested here (after the waits on the prefetched registers) and their L2 latency overlaps the barrier
    float4 vr[RBV]; float vw2[RBV];
    auto load_vrows = [&](int i0) {
#pragma unroll
      for (int u = 0; u < RBV; ++u) {
        const unsigned tt = (unsigned)min(c + C * (i0 + u * AW), Ti - 1);       // clamped: rows >= nown are discarded
        vr[u] = *reinterpret_cast<const float4*>(values1 + (unsigned)min(d0, V1 - NQ) + (size_t)tt * V1);
        vw2[u] = values2[(size_t)tt * V2 + (unsigned)max(min(lane, V2 - 1), 0)];
      }
    };
    if constexpr (!VMF) load_vrows(wave);
    // Next step's forward state and cell inputs are requested HERE, a whole step ahead of their use: the vector-memory counter
    // is in-order and on gfx9 counts loads and stores alike, so a poll of an exchange waits for every load its wave issued
    // before it.  Issued inside an exchange window (round 2: "the loads fly during the wait") these HBM / MALL reads - saved
    // forward tensors, long evicted from L2 - sat IN FRONT of the polls and added their latency to the exchange (Xd 0.65 ->
    // 1.5 us in the trace).  From here the next poll is ~5 us away (phases (b)-(d)).
    {   // the four sums, per-thread terms of row tid / context column tid (Ti, CT <= ANT), reduced per wave -> scal[wave][4]
      const int tc = min(tid, Ti - 1);
      const float okr = tid < Ti ? 1.f : 0.f;
      // (r4: from the step's prefetched REGISTERS and the carries of the previous step, in front of the barrier of (a) - the wave
      // reductions overlap the other waves' staging instead of standing between the value-row MFMAs and phase (c))
      const float ap = pf_alprev, am = pf_alm, av = okr * pf_a, alv = okr * pf_al, a2v = okr * pf_a2;
      const float dcs = dac[tc] + dac[T4 + tc] + dac[2 * T4 + tc];
      const float e1 = ext1, e2 = ext2;                      // row tid of this step (registers: see prefetch_rows)
      const float wv = unit_w ? 1.f : (1.f - ut) * ap + (tid > 0 ? ut : 0.f) * am + 1e-7f;
      const float dcx = tid < CT ? dctx_own * ctxv : 0.f;
      float r4[4];
      r4[0] = (tid < V1 ? dcx : 0.f) + (dalc[tc] + e1) * alv;
      r4[1] = dcs * av;
      r4[2] = (tid >= V1 ? dcx : 0.f) + e2 * a2v;
      r4[3] = wv * av;
#ifdef SATT_EXP_SUMS_TRANSPOSE
      const float tot = wave_sum_transpose<4>(r4);                // lane l: total of slot l & 3
      if (lane < 4) scal[wave * 4 + lane] = tot;
#else
      // r4: DPP butterflies + lane reads (wave_sum_multi) instead of the transposing reduction: that one needs four dependent trips
      // through the LDS pipe (three swizzles + a bpermute, ~100 cycles each) - fewer instructions, but this phase is one short
      // dependent chain between two barriers, not an issue-bound loop
      wave_sum_multi<4>(r4);
      if (lane == 0) *reinterpret_cast<float4*>(scal + wave * 4) = make_float4(r4[0], r4[1], r4[2], r4[3]);
#endif
    }
#ifdef SATT_PF_TOP
#ifndef SATT_EXP_NOPF_ROWS      // (timing experiments only: tools/build_variant.sh)
    prefetch_rows(p, max(t - 1, cb.t0), tid);
#endif
#ifndef SATT_EXP_NOPF_CELL
    prefetch_cell(p, max(t - 1, cb.t0), tid);
#endif
#endif
    lds_barrier();
    PROF(1); BTRACE(cb.t1 - 1 - t, 0);
    // (b) raw d alpha / d a2 of the own rows through the contexts: d ctx . value row -> LDS.  NO exchange follows (r3): the
    //     softmax / forward-attention backward below needs three sums over ALL rows, and each of them is known to every
    //     member without the other members' rows:
    //       s1 = sum_t' d alpha[t'] alpha[t'],  d alpha = (d ctx1 . V1[t']) + carry + external
    //          = d ctx1 . (sum_t' alpha[t'] V1[t']) + sum (carry + ext) alpha = d ctx1 . ctx1_t + ...   (ctx1_t: saved by the forward)
    //       s3 = d ctx2 . ctx2_t + sum ext2 a2                                                          (same identity)
    //       s2 = sum_t' d a[t'] a[t'],  d a = d g w + conv carry,  d g = (d alpha - s1) / S,  g = w a = S alpha
    //          = (sum d alpha alpha - s1 sum alpha) + sum conv-carry a = sum conv-carry a               (sum alpha = 1: the first term is 0)
    //     The contexts the forward saved were formed from the same value rows this phase multiplies with (the engine passes the
    //     bf16-rounded rows in bf16 mode), so the identities hold to fp32 rounding.  What used to be the exchange Xb plus a
    //     one-wave pass over all rows is now: row products (all waves) + four wave sums, one barrier, 40 own rows.
    if constexpr (VMF) {
      // r4: on the matrix cores.  Wave w < 3 owns N tile w (own rows 16 w .. 16 w + 15) over all K tiles of d ctx: one chained
      // accumulator per source, B tiles straight from the resident bf16 row images - 9 LDS-fed MFMAs instead of five float4 row
      // loads from L2, 25 FMAs and a 16-value transposing wave reduction per lane (and ten vector-memory instructions less in
      // front of the exchange polls).  Exact: the rows are bf16 values, d ctx is split 3-way, fp32 accumulation.
      if (wave >= AW - VMF_ROWS / 16) { nloc_load(t, wave - (AW - VMF_ROWS / 16), lane); nloc_mfma(wave - (AW - VMF_ROWS / 16), lane); }
      if (wave == VMF_ROWS / 16) {
        // r5: the totals of the four sums and 1 / S once, on a wave this phase leaves idle: phase (c) - every wave runs the same
        // instruction stream there, two waves per SIMD, issue bound - reads five values instead of summing eight partial
        // quadruples and dividing in every lane (same order, same division: bit-identical)
        float s1 = 0.f, s2 = 0.f, s3 = 0.f, S = 0.f;
#pragma unroll
        for (int w = 0; w < AW; ++w) {
          const float4 q = *reinterpret_cast<const float4*>(scal + w * 4);
          s1 += q.x; s2 += q.y; s3 += q.z; S += q.w;
        }
        if (lane == 0) { *reinterpret_cast<float4*>(scal + 4 * AW) = make_float4(s1, s2, s3, S); scal[4 * AW + 4] = 1.f / S; }
      }
      if (wave < VMF_ROWS / 16) {
        const int row = wave * 16 + (lane & 15);
        const uint16_t* arow = dcs + min(lane & 15, 3) * DCS + (lane >> 4) * 8;
        const uint16_t* brow = vs1 + row * VS1 + (lane >> 4) * 8;
        // (two halves of four K tiles through the same operand registers: sixteen live 16-byte operands spilled)
        bf16x8_t za[4]; i32x4_t vb[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          za[kt] = *reinterpret_cast<const bf16x8_t*>(arow + kt * 32);
          vb[kt] = *reinterpret_cast<const i32x4_t*>(brow + kt * 32);
        }
        f32x4_t c1, c2 = {0.f, 0.f, 0.f, 0.f};
        mfma41z_v<false>(c1, za[0], za[1], za[2], za[3], vb[0], vb[1], vb[2], vb[3]);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          za[kt] = *reinterpret_cast<const bf16x8_t*>(arow + (kt + 4) * 32);
          vb[kt] = *reinterpret_cast<const i32x4_t*>(brow + (kt + 4) * 32);
        }
        if (V2 > 0) {
          const bf16x8_t a8 = *reinterpret_cast<const bf16x8_t*>(arow + V1);
          const i32x4_t b8 = *reinterpret_cast<const i32x4_t*>(vs2 + row * VS2 + (lane >> 4) * 8);
          mfma41_v<false>(c1, za[0], za[1], za[2], za[3], vb[0], vb[1], vb[2], vb[3]);
          mfma_bf16_vreg(c2, a8, b8);
          mfma_cover(c1);
        } else {
          mfma41_v(c1, za[0], za[1], za[2], za[3], vb[0], vb[1], vb[2], vb[3]);
        }
        if (lane < 16 && row < nown) { draw[row] = c1[0] + c1[1] + c1[2]; draw[T4 + row] = c2[0] + c2[1] + c2[2]; }
      }
      PROF(9);
    } else {
      float dcr[NQ];
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) dcr[qq] = (d0 + qq) < V1 ? dctx[d0 + qq] : 0.f;
      const float dc2 = lane < V2 ? dctx[V1 + lane] : 0.f;
      for (int i0 = wave; i0 < nown; i0 += RBV * AW) {
        if (i0 != wave) load_vrows(i0);
        float red16[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) red16[q] = 0.f;
#pragma unroll
        for (int u = 0; u < RBV; ++u) {
          red16[u] = actV ? vr[u].x * dcr[0] + vr[u].y * dcr[1] + vr[u].z * dcr[2] + vr[u].w * dcr[3] : 0.f;
          red16[8 + u] = vw2[u] * dc2;        // dc2 = 0 beyond V2
        }
        const float tot = wave_sum_transpose<16>(red16);         // lane l: total of value l & 15
        const float s2r = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + 8) & 63) << 2, __float_as_int(tot)));
        if (lane < RBV) {
          const int i = i0 + lane * AW;
          if (i < nown) { draw[i] = tot; draw[T4 + i] = s2r; }
        }
      }
    }
    // SAF: the factor rows of this wave's first energy-backward pass (own rows wave + AW * u, u < RBB), requested one phase
    // ahead (they were pulled into L2 during the previous step); the second pass's rows follow at the start of (d)
    uint2 sq[RBB]; uint32_t sq2[RBB];
    auto load_saf = [&](int i0, uint2 (&q)[RBB], uint32_t (&q2)[RBB], int n) {
#pragma unroll
      for (int u = 0; u < RBB; ++u) {
        if (u < n) {
          const unsigned tt = (unsigned)min(c + C * (i0 + u * AW), Ti - 1);     // clamped: rows >= nown are discarded
          const uint16_t* row = safp + (bt * Ti + tt) * UQ;
          q[u] = *reinterpret_cast<const uint2*>(row + (unsigned)min(d0, U1 - NQ));
          q2[u] = U2 > 0 ? row[U1 + (unsigned)min(lane, U2 - 1)] : (uint16_t)0;
        }
      }
    };
    if constexpr (SAF) load_saf(wave, sq, sq2, RBB);
#ifndef SATT_SR_LATE
    uint2 sr[RBB]; uint32_t sr2[RBB];
    if constexpr (SAF) load_saf(wave + RBB * AW, sr, sr2, RBB - 1);
#endif
    PROF(10);
    PROF(11);
    BTRACE(cb.t1 - 1 - t, 1);
    lds_barrier();
    PROF(2); BTRACE(cb.t1 - 1 - t, 2);
    // (c) forward-attention recursion + softmax backward of the own rows, distributed so that NO workgroup barrier separates it
    //     from (d): wave w finishes the rows i = w + AW*l (lane l) - exactly the rows its (d) iterations process (i0 = w, step
    //     AW) and whose raw values its own lanes wrote in (b) - so the hand-off de1 / da2 is wave-local (LDS is in-order per wave).
    //     The d w values (carry for alpha_{t-1}, needed by the NEXT step) are published here and travel with the exchange Xd.
    {
      float s1 = 0.f, s2 = 0.f, s3 = 0.f, S = 0.f, invS;
      if constexpr (VMF) {       // (totals formed in phase (b), see there)
        const float4 q = *reinterpret_cast<const float4*>(scal + 4 * AW);
        s1 = q.x; s2 = q.y; s3 = q.z; S = q.w; invS = scal[4 * AW + 4];
      } else {
#pragma unroll
        for (int w = 0; w < AW; ++w) {
          const float4 q = *reinterpret_cast<const float4*>(scal + w * 4);
          s1 += q.x; s2 += q.y; s3 += q.z; S += q.w;
        }
        invS = 1.f / S;
      }
      // VMF: 8 lanes per row (lane = 8 * row slot + k) compute the row redundantly - the instruction stream is the same, more
      // lanes are active - and lane k < F publishes d fl[tt][k] = d e * N[row][k], lane F the d w carry, lanes 6 / 7 the row's
      // d e / d a2 (LDS hand-off to (d) + global outputs): five rows x 8 stores issue as one store instruction each
      const int rl = VMF ? (lane >> 3) : lane, kk = VMF ? (lane & 7) : 0;
      const int i = wave + AW * rl, tt = c + C * i;
      if (i < nown_max && tt < Ti && (!VMF || rl < RBV)) {
        float de = 0.f, d2 = 0.f, dw = 0.f;
        if (i < nown) {
          const float ap = alprev[tt], am = alprev[max(tt - 1, 0)];
          const float wv = unit_w ? 1.f : (1.f - ut) * ap + (tt > 0 ? ut : 0.f) * am + 1e-7f;
          const float e1 = exts[tt], e2 = exts[T4 + tt];
          const float dalp = ((draw[i] + dalc[tt] + e1) - s1) * invS;
          const float da = dalp * wv + (dac[tt] + dac[T4 + tt] + dac[2 * T4 + tt]);
          de = a[tt] * (da - s2);
          d2 = a2[tt] * ((draw[T4 + i] + e2) - s3);
          dw = unit_w ? 0.f : dalp * a[tt];
          if constexpr (VMF) {
            const float2 nv = *reinterpret_cast<const float2*>(nl + i * 16 + 2 * min(kk, F - 1));
            const float vs = de * (nv.x + nv.y);
            // (r5: scalar bases + 32-bit indices, see gput_s: this phase is every wave's same instruction stream)
            u64* const xrow = wp + WL.xd + C * UQ;
            if (kk < F) { gput_s(xrow, (unsigned)(tt * (F + 1) + kk), tag, vs, same_xcd); gst_s(pb.dfl + bt * Ti * F, (unsigned)(tt * F + kk), vs); }
            else if (kk == F) gput_s(xrow, (unsigned)(tt * (F + 1) + F), tag, dw, same_xcd);
          } else {
            gput(wp + WL.xd + C * UQ + tt * (F + 1) + F, tag, dw, same_xcd);
          }
        }

        if (!VMF || kk == 6) { de1[tt] = de; da2[tt] = d2; gst_s(pb.de1 + bt * Ti, (unsigned)tt, de); }
        if (!VMF || kk == 7) gst_s(pb.de2 + bt * Ti, (unsigned)tt, d2);
      }
      // (the reads of de1 / da2 in (d) may alias these stores, so the compiler keeps them behind; the hardware runs the LDS
      // operations of a wave in order.  No asm memory clobber here: it makes the wait-count pass drain EVERY outstanding
      // memory operation of the wave - including the next step's prefetch loads, an HBM round trip)
    }
    PROF(3); BTRACE(cb.t1 - 1 - t, 3);
    // (d) energy backward for own rows: partial d pq, d location-features of own rows; publish both
    //     g = de * v * (1 - tanh^2) = de * (4 v) * r * (1 - r); packed fp32 math on unit pairs
    {
      v2f vq01, vq23, Us01[F], Us23[F], pqs01, pqs23;
      {
        const float4 tv = *reinterpret_cast<const float4*>(tab + d0), tb = *reinterpret_cast<const float4*>(tab + 64 * NQ + d0);
        vq01 = (v2f){tv.x, tv.y}; vq23 = (v2f){tv.z, tv.w};
#pragma unroll
        for (int k = 0; k < F; ++k) {
          const float4 tu = *reinterpret_cast<const float4*>(tab + (2 + k) * 64 * NQ + d0);
          Us01[k] = (v2f){tu.x, tu.y}; Us23[k] = (v2f){tu.z, tu.w};
        }
        const float4 q4 = *reinterpret_cast<const float4*>(pqv + min(d0, UQ - NQ));   // clamped: zero weight beyond U1
        pqs01 = (v2f){TS * (q4.x + tb.x), TS * (q4.y + tb.y)};
        pqs23 = (v2f){TS * (q4.z + tb.z), TS * (q4.w + tb.w)};
      }
      const float v2q = tab[(2 + F) * 64 * NQ + lane];
      const v2f ts2 = (v2f){TS, TS}, one2 = (v2f){1.f, 1.f};
      v2f dpq01 = (v2f){0.f, 0.f}, dpq23 = (v2f){0.f, 0.f};
      float dpq2a = 0.f;
      float* dflg = pb.dfl + bt * Ti * F;
      // publish the d fl values of one pass: transposing reduction of the RBB x F per-lane partials, lane (u, k) stores
      auto publish_dfl = [&](int i0, float (&dfp)[RBB * F]) {
        float d16[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) d16[q] = q < RBB * F ? dfp[q] : 0.f;
        const float v = wave_sum_transpose<16>(d16);              // lane l: total of value l & 15 = (row u, filter k)
        if (lane < RBB * F) {
          const int u = lane / F, k = lane - u * F, i = i0 + u * AW, tt = c + C * i;
          const float vs = v * (1.f / TS);
          if (i < nown) { gput(wp + WL.xd + C * UQ + tt * (F + 1) + k, tag, vs, same_xcd); gst(dflg + tt * F + k, vs); }
        }
      };
      if constexpr (SAF) {
        // rows from the saved s = r - 1/2: g = d e * (4 v) * (1/4 - s^2), no keys, no location term, no exp2 / rcp
        typedef __attribute__((ext_vector_type(2))) __fp16 h2;
        auto pass = [&](auto full_tag, int i0, const uint2 (&q)[RBB], const uint32_t (&q2)[RBB], int n) {
          constexpr bool FULLW = decltype(full_tag)::value;      // every row of the pass exists: no per-row test (see the forward kernel)
#pragma unroll
          for (int u = 0; u < RBB; ++u) {
            const int i = i0 + u * AW, tt = c + C * i;
            if (u < n && (FULLW || i < nown)) {
              const float de = de1[tt], dq2 = da2[tt];
              union { uint32_t w; h2 h; } c0, c1, c2;
              c0.w = q[u].x; c1.w = q[u].y; c2.w = q2[u];
              const v2f s01 = (v2f){(float)c0.h.x, (float)c0.h.y}, s23 = (v2f){(float)c1.h.x, (float)c1.h.y};
              const v2f quarter2 = (v2f){0.25f, 0.25f};
              const v2f f01 = quarter2 - s01 * s01, f23 = quarter2 - s23 * s23;     // r (1 - r) from s = r - 1/2
              const v2f de2v = (v2f){de, de};
              dpq01 += (de2v * vq01) * f01; dpq23 += (de2v * vq23) * f23;           // lanes beyond U1: vq = 0
              const float s2 = (float)c2.h.x;
              dpq2a += dq2 * v2q * (0.25f - s2 * s2);                            // lanes beyond U2: v2q = 0
            }
          }
          // (r4: the d location-feature values of these rows were published by phase (c) from the NLOC rows)
        };
#ifdef SATT_SR_LATE
        uint2 sr[RBB]; uint32_t sr2[RBB];
        load_saf(wave + RBB * AW, sr, sr2, RBB - 1);           // second pass: own rows wave + AW * (RBB + u), u < RBB - 1
#endif
#ifdef SATT_PF_IN_D      // (experiment: the next step's prefetch behind the last loads this step consumes from registers)
        prefetch_rows(p, max(t - 1, cb.t0), tid);
        prefetch_cell(p, max(t - 1, cb.t0), tid);
#endif
        if (wave + (RBB - 1) * AW < nown) pass(std::true_type{}, wave, sq, sq2, RBB); else pass(std::false_type{}, wave, sq, sq2, RBB);
        if (wave + (2 * RBB - 2) * AW < nown) pass(std::true_type{}, wave + RBB * AW, sr, sr2, RBB - 1);
        else if (wave + RBB * AW < nown) pass(std::false_type{}, wave + RBB * AW, sr, sr2, RBB - 1);
      } else {
      const float pq2 = TS * pqv[U1 + min(lane, U2 - 1)];     // lanes beyond U2: zero weight v2q
      for (int i0 = wave; i0 < nown; i0 += RBB * AW) {
        float dfp[RBB * F];
#pragma unroll
        for (int u = 0; u < RBB; ++u) {
          const int i = i0 + u * AW, tt = c + C * i;
#pragma unroll
          for (int k = 0; k < F; ++k) dfp[u * F + k] = 0.f;
          if (i < nown) {
            const float de = de1[tt];
            float f[F];
#pragma unroll
            for (int k = 0; k < F; ++k) f[k] = fl[tt * F + k];
            float kk[NQ];
            load_key4u<KLDS>(keys1 + (KLDS ? 0 : (size_t)tt * U1), K1s, KLDS ? i : 0, U1, d0, kk);
            const float k2 = load_key1u<KLDS>(keys2 + (KLDS ? 0 : (size_t)tt * U2), K2s, KLDS ? i : 0, U2, lane);
            const float dq2 = da2[tt];
            v2f x01 = (v2f){kk[0], kk[1]} * ts2 + pqs01, x23 = (v2f){kk[2], kk[3]} * ts2 + pqs23;
#pragma unroll
            for (int k = 0; k < F; ++k) {
              const v2f f2 = (v2f){f[k], f[k]};
              x01 = f2 * Us01[k] + x01; x23 = f2 * Us23[k] + x23;
            }
            const v2f e01 = (v2f){exp2f_(x01.x), exp2f_(x01.y)} + one2, e23 = (v2f){exp2f_(x23.x), exp2f_(x23.y)} + one2;
            const v2f r01 = (v2f){__builtin_amdgcn_rcpf(e01.x), __builtin_amdgcn_rcpf(e01.y)};
            const v2f r23 = (v2f){__builtin_amdgcn_rcpf(e23.x), __builtin_amdgcn_rcpf(e23.y)};
            const v2f de2v = (v2f){de, de};
            // lanes beyond U1 / U2 hold zero weights (vq, v2q) and finite inputs: their g is exactly 0, no select needed
            const v2f g01 = (de2v * vq01) * (r01 * (one2 - r01));
            const v2f g23 = (de2v * vq23) * (r23 * (one2 - r23));
            dpq01 += g01; dpq23 += g23;
#pragma unroll
            for (int k = 0; k < F; ++k) {
              const v2f sk = g01 * Us01[k] + g23 * Us23[k];
              dfp[u * F + k] = sk.x + sk.y;            // against TS * U: rescaled once after the reduction
            }
            const float r2 = __builtin_amdgcn_rcpf(1.f + exp2f_(TS * k2 + pq2));
            dpq2a += dq2 * v2q * r2 * (1.f - r2);
          }
        }
        publish_dfl(i0, dfp);
      }
      }
      if (actU) {
        float* pw = partial + wave * UQ4 + d0;
        *reinterpret_cast<float4*>(pw) = make_float4(dpq01.x, dpq01.y, dpq23.x, dpq23.y);
      }
      if (lane < U2) partial[wave * UQ4 + U1 + lane] = dpq2a;
    }
    lds_barrier();
    if (tid < UQ) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < AW; ++w) s += partial[w * UQ4 + tid];
      gput_s(wp + WL.xd + c * UQ, (unsigned)tid, tag, s, same_xcd);
    }
    // (r5: the member-specific stores of this kernel - this one, d ctx and d pq of member 1 - spread evenly over the members as in the
    //  forward kernel: launch 2.836 -> 2.846 ms, not kept; the late member of a backward step differs from sample to sample)
    if (c == 0) { float* dflg = pb.dfl + bt * Ti * F; for (int e = tid + len * F; e < Ti * F; e += ANT) gst(dflg + e, 0.f); }
    BTRACE(cb.t1 - 1 - t, 4);
    // Xd: all C partial d pq vectors, and per memory row (rows < len: a contiguous prefix) its F d fl values + its d w value
    // (r5, tried and NOT kept: the row granules - d fl, d w: they feed the NEXT step only - gathered beside the four-wave cell phase
    //  (g) by its idle waves instead of here, d pq alone on the chain: launch 2.88 -> 2.94 ms.  The polls of waves 4..7 stand in
    //  front of their share of the next step's prefetch loads, and the barrier behind (g) waits for them)
    if constexpr (SPEC != 0 && SAF) {
      // r5: the LDS destination of every granule is formed BEFORE the poll (the generic gather divided every row-granule index by
      // F + 1 behind it, on the chain): waves 0..3 take the C x UQ = 4 x 256 partials, waves 4..7 the len x (F + 1) row granules
      static_assert(!SPEC || (SpecDimsOf<SPEC>::C * (SpecDimsOf<SPEC>::U1 + SpecDimsOf<SPEC>::U2) == 1024 && AW == 8), "");
      const int nrow = len * (F + 1), w4 = wave - 4;
      const int base = wave < 4 ? wave * 256 : C * UQ + w4 * 256, cnt = wave < 4 ? 256 : min(256, nrow - w4 * 256);
      if (cnt > 0) {
        const gu64* g[4]; u64 x[4]; float* dst[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = base + min(lane + 64 * q, cnt - 1);
          g[q] = (const gu64*)(wp + WL.xd + i); x[q] = 0;
          const int j = i - C * UQ, row = j / (F + 1), k = j - row * (F + 1);
          dst[q] = wave < 4 ? dpart + i : (k < F ? dfl + row * F + k : dal + row);
        }
        poll_or_die<4>(g, tag, x, lane, err_word, dead);
#pragma unroll
        for (int q = 0; q < 4; ++q) if (lane + 64 * q < cnt) *dst[q] = __uint_as_float((uint32_t)x[q]);
      }
    } else
    gather_span(wp + WL.xd, C * UQ + len * (F + 1), tag, wave, AW, lane,
                [&](int i, float v) {
                  if (i < C * UQ) { dpart[i] = v; return; }
                  const int j = i - C * UQ, row = j / (F + 1), k = j - row * (F + 1);
                  if (k < F) dfl[row * F + k] = v; else dal[row] = v;
                }, err_word, dead);
    lds_barrier();
    PROF(4); BTRACE(cb.t1 - 1 - t, 5);
    if (tid < UQ) {
      float s = 0.f;
      for (int k = 0; k < C; ++k) s += dpart[k * UQ + tid];     // fixed order: identical in every member
      if (agent && tid < U1) s += dz * p.agentW[V1 + tid];     // d pq1 through the agent's Dense
      xs_put(dps, DPS, tid, s);
      if (c == 1 % C) gst_s(pb.dpq + bt * UQ, (unsigned)tid, s);
    }
    // carry for alpha_{t-1} (rows >= len keep d w = 0: never written) and, with the transition agent, d u_t = sum d w * d w / d u
    for (int i = tid; i < Ti; i += ANT) dalc[i] = (1.f - ut) * dal[i] + ut * (i + 1 < Ti ? dal[i + 1] : 0.f);
    if (agent && wave == AW - 1) {
      float sdu = 0.f;
      for (int tt = lane; tt < Ti; tt += 64) sdu += dal[tt] * ((tt > 0 ? alprev[tt - 1] : 0.f) - alprev[tt]);
      sdu = wave_sum(sdu);
      if (lane == 0) du_s[0] = sdu;        // (every thread read du_s[0] at the top of the step, several barriers ago)
    }
    lds_barrier();
    PROF(5); BTRACE(cb.t1 - 1 - t, 6);
    // (f) d query of the own units = d pq x Wq^T[:, own]: K tile = wave, partials reduced by the cell phase
    if (wave < KTU) {
      const uint16_t* prow = dps + min(lane & 15, 3) * DPS + (lane >> 4) * 8 + wave * 32;
      const bf16x8_t av = *reinterpret_cast<const bf16x8_t*>(prow);
      f32x4_t q0, q1, q2, q3;
      mfma14z_a(q0, q1, q2, q3, av, wqT[0], wqT[1], wqT[2], wqT[3]);
      if (lane < 16) {
        float* dst = dqp + wave * 64 + lane;
        dst[0] = q0[0] + q0[1] + q0[2]; dst[16] = q1[0] + q1[1] + q1[2];
        dst[32] = q2[0] + q2[1] + q2[2]; dst[48] = q3[0] + q3[1] + q3[2];
      }
    }
    lds_barrier();
    PROF(6); BTRACE(cb.t1 - 1 - t, 7);
    // (g) LSTM cell backward for the own units
    float dh_direct = 0.f;
    if (GSPLIT && tid < GL) {            // (wave == gate: uniform branches)
      const int u = tid & (AU - 1), j = c * AU + u;
      const uint32_t idx = (uint32_t)bt * (uint32_t)A + (uint32_t)j;
      float kc, kh, pc, ph;
      const uint32_t zct = p.zc_thresh, zht = p.zh_thresh, zsc = p.stream_c, zsh = p.stream_h;   // one batch of loads (see forward)
      const int ztr = p.training;
      const bool keep_c = (satt_hash(seed, zsc, idx) >= zct) | (zct == 0), keep_h = (satt_hash(seed, zsh, idx) >= zht) | (zht == 0);
      if (ztr) {
        kc = keep_c ? 1.f : 0.f; pc = 1.f - kc;
        kh = keep_h ? 1.f : 0.f; ph = 1.f - kh;
      } else {
        kc = 1.f - p.zc; pc = p.zc; kh = 1.f - p.zh; ph = p.zh;
      }
      float dqj = 0.f;
      if constexpr (SPEC != 0) {       // (KTU == 8: the eight partials requested together - as a loop, four exposed LDS latencies)
        float qv[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) qv[w] = dqp[w * 64 + u];
#pragma unroll
        for (int w = 0; w < 8; ++w) dqj += qv[w];
      } else
      for (int w = 0; w < KTU; ++w) dqj += dqp[w * 64 + u];
      const float gi = cg0, gj = cg1, gf = cg2, go = cg3, cn = ccn, cp = ccp;
      const float dhn = cdh + dqj + kh * dh_state;
      dh_direct = ph * dh_state;
      const float tc = tanhf_(cn);
      const float dcn = dhn * go * (1.f - tc * tc) + kc * dc_state;
      dc_state = dcn * gf + pc * dc_state;
      float dz;
      if (wave == 0) dz = dcn * gj * gi * (1.f - gi);
      else if (wave == 1) dz = dcn * gi * (1.f - gj * gj);
      else if (wave == 2) dz = dcn * cp * gf * (1.f - gf);
      else dz = dhn * tc * go * (1.f - go);
      gst_s(pb.dxg + bt * G, (unsigned)(wave * A + j), dz);
      xs_put(dzs, DZS, wave * AU + u, dz);
    } else if (!GSPLIT && tid < AU) {
      const int j = c * AU + tid;
      const uint32_t idx = (uint32_t)bt * (uint32_t)A + (uint32_t)j;
      float kc, kh, pc, ph;
      const uint32_t zct = p.zc_thresh, zht = p.zh_thresh, zsc = p.stream_c, zsh = p.stream_h;   // one batch of loads (see forward)
      const int ztr = p.training;
      const bool keep_c = (satt_hash(seed, zsc, idx) >= zct) | (zct == 0), keep_h = (satt_hash(seed, zsh, idx) >= zht) | (zht == 0);
      if (ztr) {
        kc = keep_c ? 1.f : 0.f; pc = 1.f - kc;
        kh = keep_h ? 1.f : 0.f; ph = 1.f - kh;
      } else {
        kc = 1.f - p.zc; pc = p.zc; kh = 1.f - p.zh; ph = p.zh;
      }
      float dqj = 0.f;
      for (int w = 0; w < KTU; ++w) dqj += dqp[w * 64 + tid];
      const float gi = cg0, gj = cg1, gf = cg2, go = cg3, cn = ccn, cp = ccp;
      const float dhn = cdh + dqj + kh * dh_state;
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
      gst(dr + j, dzi); gst(dr + A + j, dzj); gst(dr + 2 * A + j, dzf); gst(dr + 3 * A + j, dzo);
      xs_put(dzs, DZS, tid, dzi); xs_put(dzs, DZS, AU + tid, dzj);
      xs_put(dzs, DZS, 2 * AU + tid, dzf); xs_put(dzs, DZS, 3 * AU + tid, dzo);
    }
#if !defined(SATT_PF_TOP) && !defined(SATT_PF_IN_D)
    // (r5, tried and NOT kept: the cell waves issuing these in FRONT of the cell arithmetic, inside its block, so that the address
    //  arithmetic fills the cell's LDS / transcendental latencies: launch 2.86 -> 3.21 ms - the cell's result stores then wait behind
    //  the loads, and the exchange Xh behind both)
    prefetch_rows(p, max(t - 1, cb.t0), tid);
    prefetch_cell(p, max(t - 1, cb.t0), tid);
#endif
    lds_barrier();
    PROF(7); BTRACE(cb.t1 - 1 - t, 8);
    // (h) partial d[ctx|h] = dz_own x Wrec[:, own]^T: K tile = wave, every N tile; reduce over waves, publish, gather
    if (NSPLIT && t > 0) {
      // N-split: own N tiles 4*wave .. +3 over all 8 K tiles; sums leave the accumulators straight into the exchange
      static_assert(MNTB == 26, "slot split below: 24 + 2 register slots, 6 LDS slots");
      const uint16_t* zrow = dzs + min(lane & 15, 3) * DZS + (lane >> 4) * 8;
      bf16x8_t za[8];
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) za[kt] = *reinterpret_cast<const bf16x8_t*>(zrow + kt * 32);
      const i32x4_t* wl = Wl + (wave * 6) * 64 + lane;
      const i32x4_t l0 = wl[0], l1 = wl[64], l2 = wl[128], l3 = wl[192], l4 = wl[256], l5 = wl[320];
      f32x4_t q0, q1, q2, q3;
      mfma24z_a<false>(q0, q1, q2, q3, za[0], za[1], wregT[0], wregT[1], wregT[2], wregT[3], wregT[4], wregT[5], wregT[6], wregT[7]);
      mfma24_a<false>(q0, q1, q2, q3, za[2], za[3], wregT[8], wregT[9], wregT[10], wregT[11], wregT[12], wregT[13], wregT[14], wregT[15]);
      mfma24_a<false>(q0, q1, q2, q3, za[4], za[5], wregT[16], wregT[17], wregT[18], wregT[19], wregT[20], wregT[21], wregT[22], wregT[23]);
      mfma24_aav6(q0, q1, q2, q3, za[6], za[7], wregT[24], wregT[25], l0, l1, l2, l3, l4, l5);
      u64* xh = wp + WL.xh + c * KR;
      if (lane < 16) {
        const int col = wave * 64 + lane;
        gput_s(xh, (unsigned)col, tag, q0[0] + q0[1] + q0[2], same_xcd); gput_s(xh, (unsigned)(col + 16), tag, q1[0] + q1[1] + q1[2], same_xcd);
        gput_s(xh, (unsigned)(col + 32), tag, q2[0] + q2[1] + q2[2], same_xcd); gput_s(xh, (unsigned)(col + 48), tag, q3[0] + q3[1] + q3[2], same_xcd);
      }
      const int NX = NTK - 4 * AW;
      if (wave >= AW - NX) {                 // extra tile of this wave (columns 64*AW + 16*x ..): one chained accumulator
        const i32x4_t* wx = Wl + (6 * AW + (wave - (AW - NX)) * 8) * 64 + lane;
        f32x4_t qx;
        mfma41z_v<false>(qx, za[0], za[1], za[2], za[3], wx[0], wx[64], wx[128], wx[192]);
        mfma41_v(qx, za[4], za[5], za[6], za[7], wx[256], wx[320], wx[384], wx[448]);
        const int col = 64 * AW + (wave - (AW - NX)) * 16 + lane;
        if (lane < 16 && col < KR) gput_s(xh, (unsigned)col, tag, qx[0] + qx[1] + qx[2], same_xcd);
      }
      BTRACE(cb.t1 - 1 - t, 9); BTRACE(cb.t1 - 1 - t, 10);
      conv_bwd(tid, ANT);                                  // carry for a_{t-1}: first read by the next step's (b) / (c)
      // Xh: a member needs every peer's partial d ctx (CT columns) but only its OWN units of the partial d h (AU of A): 1408
      // granules instead of 2176 (3 polling loads per lane instead of 6).  The last step of a launch that hands its carried
      // gradient to another launch (chunked schedule) gathers everything.
      if (SPEC && !(cb.t0 > 0 && t == t_last)) {
        // (r5: two waves per peer - member wave >> 1, values [0, 192) and [192, CT + AU) of it: the generic mapped gather divided
        //  every granule index by CT + AU in front of the first poll, on the chain)
        static_assert(!SPEC || (SpecDimsOf<SPEC>::C == 4 && AW == 8), "two waves per member");
        const int NV = CT + AU, km = wave >> 1, r0 = (wave & 1) * 192, cnt = (wave & 1) ? NV - 192 : 192;
        const gu64* g[3]; u64 x[3]; int phys[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int r = r0 + min(lane + 64 * q, cnt - 1);
          phys[q] = km * KR + (r < CT ? r : r + c * AU);
          g[q] = (const gu64*)(wp + WL.xh + phys[q]); x[q] = 0;
        }
#ifdef SATT_EXP_XH_ONE_POLL      // (timing experiment, WRONG results: one polling load per lane instead of three - the upper bound of what wider
        //                              granules could buy this exchange)
        { const gu64* g1[1] = {g[0]}; u64 x1[1] = {0}; poll_or_die<1>(g1, tag, x1, lane, err_word, dead); x[0] = x[1] = x[2] = x1[0]; }
#else
        poll_or_die<3>(g, tag, x, lane, err_word, dead);
#endif
#pragma unroll
        for (int q = 0; q < 3; ++q) if (lane + 64 * q < cnt) cgx[phys[q]] = __uint_as_float((uint32_t)x[q]);
      } else
      gather_span(wp + WL.xh, C * KR, tag, wave, AW, lane, [&](int i, float v) { cgx[i] = v; }, err_word, dead);
      lds_barrier();
      if (tid < GL) {
        float s = dh_direct;
        if constexpr (SPEC != 0) {     // the four partials requested together (see phase (a))
          const float* cq = cgx + CT + c * AU + tid % AU;
          const float c0 = cq[0], c1 = cq[KR], c2 = cq[2 * KR], c3 = cq[3 * KR];
          s = (((s + c0) + c1) + c2) + c3;
        } else
        for (int k = 0; k < C; ++k) s += cgx[k * KR + CT + c * AU + tid % AU];
        dh_state = s;
      }
    } else if (t > 0) {
      if (wave < KTN) {
        const uint16_t* zrow = dzs + min(lane & 15, 3) * DZS + (lane >> 4) * 8 + wave * 32;
        const bf16x8_t av = *reinterpret_cast<const bf16x8_t*>(zrow);
        float* hp = hpart + wave * KRP + lane;
#pragma unroll
        for (int nt = 0; nt + 3 < MNTB; nt += 4) {
          f32x4_t q0, q1, q2, q3;
          mfma14z_a(q0, q1, q2, q3, av, wregT[nt], wregT[nt + 1], wregT[nt + 2], wregT[nt + 3]);
          if (lane < 16) {
            hp[nt * 16] = q0[0] + q0[1] + q0[2]; hp[nt * 16 + 16] = q1[0] + q1[1] + q1[2];
            hp[nt * 16 + 32] = q2[0] + q2[1] + q2[2]; hp[nt * 16 + 48] = q3[0] + q3[1] + q3[2];
          }
        }
        static_assert(MNTB % 4 == 2, "tail block below handles exactly two tiles");
        {
          f32x4_t q0, q1;
          mfma12z_a(q0, q1, av, wregT[MNTB - 2], wregT[MNTB - 1]);
          if (lane < 16) { hp[(MNTB - 2) * 16] = q0[0] + q0[1] + q0[2]; hp[(MNTB - 1) * 16] = q1[0] + q1[1] + q1[2]; }
        }
        for (int nl = 0; nl < NTL; nl += 4) {
          f32x4_t q0, q1, q2, q3;
          const i32x4_t* w0 = Wl + (wave * NTL + nl) * 64 + lane;
          mfma14z_v(q0, q1, q2, q3, av, w0[0], w0[64], w0[128], w0[192]);
          if (lane < 16) {
            float* h2 = hp + (MNTB + nl) * 16;
            h2[0] = q0[0] + q0[1] + q0[2]; h2[16] = q1[0] + q1[1] + q1[2];
            h2[32] = q2[0] + q2[1] + q2[2]; h2[48] = q3[0] + q3[1] + q3[2];
          }
        }
      }
      lds_barrier();
      BTRACE(cb.t1 - 1 - t, 9);
      for (int i = tid; i < KR; i += ANT) {
        float s = 0.f;
        for (int w = 0; w < KTN; ++w) s += hpart[w * KRP + i];
        gput(wp + WL.xh + c * KR + i, tag, s, same_xcd);
      }
      BTRACE(cb.t1 - 1 - t, 10);
      conv_bwd(tid, ANT);                                  // carry for a_{t-1}: first read by the next step's (b) / (c)
      gather_span(wp + WL.xh, C * KR, tag, wave, AW, lane, [&](int i, float v) { cgx[i] = v; }, err_word, dead);
      lds_barrier();
      if (tid < GL) {
        float s = dh_direct;
        for (int k = 0; k < C; ++k) s += cgx[k * KR + CT + c * AU + tid % AU];
        dh_state = s;
      }
    }
    PROF(8); BTRACE(cb.t1 - 1 - t, 11);
    if (t == next_lo) {                  // chunk finished: its per-step gradients are complete -> visible, then count
      chunk_release();
      __syncthreads();
      if (threadIdx.x == 0) atomicAdd(cb.done + bidx, 1u);       // one word per chunk: samples run at different speeds
      ++bidx;
      next_lo = bidx < cb.nbound ? cb.bound[bidx] : -1;
    }
  }
  if (cb.t0 > 0) {   // hand the carried gradients to the next (earlier) chunk
    const int tid = threadIdx.x;
    if (c == 0) {
      for (int i = tid; i < KR; i += ANT) { float s = 0.f; for (int k = 0; k < C; ++k) s += cgx[k * KR + i]; stb[i] = s; }
      for (int i = tid; i < Ti; i += ANT) { stb[C * NWP + 2 * A + i] = dac[i] + dac[T4 + i] + dac[2 * T4 + i]; stb[C * NWP + 2 * A + Ti + i] = dalc[i]; }
      if (tid == 0) stb[C * NWP + 2 * A + 2 * Ti] = du_s[0];
    }
    if (tid < AU) { stb[C * NWP + c * AU + tid] = dc_state; stb[C * NWP + A + c * AU + tid] = dh_state; }
  }
  PROF_STORE(16);
}

__global__ void attn_cluster_pack_k(const float* __restrict__ W, int64_t ld, uint16_t* __restrict__ WP,
                                    uint16_t* __restrict__ WTP, int K, int A, int C) {
  const int AU = A / C, NL = 4 * AU, KT = kt_of(K), MNTW = mntw_of(NL), NTK = (K + 15) / 16;
  const bool nsplit = nsplit_of(K, A, C);
  const int NSRC = nsplit ? NS_SLOTS : NTK, NX = NTK - 4 * AW;
  const int64_t n1 = (int64_t)C * AW * MNTW * KT * 512, n2 = (int64_t)C * AW * NSRC * 512;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n1 + n2; e += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(e & 7), l = (int)((e >> 3) & 63);
    if (e < n1) {          // forward slice in MFMA B-operand order: [C][AW][MNTW][KT][64 lanes][8]
      int64_t r = e >> 9;
      const int kt = (int)(r % KT); r /= KT;
      const int j = (int)(r % MNTW); r /= MNTW;
      const int wv = (int)(r % AW), c = (int)(r / AW);
      const int n = (wv * MNTW + j) * 16 + (l & 15), k = kt * 32 + (l >> 4) * 8 + i;
      uint16_t v = 0;
      if (n < NL && k < K) { const int g = n / AU, u = n - g * AU; v = f2bf(W[(int64_t)k * ld + g * A + c * AU + u]); }
      WP[e] = v;
    } else {               // backward slice (transposed): [C][AW = K tile over own gate columns][NTK][64 lanes][8]
      int64_t r = (e - n1) >> 9;
      int nt = (int)(r % NSRC); r /= NSRC;
      const int wv = (int)(r % AW), c = (int)(r / AW);
      int ktile = wv;                                    // K-split layout: the wave is the K tile
      if (nsplit) {                                      // N-split layout: slot -> (K tile, N tile) of wave wv
        const int slot = nt;
        if (slot < 32) { ktile = slot >> 2; nt = 4 * wv + (slot & 3); }
        else { ktile = slot - 32; nt = wv >= AW - NX ? 4 * AW + wv - (AW - NX) : -1; }
      }
      const int k = ktile * 32 + (l >> 4) * 8 + i;       // own gate column, local order g*AU + u
      const int n = nt * 16 + (l & 15);                  // input row of Wrec
      uint16_t v = 0;
      if (nt >= 0 && k < NL && n < K) { const int g = k / AU, u = k - g * AU; v = f2bf(W[(int64_t)n * ld + g * A + c * AU + u]); }
      WTP[e - n1] = v;
    }
  }
}

inline int ccheck(const satt_attn_rnn_params& p, int C);
// dynamic LDS of a launch: the folded forward / saved-factor backward kernels use the fixed layout of Ti = 32 FKT
inline size_t fwd_smem(const satt_attn_rnn_params& p, int C, bool klds, bool fold) {
  const int Ti = fold ? 32 * FKT : p.Ti;
  return sizeof(float) * carve_cf(p.A, p.V1 + p.V2, p.U1 + p.U2, Ti, 5, p.kernel, 4 * (p.A / C), (Ti + C - 1) / C, klds, fold ? p.V1 : 0).total;
}
inline size_t bwd_smem(const satt_attn_rnn_params& p, int C, bool klds, bool saf) {
  const int Ti = saf ? 32 * FKT : p.Ti;
  return sizeof(float) * carve_cb(p.A, p.V1 + p.V2, p.U1 + p.U2, Ti, 5, p.kernel, C, (Ti + C - 1) / C, klds, saf).total;
}
inline bool fold_ok(const satt_attn_rnn_params& p, int C) {
  return spec_dims(p, C) != 0 && p.keys_lds_bf16 != 0 && p.Ti <= 32 * FKT && p.teach1 == nullptr && p.teach2 == nullptr;
}
inline int ccheck(const satt_attn_rnn_params& p, int C) {
  if (p.B <= 0 || p.Td <= 0 || p.Ti <= 0 || C < 2 || C > 8) return SATT_E_BADARG;
  if (p.att1_mode < 0 || p.att1_mode > 1 || (p.cumulative && !p.acum)) return SATT_E_BADARG;
  if (p.agentW && (!p.agentb || !p.ustate)) return SATT_E_BADARG;
  if (p.filters != 5) return SATT_E_UNSUPPORTED;
  if (p.U1 > 64 * NQ || p.V1 > 64 * NQ || p.U2 > 64 || p.V2 > 64 || p.U1 % 4 || p.V1 % 4) return SATT_E_UNSUPPORTED;
  if ((p.U1 + p.U2) % 8 || (p.V1 + p.V2 + p.A) % 8 || p.A % 8) return SATT_E_UNSUPPORTED;
  if (p.A % C || (p.A / C) % 8 || p.A > ANT || 4 * p.A > 8 * ANT) return SATT_E_UNSUPPORTED;
  if (p.U1 + p.U2 > ANT || p.V1 + p.V2 > ANT) return SATT_E_UNSUPPORTED;
  if (nwp_of(p.V1 + p.V2 + p.A, C) > ANT) return SATT_E_UNSUPPORTED;
  if (p.B * C > 1024) return SATT_E_UNSUPPORTED;   // (host-only sanity bound; the launchers compare with the device's resident capacity)
  // forward slice: MNTW tiles of 16 gate columns per wave (K tiles beyond the register budget go to LDS)
  const int mntw = mntw_of(4 * (p.A / C));
  if (mntw > 2 || p.A / C > 64 || p.U1 + p.U2 > 16 * MNTQ * AW) return SATT_E_UNSUPPORTED;
  if (p.V1 % 16) return SATT_E_UNSUPPORTED;        // context tiles must not straddle the two value sources
  if (p.Ti > 64 * GQ || p.A > 64 * GQ || C * (p.V1 + p.V2 + NSC) > 64 * GQ * (AW - 3) || C * (p.U1 + p.U2) > 64 * GQ * AW ||
      C * (p.V1 + p.V2 + p.A) > 64 * GQ * AW || C * (p.U1 + p.U2) + p.Ti * (p.filters + 1) > 64 * GQ * AW)
    return SATT_E_UNSUPPORTED;                     // single-pass gathers (gather_span)
  return SATT_OK;
}

// which instantiation a launch runs (one place for the launchers and the residency queries)
inline const void* fwd_kernel(bool fold, bool klds, int spec, int mntw) {
  if (fold) return spec == 1 ? (const void*)attn_cluster_fwd_k<5, true, 2, 1, true> : (const void*)attn_cluster_fwd_k<5, true, 2, 2, true>;
  if (klds) {
    if (spec == 1) return (const void*)attn_cluster_fwd_k<5, true, 2, 1>;
    if (spec == 2) return (const void*)attn_cluster_fwd_k<5, true, 2, 2>;
    return mntw == 1 ? (const void*)attn_cluster_fwd_k<5, true, 1, 0> : (const void*)attn_cluster_fwd_k<5, true, 2, 0>;
  }
  if (spec == 1) return (const void*)attn_cluster_fwd_k<5, false, 2, 1>;
  if (spec == 2) return (const void*)attn_cluster_fwd_k<5, false, 2, 2>;
  return mntw == 1 ? (const void*)attn_cluster_fwd_k<5, false, 1, 0> : (const void*)attn_cluster_fwd_k<5, false, 2, 0>;
}
inline const void* bwd_kernel(bool saf, bool klds, int spec, bool nsp) {
  if (saf) return spec == 1 ? (const void*)attn_cluster_bwd_k<5, true, 1, true, true> : (const void*)attn_cluster_bwd_k<5, true, 2, true, true>;
  if (klds) {
    if (spec == 1) return (const void*)attn_cluster_bwd_k<5, true, 1, true>;
    if (spec == 2) return (const void*)attn_cluster_bwd_k<5, true, 2, true>;
    return nsp ? (const void*)attn_cluster_bwd_k<5, true, 0, true> : (const void*)attn_cluster_bwd_k<5, true, 0, false>;
  }
  if (spec == 1) return (const void*)attn_cluster_bwd_k<5, false, 1, true>;
  if (spec == 2) return (const void*)attn_cluster_bwd_k<5, false, 2, true>;
  return nsp ? (const void*)attn_cluster_bwd_k<5, false, 0, true> : (const void*)attn_cluster_bwd_k<5, false, 0, false>;
}
inline bool bwd_uses_saf(const satt_attn_rnn_params& p, int C) {
  static const bool bwd_nosaf = getenv("SATT_BWD_NOSAF") != nullptr;      // diagnosis switch
  return !bwd_nosaf && spec_dims(p, C) != 0 && p.keys_lds_bf16 != 0 && p.saf != nullptr && (p.Ti + C - 1) / C <= (2 * RBB - 1) * AW;
}

}  // namespace

extern "C" int64_t satt_attn_cluster_ws_bytes(const satt_attn_rnn_params* f, int C) {
  if (!f) return 0;
  return (int64_t)sizeof(u64) * ws_tail_words(f->B, f->A, f->Ti, C, f->U1 + f->U2, 5, f->V1 + f->V2, spec_dims(*f, C) != 0) + 64;
}
extern "C" int64_t satt_attn_cluster_state_floats(const satt_attn_rnn_params* f, int C) {
  if (!f) return 0;
  return (int64_t)f->B * (C * nwp_of(f->V1 + f->V2 + f->A, C) + 2 * f->A + 2 * f->Ti + 4);
}
extern "C" int64_t satt_attn_cluster_pack_elems(int K, int A, int C, int transposed) {
  if (transposed) return (int64_t)C * AW * (nsplit_of(K, A, C) ? NS_SLOTS : (K + 15) / 16) * 512;
  return (int64_t)C * AW * mntw_of(4 * (A / C)) * kt_of(K) * 512;
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
  if (cp->progress && (cp->nbound < 0 || cp->nbound > SATT_MAX_BOUNDS)) return SATT_E_BADARG;
  const int C = cp->C, CT = p.V1 + p.V2, UQ = p.U1 + p.U2, NL = 4 * (p.A / C), nown = (p.Ti + C - 1) / C;
  const bool klds = p.keys_lds_bf16 != 0;
  const bool fold = cp->vw1 != nullptr;
  if (fold && !fold_ok(p, C)) return SATT_E_BADARG;      // the caller asks satt_attn_cluster_fold first
  const size_t smem = fwd_smem(p, C, klds, fold);
  if (smem > 160 * 1024) return SATT_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(cp->ws, 0, (size_t)satt_attn_cluster_ws_bytes(&p, C) - 64, s) != hipSuccess) return SATT_E_LAUNCH;   // not the sticky tail
  const int mntw = mntw_of(NL);
  satt_attn_cluster_params cq = *cp;
  single_source_fixup(cq.f);
  const void* fn = fwd_kernel(fold, klds, spec_dims(p, C), mntw);
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const int cap = cluster_capacity(fn, ANT, smem);
  if (cap >= 0 && p.B * C > cap) return SATT_E_UNSUPPORTED;        // not every member could be resident: peers would spin for them
  void* args[] = {&cq};
  if (hipLaunchKernel(fn, dim3(p.B, C), dim3(ANT), args, smem, s) != hipSuccess) return SATT_E_LAUNCH;
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

/* 1 if satt_attn_cluster_fwd runs the FOLDED form for this problem when given vw1 (the specialised bf16 kernel, Ti <= 32 * FKT,
 * both sources, plain forward / location-sensitive attention); the caller then passes vw1 = values1 x Wrec[ctx1 rows] ([B*Ti, 4A]),
 * a forward pack made from the rows [ctx2 | h] of Wrec only (satt_attn_cluster_pack with K = V2 + A), and fills the ctx1 columns
 * of `out` itself (alpha x values1) - the kernel writes h and ctx2 only. */
extern "C" int satt_attn_cluster_fold(const satt_attn_rnn_params* f, int C) {
  if (!f || ccheck(*f, C)) return 0;
  if (!fold_ok(*f, C)) return 0;
  const int CT = f->V1 + f->V2, UQ = f->U1 + f->U2, NL = 4 * (f->A / C), nown = (f->Ti + C - 1) / C;
  return fwd_smem(*f, C, true, true) <= 160 * 1024;
}

/* SATT_OK if the cluster kernels support this problem with C members per sample (sizes, LDS, residency) */
extern "C" int satt_attn_cluster_check(const satt_attn_rnn_params* f, int C) {
  if (!f) return SATT_E_BADARG;
  int rc = ccheck(*f, C);
  if (rc) return rc;
  const int CT = f->V1 + f->V2, UQ = f->U1 + f->U2, NL = 4 * (f->A / C), nown = (f->Ti + C - 1) / C;
  const bool klds = f->keys_lds_bf16 != 0;
  if (sizeof(float) * carve_cf(f->A, CT, UQ, f->Ti, 5, f->kernel, NL, nown, klds).total > 160 * 1024) return SATT_E_UNSUPPORTED;
  if (sizeof(float) * carve_cb(f->A, CT, UQ, f->Ti, 5, f->kernel, C, nown, klds).total > 160 * 1024) return SATT_E_UNSUPPORTED;
  if (spec_dims(*f, C) != 0 && klds && nown <= (2 * RBB - 1) * AW &&        // the saved-factor kernel's layout (value-row images)
      bwd_smem(*f, C, klds, true) > 160 * 1024) return SATT_E_UNSUPPORTED;
  return SATT_OK;
}

extern "C" int satt_attn_cluster_bwd(const satt_attn_cluster_bwd_params* cb, void* stream) {
  if (!cb) return SATT_E_BADARG;
  const satt_attn_rnn_params& p = cb->b.f;
  if (p.teach1 || p.teach2) return SATT_E_UNSUPPORTED;   // forced alignments are an inference-time mode
  if (p.agentW && !cb->b.dz) return SATT_E_BADARG;
  if (cb->ready && (!cb->done || cb->nbound < 1 || cb->nbound > SATT_MAX_BOUNDS || cb->bound[cb->nbound - 1] != cb->t0))
    return SATT_E_BADARG;
  int rc = ccheck(p, cb->C);
  if (rc) return rc;
  if (cb->t0 < 0 || cb->t1 > p.Td || cb->t0 >= cb->t1 || ((cb->t0 > 0 || cb->t1 < p.Td) && !cb->state)) return SATT_E_BADARG;
  const int C = cb->C, CT = p.V1 + p.V2, UQ = p.U1 + p.U2, nown = (p.Ti + C - 1) / C;
  const bool klds = p.keys_lds_bf16 != 0;
  const int spec = spec_dims(p, C);        // != 0 implies the N-split layout of the packed backward slice
  // saved derivative factors (written by the folded forward launch of the same step): two passes of RBB and RBB - 1 own rows
  const bool saf = bwd_uses_saf(p, C);
  const size_t smem = bwd_smem(p, C, klds, saf);
  if (smem > 160 * 1024) return SATT_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(cb->ws, 0, (size_t)satt_attn_cluster_ws_bytes(&p, C) - 64, s) != hipSuccess) return SATT_E_LAUNCH;   // not the sticky tail
  satt_attn_cluster_bwd_params cq = *cb;
  single_source_fixup(cq.b.f);
  const void* fn = bwd_kernel(saf, klds, spec, nsplit_of(p.V1 + p.V2 + p.A, p.A, C));
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const int cap = cluster_capacity(fn, ANT, smem);
  if (cap >= 0 && p.B * C > cap) return SATT_E_UNSUPPORTED;
  void* args[] = {&cq};
  if (hipLaunchKernel(fn, dim3(p.B, C), dim3(ANT), args, smem, s) != hipSuccess) return SATT_E_LAUNCH;
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

/* Resident footprint of the launch satt_attn_cluster_fwd / _bwd would make for these parameters (needs a device): *workgroups = B*C,
 * *per_cu = workgroups of this kernel one CU can hold (occupancy calculator: registers, LDS), *cus = CUs of the current device.
 * The launchers refuse workgroups > per_cu * cus; a caller that keeps SEVERAL cluster launches in flight (the layer pipeline) must
 * keep the sum of their footprints within the device - a workgroup that spins for a peer which cannot become resident times out. */
extern "C" int satt_attn_cluster_residency(const satt_attn_cluster_params* cp, int* workgroups, int* per_cu, int* cus) {
  if (!cp || !workgroups || !per_cu || !cus) return SATT_E_BADARG;
  const satt_attn_rnn_params& p = cp->f;
  int rc = ccheck(p, cp->C);
  if (rc) return rc;
  const int C = cp->C, CT = p.V1 + p.V2, UQ = p.U1 + p.U2, NL = 4 * (p.A / C), nown = (p.Ti + C - 1) / C;
  const bool klds = p.keys_lds_bf16 != 0, fold = cp->vw1 != nullptr;
  if (fold && !fold_ok(p, C)) return SATT_E_BADARG;
  const size_t smem = fwd_smem(p, C, klds, fold);
  const void* fn = fwd_kernel(fold, klds, spec_dims(p, C), mntw_of(NL));
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (cluster_capacity(fn, ANT, smem, per_cu, cus) < 0) return SATT_E_LAUNCH;
  *workgroups = p.B * C;
  return SATT_OK;
}
extern "C" int satt_attn_cluster_bwd_residency(const satt_attn_cluster_bwd_params* cb, int* workgroups, int* per_cu, int* cus) {
  if (!cb || !workgroups || !per_cu || !cus) return SATT_E_BADARG;
  const satt_attn_rnn_params& p = cb->b.f;
  int rc = ccheck(p, cb->C);
  if (rc) return rc;
  const int C = cb->C, CT = p.V1 + p.V2, UQ = p.U1 + p.U2, nown = (p.Ti + C - 1) / C;
  const bool klds = p.keys_lds_bf16 != 0, saf = bwd_uses_saf(p, C);
  const size_t smem = bwd_smem(p, C, klds, saf);
  const void* fn = bwd_kernel(saf, klds, spec_dims(p, C), nsplit_of(p.V1 + p.V2 + p.A, p.A, C));
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (cluster_capacity(fn, ANT, smem, per_cu, cus) < 0) return SATT_E_LAUNCH;
  *workgroups = p.B * C;
  return SATT_OK;
}

/* host-synchronous: non-zero if a hand-off of ANY launch on `ws` timed out since the caller zeroed the workspace (the error
 * word lives in the 64-byte tail, which launches never clear; satt_adam_step reads the same word on the device) */
extern "C" int satt_attn_cluster_status(const satt_attn_rnn_params* f, int C, const void* ws, void* stream) {
  unsigned int v = 0;
  const char* pz = (const char*)ws + satt_attn_cluster_ws_bytes(f, C) - 64;
  if (hipMemcpyAsync(&v, pz, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return SATT_E_LAUNCH;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return SATT_E_LAUNCH;
  return v ? SATT_E_LAUNCH : SATT_OK;
}

/* host-synchronous (tests): *count = workgroup-launches on `ws` (since the caller zeroed it) whose start-up handshake found
 * every member of their cluster on one XCD and that therefore exchanged with plain stores (cluster_xchg.h): a multiple of
 * B*C; *slow (optional) = workgroup-launches that took the write-through path instead */
extern "C" int satt_attn_cluster_fastpath(const satt_attn_rnn_params* f, int C, const void* ws, void* stream, int* count,
                                          int* slow) {
  if (!f || !ws || !count) return SATT_E_BADARG;
  unsigned int v[2] = {0, 0};
  const char* pz = (const char*)ws + satt_attn_cluster_ws_bytes(f, C) - 64 + 4;
  if (hipMemcpyAsync(v, pz, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return SATT_E_LAUNCH;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return SATT_E_LAUNCH;
  *count = (int)v[0];
  if (slow) *slow = (int)v[1];
  return SATT_OK;
}

#ifdef SATT_PROFILE
extern "C" int satt_prof_read_prolog(unsigned long long* host8) {
  return hipMemcpyFromSymbol(host8, HIP_SYMBOL(satt_prolog), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -3;
}
extern "C" int satt_prof_read_trace(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(satt_prof_trace), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -3;
}
extern "C" int satt_prof_read_cluster(unsigned long long* host32) {
  return hipMemcpyFromSymbol(host32, HIP_SYMBOL(satt_prof_acc), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -3;
}
#endif
