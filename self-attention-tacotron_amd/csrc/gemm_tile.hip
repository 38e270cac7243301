// Large-tile bf16 MFMA GEMMs for gfx950 (the benchmark precision).  satt_gemm (gemm.hip) routes a problem here when
// it fits one of the two families; everything else - and the exact-fp32 parity mode - stays on the generic kernel.
//
//   gemm_rk_k  "reduction-contiguous": C = epilogue(A * B) with A fp32 [m][k] rows (plain, or the virtual im2col of a
//              SAME Conv1D: rows shifted by the tap) and B a bf16 SHADOW of the weights laid out k-contiguous per output
//              column (satt_shadow_pack).  Forward Dense / Conv1D / conv bank and their input gradients.
//   gemm_dw_k  weight gradients: C[i][j] += sum_m A(m, i) * B(m, j), both operands fp32 rows of activations / gradients
//              (reduction along the STRIDED dimension), transposed on their way into LDS; optional fused column sum of
//              B (the bias gradient, formerly a separate colsum launch); conv-bank form: the 16 weight gradients of the
//              bank in ONE launch.
// Split reductions (split-K, the bank's input gradient) do NOT use atomics when the caller passes a workspace: fp32
// atomic adds retire at ~50 G lanes/s on this chip (the split weight gradients of round 1 were bound by them); every
// split writes its partial tile to its own slab with plain stores and slab_reduce_k sums the slabs into C
// (deterministic, and C needs no zeroing first).
//
// Common structure: BM x BN output tile per 256-thread workgroup (2 x 2 waves, each wave (BM/32) x (BN/32) tiles of
// v_mfma_f32_16x16x32_bf16), BK = 32 per stage, DOUBLE-BUFFERED LDS: global loads of stage s+2 are issued right after
// the LDS writes of stage s+1 and stay in flight across the barrier (lds_barrier() waits for LDS traffic only), so a
// stage costs one barrier and the load round trip hides behind a whole stage of MFMAs.
// LDS image of an operand tile: [k/8][row][8 bf16] - a lane's MFMA fragment (row l&15, k chunk l>>4) is ONE 16-byte
// read, and the 16 lanes of a ds_read_b128 service group always touch 16 different rows of the same or the neighbouring
// chunk.  Physical row = row ^ swz(chunk) with swz = 2*(chunk&3) ^ (chunk>>2): the two row quads a group takes from the
// neighbouring chunk stay disjoint from the other two (conflict-free reads), and the 8 lanes of a ds_write_b128 group
// (2 rows x 4 chunks) land on 8 different 16-byte slots (conflict-free writes).
// blockIdx -> tile: XCD-aware (block b runs on XCD b % 8): every XCD gets a contiguous range of tiles, N tiles fastest,
// so the workgroups that share an A row panel share one L2 instead of fetching it into all eight.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include "common.h"
#include "gemm_tile.h"

namespace {

constexpr int TNT = 256;
constexpr int TBK = 32;
constexpr int DW_PREFETCH_DEFAULT = 2;      // the same for gemm_dw_k (SATT_DW_PREFETCH = 1 .. 3; profiles/r06_rk_prefetch.txt)
constexpr int RK_PREFETCH_DEFAULT = 2;      // register stages of gemm_rk_k's operand prefetch (SATT_RK_PREFETCH = 1 .. 4 overrides; r6 sweep: profiles/r06_rk_prefetch.txt)

__device__ __forceinline__ int swz(int kq) { return (2 * (kq & 3)) ^ (kq >> 2); }

__device__ __forceinline__ int xcd_remap(int bid, int n) {
  const int q = n >> 3, r = n & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// ------------------------------------------------------------------------------------------------ gemm_rk_k
// PD = register stages of operand prefetch (r6).  PD = 1 is the r2 scheme: the loads of K step kt + 1 are in flight during the 8 - 16
// MFMAs of step kt - ~50 - 100 ns of matrix work against a ~500 ns L2 round trip, hidden only by the other workgroups of the CU.  A
// stage is 16 - 24 registers (one or two 32-byte A segments + two 16-byte B segments per thread), so PD stages of them cost little:
// stage s lives in register set s % PD, the K loop is unrolled by PD so that every set index is a compile-time constant.
template <int D, class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int D, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<D>(f, std::make_integer_sequence<int, D>{}); }

template <int BM, int BN, bool CONV, int PD>
__global__ __launch_bounds__(TNT) void gemm_rk_k(const satt_gemm_params p, const int ntn, const int ntiles) {
  constexpr int BK = TBK, KQ = BK / 8;
  constexpr int SA = BM * KQ / TNT, SB = BN * KQ / TNT;     // 16-byte (8 x bf16) segments per thread and stage
  constexpr int TM = BM / 32, TN = BN / 32;
  constexpr int STG = (BM + BN) * BK;                        // bf16 elements per stage
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * STG];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lin = xcd_remap((int)blockIdx.x, ntiles);
  const int tm = lin / ntn, tn = lin - tm * ntn;
  const int m0 = tm * BM, n0 = tn * BN;

  int z = blockIdx.z;
  const bool bank = CONV && p.bank_ng > 0;
  const bool bank_sum = bank && p.bank_c_col == 0;      // every group adds into the same C: groups are paired per z
  const int zk = bank ? 0 : z % p.splitk;
  if (!bank) z /= p.splitk;
  const int zo = bank ? 0 : z / p.nb_inner, zi = bank ? 0 : z - zo * p.nb_inner;
  float* __restrict__ C = p.C + zo * p.strideC_o + zi * p.strideC_i;
  // slab output (split reductions with a workspace): dense [M][N] partial tile sums, one slab per blockIdx.z
  const bool slab_out = p.ws != nullptr && (p.splitk > 1 || bank_sum);
  bool atomic_out = !slab_out && (p.splitk > 1 || bank_sum);

  // ---- staging maps (fixed per thread): segment e = tid + 256 g -> (row e / KQ, chunk e % KQ)
  int64_t arel[SA]; int at[SA]; bool aok[SA]; int aoff[SA];
#pragma unroll
  for (int g = 0; g < SA; ++g) {
    const int e = tid + TNT * g, row = e / KQ, sq = e % KQ;
    const int m = m0 + row;
    aok[g] = m < p.M;
    arel[g] = (int64_t)m * p.lda + sq * 8;
    at[g] = CONV ? m % p.conv_T : 0;
    aoff[g] = (sq * BM + (row ^ swz(sq))) * 8;
  }
  int64_t brel[SB]; bool bok[SB]; int boff[SB];
#pragma unroll
  for (int g = 0; g < SB; ++g) {
    const int e = tid + TNT * g, n = e / KQ, sq = e % KQ;
    bok[g] = n0 + n < p.N;
    brel[g] = (int64_t)(n0 + n) * p.sbs_n + sq * 8;
    boff[g] = BM * BK + (sq * BN + (n ^ swz(sq))) * 8;
  }

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int kq = lane >> 4, pr = (lane & 15) ^ swz(kq);        // chunk / physical row inside a 16-row fragment

  // one pass = one K loop; the summed bank (input gradient of the conv bank) runs two passes per workgroup - groups z and
  // ng-1-z, whose reduction lengths add up to the same total for every z - into the same accumulators
  const int npass = bank_sum ? ((int)blockIdx.z * 2 + 1 == p.bank_ng ? 1 : 2) : 1;
  for (int pass = 0; pass < npass; ++pass) {
  const float* __restrict__ A = p.A + zo * p.strideA_o + zi * p.strideA_i;
  const uint16_t* __restrict__ Bs = p.Bs;
  int Kz = p.K, conv_off = p.conv_off;
  if (bank) {                     // conv bank: blockIdx.z is the group, widest (longest K) first
    int g = p.bank_ng - 1 - (int)blockIdx.z;
    if (bank_sum && pass == 1) g = (int)blockIdx.z;
    Kz = (g + 1) * p.conv_C;
    conv_off = -p.conv_sgn * (g / 2);
    A = p.A + (int64_t)g * p.bank_a_col;
    Bs = p.Bs + p.bank_b_unit * (int64_t)(g * (g + 1) / 2);
    if (!bank_sum) C = p.C + (int64_t)g * p.bank_c_col;
  }
  int kbeg = 0, kend = Kz;
  if (!bank && p.splitk > 1) {
    int chunk = (p.K + p.splitk - 1) / p.splitk;
    chunk = (chunk + BK - 1) / BK * BK;
    kbeg = zk * chunk;
    kend = min(p.K, kbeg + chunk);
  }
  const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  // K-step bookkeeping (uniform): (atap, ac0) tap / first channel of the A tile, (btap, br0) of the B tile
  int atap = 0, ac0 = kbeg, btap = 0, br0 = kbeg;
  if (CONV) {
    atap = kbeg / p.conv_C; ac0 = kbeg - atap * p.conv_C;
    btap = kbeg / p.kin; br0 = kbeg - btap * p.kin;
  }
  // Register stage of the NEXT K step: the loads are issued at clamped (always valid) addresses and stay RAW until the stage is
  // written to LDS one K step later - masking them right behind the load (v = ok ? v : 0) makes the compiler wait for the data
  // at the load, i.e. no prefetch at all and one exposed memory latency per K step.
  float4 ra[PD][SA][2]; u32x4_t rb[PD][SB];
  bool rao[PD][SA], rbo[PD][SB];
  int kload = kbeg;
  auto gload = [&](auto SC) {                       // called once per K step, in order; SC: the register set (compile time)
    constexpr int S = decltype(SC)::value;
    const int shift = CONV ? p.conv_sgn * atap + conv_off : 0;
    const int64_t ao = CONV ? (int64_t)shift * p.lda + ac0 : (int64_t)ac0;
    const int64_t bo = CONV ? (int64_t)btap * p.sbs_tap + br0 : (int64_t)br0;
#pragma unroll
    for (int g = 0; g < SA; ++g) {
      const int sq = (tid + TNT * g) % KQ;
      bool ok = aok[g] && kload + sq * 8 < kend;
      if (CONV) ok = ok && (unsigned)(at[g] + shift) < (unsigned)p.conv_T;
      const float4* src = reinterpret_cast<const float4*>(ok ? A + arel[g] + ao : A);
      ra[S][g][0] = src[0]; ra[S][g][1] = src[ok ? 1 : 0];
      rao[S][g] = ok;
    }
#pragma unroll
    for (int g = 0; g < SB; ++g) {
      const int sq = (tid + TNT * g) % KQ;
      const bool ok = bok[g] && kload + sq * 8 < kend;
      rb[S][g] = *reinterpret_cast<const u32x4_t*>(ok ? Bs + brel[g] + bo : Bs);
      rbo[S][g] = ok;
    }
    kload += BK;
    if (CONV) {
      ac0 += BK; if (ac0 >= p.conv_C) { ac0 = 0; ++atap; }
      br0 += BK; if (br0 >= p.kin) { br0 = 0; ++btap; }
    } else {
      ac0 += BK; br0 += BK;
    }
  };
  auto swrite = [&](auto SC, int buf) {
    constexpr int S = decltype(SC)::value;
    uint16_t* base = lds + buf * STG;
#pragma unroll
    for (int g = 0; g < SA; ++g) {
      u32x4_t w;
      w[0] = pack_bf16x2(ra[S][g][0].x, ra[S][g][0].y); w[1] = pack_bf16x2(ra[S][g][0].z, ra[S][g][0].w);
      w[2] = pack_bf16x2(ra[S][g][1].x, ra[S][g][1].y); w[3] = pack_bf16x2(ra[S][g][1].z, ra[S][g][1].w);
      if (!rao[S][g]) w = (u32x4_t){0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4_t*>(base + aoff[g]) = w;
    }
#pragma unroll
    for (int g = 0; g < SB; ++g) *reinterpret_cast<u32x4_t*>(base + boff[g]) = rbo[S][g] ? rb[S][g] : (u32x4_t){0u, 0u, 0u, 0u};
  };

  if (nk > 0) {
    gload(std::integral_constant<int, 0>{}); swrite(std::integral_constant<int, 0>{}, 0);
    lds_barrier();
    // stages 1 .. PD into sets 1 .. PD - 1, 0 (set 0 is free again)
    static_for<PD>([&](auto J) { constexpr int d = decltype(J)::value + 1; if (nk > d) gload(std::integral_constant<int, d % PD>{}); });
  }
  for (int kt0 = 0; kt0 < nk; kt0 += PD)
  static_for<PD>([&](auto J) {
    const int kt = kt0 + decltype(J)::value;
    if (kt >= nk) return;
    typedef std::integral_constant<int, (decltype(J)::value + 1) % PD> SN;      // the set that holds stage kt + 1
    const uint16_t* As = lds + (kt & 1) * STG;
    const uint16_t* Bt = As + BM * BK;
    bf16x8_t a[TM], b[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
      a[i] = *reinterpret_cast<const bf16x8_t*>(As + (kq * BM + wm * (BM / 2) + i * 16 + pr) * 8);
#pragma unroll
    for (int j = 0; j < TN; ++j)
      b[j] = *reinterpret_cast<const bf16x8_t*>(Bt + (kq * BN + wn * (BN / 2) + j * 16 + pr) * 8);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    if (kt + 1 < nk) {
      swrite(SN{}, (kt + 1) & 1);
      if (kt + 1 + PD < nk) gload(SN{});
    }
    lds_barrier();
  });
  }   // pass

  const uint32_t seed = (p.drop_thresh != 0 && p.seed) ? *p.seed : 0u;
  float* __restrict__ slab = slab_out ? p.ws + (int64_t)blockIdx.z * p.M * p.N : nullptr;
  // Epilogue operands FIRST (bias per column, residual / previous C per element; clamped addresses, uniform branches), then
  // arithmetic and stores without a single wait: a load next to each store (`if (p.bias) v += p.bias[col]` per element)
  // compiles to load - s_waitcnt vmcnt(0) - store per element, and on gfx9 vmcnt also counts the STORES in flight - every
  // element then waits for the previous element's write acknowledgement plus its own load: 16..32 serial round trips per thread.
  const bool plain = !slab_out && !atomic_out;
  float bcol[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bcol[j] = 0.f;
  if (plain && p.bias) {
#pragma unroll
    for (int j = 0; j < TN; ++j) bcol[j] = p.bias[min(n0 + wn * (BN / 2) + j * 16 + (lane & 15), p.N - 1)];
  }
  // one loop per output mode; plain mode: every value final in registers (all loads consumed) before the first store
  if (slab_out || atomic_out) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm * (BM / 2) + i * 16 + (lane >> 4) * 4 + r;
          const int col = n0 + wn * (BN / 2) + j * 16 + (lane & 15);
          if (row < p.M && col < p.N) {
            const float v = p.alpha * acc[i][j][r];
            if (slab_out) slab[(int64_t)row * p.N + col] = v;
            else atomicAdd(C + (int64_t)row * p.ldc + col, v);
          }
        }
    return;
  }
  // Interior tiles (the usual case): per 16-row fragment i, residual / previous C of its TN x 4 elements first (two batches),
  // then the values, then the stores.  Four row pointers per fragment and constant column offsets (16 j floats) keep the
  // address registers at 8 - with per-element clamped addresses the prefetch doubled the kernel's register count and halved the
  // occupancy of the compute-bound shapes.  Edge tiles take the element-by-element form.
  const bool interior = m0 + BM <= p.M && n0 + BN <= p.N;
  const int rbase = m0 + wm * (BM / 2) + (lane >> 4) * 4, cbase = n0 + wn * (BN / 2) + (lane & 15);
  if (interior) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float ex[TN][4];
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) ex[j][r] = 0.f;
      if (p.residual) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* rp = p.residual + (int64_t)(rbase + i * 16 + r) * p.ldr + cbase;
#pragma unroll
          for (int j = 0; j < TN; ++j) ex[j][r] = rp[j * 16];
        }
      }
      if (p.accumulate) {
        float pc[TN][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* cp = C + (int64_t)(rbase + i * 16 + r) * p.ldc + cbase;
#pragma unroll
          for (int j = 0; j < TN; ++j) pc[j][r] = cp[j * 16];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) ex[j][r] += pc[j][r];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = p.alpha * acc[i][j][r] + bcol[j];
          if (p.act == SATT_ACT_RELU) v = fmaxf(v, 0.f);
          else if (p.act == SATT_ACT_TANH) v = tanhf(v);
          else if (p.act == SATT_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
          else if (p.act == SATT_ACT_SOFTSIGN) v = v / (1.f + fabsf(v));
          if (p.drop_thresh != 0)
            v = satt_keep(seed, p.drop_stream, (uint32_t)(rbase + i * 16 + r) * (uint32_t)p.N + (uint32_t)(cbase + j * 16), p.drop_thresh)
                    ? v * p.drop_scale : 0.f;
          ex[j][r] += v;
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* cp = C + (int64_t)(rbase + i * 16 + r) * p.ldc + cbase;
#pragma unroll
        for (int j = 0; j < TN; ++j) cp[j * 16] = ex[j][r];
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rbase + i * 16 + r, col = cbase + j * 16;
        if (row < p.M && col < p.N) {
          float v = p.alpha * acc[i][j][r] + bcol[j];
          float* dst = C + (int64_t)row * p.ldc + col;
          if (p.act == SATT_ACT_RELU) v = fmaxf(v, 0.f);
          else if (p.act == SATT_ACT_TANH) v = tanhf(v);
          else if (p.act == SATT_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
          else if (p.act == SATT_ACT_SOFTSIGN) v = v / (1.f + fabsf(v));
          if (p.drop_thresh != 0)
            v = satt_keep(seed, p.drop_stream, (uint32_t)row * (uint32_t)p.N + (uint32_t)col, p.drop_thresh)
                    ? v * p.drop_scale : 0.f;
          if (p.residual) v += p.residual[(int64_t)row * p.ldr + col];
          if (p.accumulate) v += *dst;
          *dst = v;
        }
      }
}

// ------------------------------------------------------------------------------------------------ conv_bank_fwd_k
// The conv bank's forward - and its input gradient, see the epilogue - (r6): widths 1 .. ng (<= 16) over the same [B*T, 128] input, 128 filters each.  On gemm_rk_k a K step of a
// 64 x 128 tile moves 16 KB (8 KB of it fp32 input rows, fetched again for every tap and every width) for 0.5 MFLOP: 713 MB through
// the L2s per launch, 32 flop per byte, 0.16 of the MFMA peak.  Here a workgroup owns one (128-row tile, width) job:
//   - the input rows of the tile plus its halo go to LDS ONCE, as bf16, [16 chunks of 8 channels][image row][8] - a tap is a row
//     offset of the fragment read, so the A operand costs no global traffic and no conversion inside the K loop.  Samples are kept
//     apart in the image by `IPAD` rows of zeros (image row = global row + IPAD x samples crossed): a shifted read that leaves its
//     sample lands on zeros, no per-tap masks.  Chunk stride = nr x 16 bytes with nr % 16 == 0: the 16 lanes of a ds_read_b128
//     group ({0-3, 12-15} of chunk c, {4-11} of chunk c + 1) hit 16 different 16-byte slots at any row shift.
//   - only the weights stream: 64-wide K stages (128 filters x 64 channels of one tap, 16 KB, double-buffered in LDS behind PD
//     register stages), 128 x 128 x 64 per stage = 2 MFLOP per 16 KB - 128 flop per byte, 178 MB per launch.
//   - 2 x 2 waves of 64 x 64: 32 MFMAs per wave and barrier; operands swapped (D = W-fragment x X-fragment = C^T) so that a lane
//     holds 4 consecutive filters of one row: the output leaves as 16-byte stores.
// Jobs differ 16 : 1 in length (taps).  blockIdx -> job: longest first for the first 256 workgroups, the NEXT 256 in ascending
// order (the workgroup that joins a CU's first one complements it: every pair sums to the same number of stages), then the rest.
constexpr int CB_BM = 128, CB_BN = 128, CB_BK = 64, CB_C = 128, CB_PAD = 8, CB_NR_MAX = 176;
constexpr int CB_BSTG = CB_BN * CB_BK;              // bf16 elements of a weight stage
constexpr int CB_PD = 2;

__host__ __device__ inline int cb_image_rows(int T, int ng) {
  const int L = CB_BM + ng - 1;
  const int nb = (L - 1) / T + 1;                   // sample boundaries inside L consecutive rows (upper bound)
  return (L + CB_PAD * nb + 15) / 16 * 16;
}

__global__ __launch_bounds__(TNT, 2) void conv_bank_fwd_k(const satt_gemm_params p, const int ntm, const int nr) {
  extern __shared__ __attribute__((aligned(16))) uint16_t cb_lds[];
  uint16_t* img = cb_lds;                           // [16][nr][8]
  uint16_t* bst = cb_lds + 16 * nr * 8;             // 2 x [8][128][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int kq = lane >> 4, l15 = lane & 15;

  const int J = ntm * p.bank_ng, n1 = min(J, 256), n2 = min(J, 512);
  const int sidx = (int)blockIdx.x;
  const int idx = sidx < n1 ? sidx : sidx < n2 ? n1 + (n2 - 1 - sidx) : sidx;
  const int rk = idx / ntm, mt = idx - rk * ntm;
  // row shift of tap t: forward t - g / 2 (conv_sgn = 1), input gradient g / 2 - t (conv_sgn = -1: the transposed convolution)
  const bool fwd = p.conv_sgn > 0;
  const int g = p.bank_ng - 1 - rk, taps = g + 1, HL = fwd ? g / 2 : g - g / 2, HR = g - HL;
  const int T = p.conv_T, m0 = mt * CB_BM;
  const int mo = m0 - HL, sbase = max(mo, 0) / T;
  const uint16_t* __restrict__ Bg = p.Bs + p.bank_b_unit * (int64_t)(g * (g + 1) / 2);
  const int nst = 2 * taps;

  // weight stages: segment e = tid + 256 q -> (filter e / 8, chunk e % 8); LDS slot (chunk, filter ^ chunk)
  int brel[4], boff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + TNT * q, n = e >> 3, sq = e & 7;
    brel[q] = n * (int)p.sbs_n + sq * 8;
    boff[q] = (sq * CB_BN + (n ^ sq)) * 8;
  }
  u32x4_t rb[CB_PD][4];
  auto gload = [&](auto SC, int st) {
    constexpr int S = decltype(SC)::value;
    const uint16_t* src = Bg + (int64_t)(st >> 1) * p.sbs_tap + (st & 1) * CB_BK;
#pragma unroll
    for (int q = 0; q < 4; ++q) rb[S][q] = *reinterpret_cast<const u32x4_t*>(src + brel[q]);
  };
  auto swrite = [&](auto SC, int buf) {
    constexpr int S = decltype(SC)::value;
    uint16_t* base = bst + buf * CB_BSTG;
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x4_t*>(base + boff[q]) = rb[S][q];
  };
  gload(std::integral_constant<int, 0>{}, 0);

  // ---- the image: zeros, then the rows [max(0, mo), min(M, m0 + BM + HR))
  for (int i = tid; i < nr * 16; i += TNT) reinterpret_cast<u32x4_t*>(img)[i] = (u32x4_t){0u, 0u, 0u, 0u};
  lds_barrier();
  {
    const int lo = max(mo, 0), hi = min(p.M, m0 + CB_BM + HR);
    const int nunits = ((hi - lo + 7) >> 3) * 2;                 // (8 rows) x (8 chunks) per wave pass
    const float* __restrict__ X = p.A + (int64_t)g * p.bank_a_col;
    for (int u0 = wave; u0 < nunits; u0 += 12) {
      float4 v[3][2]; int dst[3];
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        const int u = u0 + 4 * w;
        const int m = lo + (u >> 1) * 8 + (lane & 7), ch = (u & 1) * 8 + (lane >> 3);
        const bool ok = u < nunits && m < hi;
        const float4* src = reinterpret_cast<const float4*>(X + (int64_t)(ok ? m : lo) * p.lda + ch * 8);
        v[w][0] = src[0]; v[w][1] = src[1];
        dst[w] = ok ? (ch * nr + (m - mo) + CB_PAD * (m / T - sbase)) * 8 : -1;
      }
#pragma unroll
      for (int w = 0; w < 3; ++w)
        if (dst[w] >= 0) {
          u32x4_t o;
          o[0] = pack_bf16x2(v[w][0].x, v[w][0].y); o[1] = pack_bf16x2(v[w][0].z, v[w][0].w);
          o[2] = pack_bf16x2(v[w][1].x, v[w][1].y); o[3] = pack_bf16x2(v[w][1].z, v[w][1].w);
          *reinterpret_cast<u32x4_t*>(img + dst[w]) = o;
        }
    }
  }
  swrite(std::integral_constant<int, 0>{}, 0);
  static_for<CB_PD>([&](auto Jc) { constexpr int d = decltype(Jc)::value + 1; if (nst > d) gload(std::integral_constant<int, d % CB_PD>{}, d); });

  // per-lane image address (elements) of tap 0 / chunk kq of the lane's row in each of the 4 row fragments
  int aaddr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = min(m0 + wm * 64 + i * 16 + l15, p.M - 1);
    aaddr[i] = (kq * nr + (m - mo) + CB_PAD * (m / T - sbase) + (fwd ? -HL : HR)) * 8;
  }
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int tsgn = fwd ? 1 : -1;
  lds_barrier();

  for (int kt0 = 0; kt0 < nst; kt0 += CB_PD)
  static_for<CB_PD>([&](auto Jc) {
    const int kt = kt0 + decltype(Jc)::value;
    if (kt >= nst) return;
    typedef std::integral_constant<int, (decltype(Jc)::value + 1) % CB_PD> SN;
    const uint16_t* Bt = bst + (kt & 1) * CB_BSTG;
    const int aoff = (tsgn * (kt >> 1) + (kt & 1) * 8 * nr) * 8;   // tap rows down (forward) / up, 8 chunks across per half
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bf16x8_t a[4], b[4];
      const int cb = h * 4 + kq;
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(img + aaddr[i] + aoff + h * 4 * nr * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(Bt + (cb * CB_BN + wn * 64 + j * 16 + (l15 ^ cb)) * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nst) {
      swrite(SN{}, (kt + 1) & 1);
      if (kt + 1 + CB_PD < nst) gload(SN{}, kt + 1 + CB_PD);
    }
    lds_barrier();
  });

  // D = C^T fragment: lane -> row m = l15 of fragment i, filters 4 (lane >> 4) .. + 4 of fragment j
  // forward: the width's 128 columns of C.  Input gradient (bank_c_col == 0: every width adds into the same C): the width's own
  // dense [M][128] slab of the workspace; slab_reduce_k sums the ng slabs into C in a fixed order.
  const bool slab = p.bank_c_col == 0;
  float* __restrict__ C = slab ? p.ws + (int64_t)g * p.M * CB_BN : p.C + (int64_t)g * p.bank_c_col;
  const int64_t ldc = slab ? CB_BN : p.ldc;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + l15;
    if (m < p.M) {
      float* cp = C + (int64_t)m * ldc + wn * 64 + kq * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4_t*>(cp + j * 16) = acc[i][j];
    }
  }
}

// ------------------------------------------------------------------------------------------------ gemm_rows_k
// Few rows per batch, long reduction: the per-chunk products of the recurrent pipelines (B samples x a chunk of 8..32 decoder
// steps against [1024, .] gate weights - ops.linear_dx_rows / linear_rows).  On 64-row tiles such a product is ONE dependent
// chain of K / 32 load -> LDS -> MFMA stages per workgroup (K = 1024: 32 stages, 23-25 us for 1 GFLOP, on the LSTM stream's
// critical path at both ends of the pipelines).  Here the reduction is split INSIDE the workgroup: the four waves take a
// quarter of K each, every global operand of a wave (A rows fp32 -> bf16 fragments, B fragments straight from the
// k-contiguous bf16 shadow) is requested before its first MFMA - no LDS staging, one memory round trip - and the four
// partial 16 TM x 64 tiles are summed through LDS in a fixed order (deterministic, no atomics).
// grid (N tiles of 64, row tiles of 16 TM, batches); same operands, rounding and epilogue subset (alpha, bias, previous C)
// as gemm_rk_k.
constexpr int RW_KS = 8;          // K steps of 32 per wave held in flight (K <= 4 * 8 * 32 per pass)

template <int TM>
__global__ __launch_bounds__(TNT) void gemm_rows_k(const satt_gemm_params p) {
  __shared__ __attribute__((aligned(16))) float red[4 * TM * 16 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = (int)blockIdx.x * 64, m0 = (int)blockIdx.y * 16 * TM;
  const int z = (int)blockIdx.z, zo = z / p.nb_inner, zi = z - zo * p.nb_inner;
  const float* __restrict__ A = p.A + zo * p.strideA_o + zi * p.strideA_i;
  float* __restrict__ C = p.C + zo * p.strideC_o + zi * p.strideC_i;
  const uint16_t* __restrict__ Bs = p.Bs;
  const int fr = lane & 15, kq = lane >> 4;
  // this wave's share of the reduction: whole 32-wide steps
  const int steps = (p.K + 31) / 32, per = (steps + 3) / 4;
  const int s0 = wave * per, s1 = min(steps, s0 + per);
  f32x4_t acc[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const float* arow[TM]; bool aok[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + 16 * i + fr;
    aok[i] = m < p.M;
    arow[i] = A + (int64_t)min(m, p.M - 1) * p.lda;
  }
  const uint16_t* brow[4]; bool bok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + 16 * j + fr;
    bok[j] = n < p.N;
    brow[j] = Bs + (int64_t)min(n, p.N - 1) * p.sbs_n;
  }
  for (int sb = s0; sb < s1; sb += RW_KS) {
    // every load of up to RW_KS steps first (clamped, always valid addresses; masked when consumed), then the MFMAs
    float4 ra[RW_KS][TM][2]; u32x4_t rb[RW_KS][4];
#pragma unroll
    for (int u = 0; u < RW_KS; ++u) {
      const int k = min((sb + u) * 32 + 8 * kq, p.K - 8);        // K % 8 == 0, K >= 8
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ra[u][i][0] = *reinterpret_cast<const float4*>(arow[i] + k);
        ra[u][i][1] = *reinterpret_cast<const float4*>(arow[i] + k + 4);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) rb[u][j] = *reinterpret_cast<const u32x4_t*>(brow[j] + k);
    }
#pragma unroll
    for (int u = 0; u < RW_KS; ++u) {
      const bool kok = sb + u < s1 && (sb + u) * 32 + 8 * kq < p.K;
      bf16x8_t a[TM], b[4];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        u32x4_t w;
        w[0] = pack_bf16x2(ra[u][i][0].x, ra[u][i][0].y); w[1] = pack_bf16x2(ra[u][i][0].z, ra[u][i][0].w);
        w[2] = pack_bf16x2(ra[u][i][1].x, ra[u][i][1].y); w[3] = pack_bf16x2(ra[u][i][1].z, ra[u][i][1].w);
        if (!(kok && aok[i])) w = (u32x4_t){0u, 0u, 0u, 0u};
        a[i] = __builtin_bit_cast(bf16x8_t, w);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32x4_t w = rb[u][j];
        if (!(kok && bok[j])) w = (u32x4_t){0u, 0u, 0u, 0u};
        b[j] = __builtin_bit_cast(bf16x8_t, w);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // partial tiles -> LDS [wave][row][64]; element (row 16 i + 4 kq + r, col 16 j + fr)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * TM * 16 + 16 * i + 4 * kq + r) * 64 + 16 * j + fr] = acc[i][j][r];
  lds_barrier();
  // thread -> (row, 4 consecutive columns): TM * 16 * 16 float4 outputs over 256 threads
#pragma unroll
  for (int e = tid; e < TM * 16 * 16; e += TNT) {
    const int row = e >> 4, c4 = (e & 15) * 4;
    const int m = m0 + row, n = n0 + c4;
    if (m >= p.M || n >= p.N) continue;
    float4 v = *reinterpret_cast<const float4*>(red + row * 64 + c4);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 t = *reinterpret_cast<const float4*>(red + (w * TM * 16 + row) * 64 + c4);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    float o[4] = {p.alpha * v.x, p.alpha * v.y, p.alpha * v.z, p.alpha * v.w};
    float* dst = C + (int64_t)m * p.ldc + n;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (n + q < p.N) {
        float x = o[q];
        if (p.bias) x += p.bias[n + q];
        if (p.accumulate) x += dst[q];
        dst[q] = x;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ gemm_dw_k
// C[i][j] += alpha * sum_k A(i, k) * B(k, j);  A(i, k) = x[(k + shift_i) * lda + c_i] with i = (tap, c),
// shift_i = conv_sgn * tap + conv_off, zero where the shifted step leaves [0, conv_T) of its sample (a_mode 3; a_mode 1
// is the single-tap case without boundaries); B(k, j) = B[k * sb_k + j].
// A thread owns a 4 x 4 block (4 consecutive columns of the source x 4 consecutive reduction rows) of each operand and
// writes it transposed: one 8-byte LDS store per column.  Inside every block of 64 tile rows, logical row 4 q + jj lives
// at physical row 16 jj + q: the 16 lanes of a store group write 16 consecutive physical rows, and in the epilogue a lane
// finds the four accumulators of 4 CONSECUTIVE output columns in its own registers (one 16-byte store / read-modify-write).
// Output: splitk == 1 -> C += tile (plain read-modify-write: the caller guarantees nobody else updates C concurrently);
// splitk > 1 with a workspace -> partial tile to slab blockIdx.z (summed into C by slab_reduce_k); without -> atomics.
// Conv-bank form (bank_ng > 0): ONE launch computes the weight gradients of all widths 1..ng over the same x: group g has
// (g + 1) * conv_C rows (tap, c), conv_off = -conv_sgn * (g / 2), B columns start at g * bank_c_col, and its rows start
// at conv_C * g (g + 1) / 2 of C (the weights of all widths are contiguous) - tiles never straddle groups.
__device__ __forceinline__ int perm64(int pr) { return (pr & ~63) | ((pr & 15) << 2) | ((pr & 63) >> 4); }   // physical -> logical

template <int BM, int PD>
__global__ __launch_bounds__(TNT) void gemm_dw_k(const satt_gemm_params p, const int ntn, const int ntiles,
                                                 const int64_t slab_rows) {
  constexpr int BK = TBK, BN = 128;
  constexpr int TM = BM / 32, TN = BN / 32;
  constexpr int STG = (BM + BN) * BK;
  constexpr int QA = BM / 4, QB = BN / 4;           // column quads per operand; 8 row quads of 4 reduction rows
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * STG];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // Workgroup -> (tile, reduction slice).  Workgroups are dealt to the 8 XCDs round robin in launch order (x fastest, then z), and
  // each XCD has its own L2.  Slice-major placement (r3): with splitk % 8 == 0 every XCD takes WHOLE reduction slices - all tiles
  // of slice zk run on XCD zk % 8, back to back - so the rows of A and B of a slice are pulled from HBM ONCE, into that XCD's L2,
  // and every tile of the slice re-reads them there.  The tile-major map (every XCD a range of tiles over all slices) made each
  // XCD read its operand COLUMNS over all rows: with 5 x 8 tiles the [12800, 1024] gradient operand crossed the fabric 5 times
  // (rocprofv3 FETCH_SIZE over the weight-gradient launches of a step: 2.2 GB against 0.76 GB of operands; 0.97 GB with this map).
  int lin, zk_;
  if (p.bank_ng == 0 && (p.splitk & 7) == 0) {
    const int L = (int)blockIdx.x + ntiles * (int)blockIdx.z, x = L & 7, j = L >> 3;
    zk_ = x + 8 * (j / ntiles);
    lin = j % ntiles;
  } else {
    lin = xcd_remap((int)blockIdx.x, ntiles);
    zk_ = blockIdx.z;
  }
  // conv bank: tiles are enumerated group by group, widest first
  int Mg = p.M, conv_off = p.conv_off;
  int64_t crow0 = 0;
  const float* Bbase = p.B;
  if (p.bank_ng > 0) {
    int g = p.bank_ng - 1;
    for (; g > 0; --g) {
      const int tg = (((g + 1) * p.conv_C + BM - 1) / BM) * ntn;
      if (lin < tg) break;
      lin -= tg;
    }
    Mg = (g + 1) * p.conv_C;
    conv_off = -p.conv_sgn * (g / 2);
    crow0 = (int64_t)p.conv_C * (g * (g + 1) / 2);
    Bbase = p.B + (int64_t)g * p.bank_c_col;
  }
  const int tm = lin / ntn, tn = lin - tm * ntn;
  const int m0 = tm * BM, n0 = tn * BN;
  const int zk = zk_;
  int chunk = (p.K + p.splitk - 1) / p.splitk;
  chunk = (chunk + BK - 1) / BK * BK;
  const int kbeg = zk * chunk, kend = min(p.K, kbeg + chunk);
  const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  const bool shifted = p.a_mode == 3;
  const int T = shifted ? p.conv_T : 0x3fffffff;

  // operand A: thread (iq, mq)
  const bool a_act = tid < QA * 8;
  const int aiq = tid % QA, amq = tid / QA;
  const int ai = m0 + 4 * aiq;
  const bool a_ok = a_act && ai < Mg;
  int ashift = 0, ac = ai;
  if (shifted) { const int tap = ai / p.conv_C; ac = ai - tap * p.conv_C; ashift = p.conv_sgn * tap + conv_off; }
  const float* ap = p.A + ac + (int64_t)(kbeg + 4 * amq + ashift) * p.lda;
  int at0 = shifted ? (kbeg + 4 * amq) % T : kbeg + 4 * amq;
  const int akq = amq >> 1, ah = amq & 1;
  const int aprow = (aiq >> 4) * 64 + (aiq & 15);          // + 16 * ii
  // operand B: thread (jq, mq)
  const int bjq = tid % QB, bmq = tid / QB;
  const int bj = n0 + 4 * bjq;
  const bool b_ok = bj < p.N;
  const float* bp = Bbase + bj + (int64_t)(kbeg + 4 * bmq) * p.sb_k;
  const int bkq = bmq >> 1, bh = bmq & 1;
  const int bprow = (bjq >> 4) * 64 + (bjq & 15);

  // PD register stages of the coming K steps (see gemm_rk_k): raw loads at clamped addresses, masked when written to LDS
  float4 ra[PD][4], rb[PD][4];
  bool rao[PD][4], rbo[PD][4];
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  int kload = kbeg;
  auto gload = [&](auto SC) {
    constexpr int S = decltype(SC)::value;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int t = at0 + r; if (t >= T) t -= T;
      const bool ok = a_ok && kload + 4 * amq + r < kend && (unsigned)(t + ashift) < (unsigned)T;
      ra[S][r] = *reinterpret_cast<const float4*>(ok ? ap + (int64_t)r * p.lda : p.A);
      rao[S][r] = ok;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = b_ok && kload + 4 * bmq + r < kend;
      rb[S][r] = *reinterpret_cast<const float4*>(ok ? bp + (int64_t)r * p.sb_k : p.B);
      rbo[S][r] = ok;
    }
    kload += BK;
    ap += (int64_t)BK * p.lda; bp += (int64_t)BK * p.sb_k;
    at0 += BK; while (at0 >= T) at0 -= T;
  };
  auto swrite = [&](auto SC, int buf) {
    constexpr int S = decltype(SC)::value;
    uint16_t* base = lds + buf * STG;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (!rao[S][r]) ra[S][r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!rbo[S][r]) rb[S][r] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (a_act) {
      const float* f = reinterpret_cast<const float*>(ra[S]);
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        uint2 w;
        w.x = pack_bf16x2(f[0 * 4 + ii], f[1 * 4 + ii]); w.y = pack_bf16x2(f[2 * 4 + ii], f[3 * 4 + ii]);
        const int prow = aprow + 16 * ii;
        *reinterpret_cast<uint2*>(base + (akq * BM + (prow ^ swz(akq))) * 8 + ah * 4) = w;
      }
    }
    {
      const float* f = reinterpret_cast<const float*>(rb[S]);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        uint2 w;
        w.x = pack_bf16x2(f[0 * 4 + jj], f[1 * 4 + jj]); w.y = pack_bf16x2(f[2 * 4 + jj], f[3 * 4 + jj]);
        const int prow = bprow + 16 * jj;
        *reinterpret_cast<uint2*>(base + BM * BK + (bkq * BN + (prow ^ swz(bkq))) * 8 + bh * 4) = w;
        cs[jj] += (f[0 * 4 + jj] + f[1 * 4 + jj]) + (f[2 * 4 + jj] + f[3 * 4 + jj]);
      }
    }
  };

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    gload(std::integral_constant<int, 0>{}); swrite(std::integral_constant<int, 0>{}, 0);
    lds_barrier();
    static_for<PD>([&](auto J) { constexpr int d = decltype(J)::value + 1; if (nk > d) gload(std::integral_constant<int, d % PD>{}); });
  }
  const int kq = lane >> 4, pr = (lane & 15) ^ swz(kq);
  for (int kt0 = 0; kt0 < nk; kt0 += PD)
  static_for<PD>([&](auto J) {
    const int kt = kt0 + decltype(J)::value;
    if (kt >= nk) return;
    typedef std::integral_constant<int, (decltype(J)::value + 1) % PD> SN;      // the set that holds stage kt + 1
    const uint16_t* As = lds + (kt & 1) * STG;
    const uint16_t* Bt = As + BM * BK;
    bf16x8_t a[TM], b[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
      a[i] = *reinterpret_cast<const bf16x8_t*>(As + (kq * BM + wm * (BM / 2) + i * 16 + pr) * 8);
#pragma unroll
    for (int j = 0; j < TN; ++j)
      b[j] = *reinterpret_cast<const bf16x8_t*>(Bt + (kq * BN + wn * (BN / 2) + j * 16 + pr) * 8);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    if (kt + 1 < nk) {
      swrite(SN{}, (kt + 1) & 1);
      if (kt + 1 + PD < nk) gload(SN{});
    }
    lds_barrier();
  });

  // epilogue: the wave's 64 output columns are one permutation block: acc[i][0..3][r] of lane q = l & 15 are the logical
  // columns n0 + 64 wn + 4 q + {0, 1, 2, 3}
  const bool slab_out = p.ws != nullptr && p.splitk > 1;
  const bool rmw = p.splitk == 1;
  const int col = n0 + wn * 64 + 4 * (lane & 15);
  const bool vec = rmw ? ((p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0) : true;
  // One loop per output mode (uniform branch outside): inside a loop that mixes stores, atomics and loads behind per-element
  // branches the compiler falls back to s_waitcnt vmcnt(0) in front of every store.  Read-modify-write form: the previous
  // values of every owned quad first (clamped addresses), then adds and stores (see gemm_rk_k's epilogue).
  auto quad = [&](int i, int r) {
    return make_float4(p.alpha * acc[i][0][r], p.alpha * acc[i][1][r], p.alpha * acc[i][2][r], p.alpha * acc[i][3][r]);
  };
  if (slab_out) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + perm64(wm * (BM / 2) + i * 16 + (lane >> 4) * 4 + r);
        if (row < Mg && col < p.N)          // N % 4 == 0: a column quad is inside or outside as a whole
          *reinterpret_cast<float4*>(p.ws + ((int64_t)zk * slab_rows + crow0 + row) * p.N + col) = quad(i, r);
      }
  } else if (rmw && vec) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {        // a fragment at a time: four quads of previous values, then four stores
      float4 prev[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = min(m0 + perm64(wm * (BM / 2) + i * 16 + (lane >> 4) * 4 + r), Mg - 1);
        prev[r] = *reinterpret_cast<const float4*>(p.C + (crow0 + row) * p.ldc + min(col, p.N - 4));
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + perm64(wm * (BM / 2) + i * 16 + (lane >> 4) * 4 + r);
        if (row < Mg && col < p.N) {
          const float4 v = quad(i, r);
          float4 o = prev[r];
          o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
          *reinterpret_cast<float4*>(p.C + (crow0 + row) * p.ldc + col) = o;
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + perm64(wm * (BM / 2) + i * 16 + (lane >> 4) * 4 + r);
        if (row < Mg && col < p.N) {
          const float4 v = quad(i, r);
          float* dst = p.C + (crow0 + row) * p.ldc + col;
          if (rmw) {
            dst[0] += v.x; dst[1] += v.y; dst[2] += v.z; dst[3] += v.w;
          } else {
            atomicAdd(dst, v.x); atomicAdd(dst + 1, v.y); atomicAdd(dst + 2, v.z); atomicAdd(dst + 3, v.w);
          }
        }
      }
  }
  // fused bias gradient: column sums of B over this block's reduction range, once per column tile (tm == 0)
  if (p.colsum && tm == 0) {
    float* red = reinterpret_cast<float*>(lds);
    if (tid < BN) red[tid] = 0.f;
    __syncthreads();
    if (b_ok) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) atomicAdd(red + 4 * bjq + jj, cs[jj]);
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) atomicAdd(p.colsum + n0 + tid, red[tid]);
  }
}

// C[m][n] (+)= sum_s ws[s][m][n]: the second half of a split reduction (slabs are dense [M][N])
__global__ __launch_bounds__(256) void slab_reduce_k(const float* __restrict__ ws, int nslab, int64_t M, int N, float* __restrict__ C,
                                                     int64_t ldc, int accumulate) {
  const int nq = N >> 2;
  const int64_t total = M * nq, slab = M * N;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = e / nq; const int q = (int)(e - m * nq);
    const float* src = ws + m * N + 4 * q;
    float4 a = *reinterpret_cast<const float4*>(src);
    for (int s = 1; s < nslab; ++s) {
      const float4 b = *reinterpret_cast<const float4*>(src + s * slab);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float* dst = C + m * ldc + 4 * q;
    if ((ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0) {
      if (accumulate) { const float4 o = *reinterpret_cast<const float4*>(dst); a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
      *reinterpret_cast<float4*>(dst) = a;
    } else {
      if (accumulate) { a.x += dst[0]; a.y += dst[1]; a.z += dst[2]; a.w += dst[3]; }
      dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
    }
  }
}

// ------------------------------------------------------------------------------------------------ shadow pack
// table entry w: {offset, taps, rows, cols} (int64 x 4) of a weight [taps][rows][cols] inside the flat fp32 buffer.
//   sn[offset + e] = bf16(flat[offset + e])                                  (plain cast: the dX orientation)
//   st[offset + (tap * cols + c) * rows + r] = bf16(flat[offset + (tap * rows + r) * cols + c])   (per-tap transpose)
__global__ __launch_bounds__(256) void shadow_pack_k(const float* __restrict__ flat, const int64_t* __restrict__ table,
                                                     uint16_t* __restrict__ st, uint16_t* __restrict__ sn) {
  __shared__ float tile[32][33];
  const int64_t* e = table + 4 * blockIdx.y;
  const int64_t off = e[0];
  const int taps = (int)e[1], R = (int)e[2], Cc = (int)e[3];
  const int tr = (R + 31) / 32, tc = (Cc + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
  for (int t = blockIdx.x; t < taps * tr * tc; t += gridDim.x) {
    const int tap = t / (tr * tc), rem = t - tap * (tr * tc);
    const int r0 = (rem / tc) * 32, c0 = (rem % tc) * 32;
    const int64_t base = off + (int64_t)tap * R * Cc;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = r0 + ty + 8 * k, c = c0 + tx;
      if (r < R && c < Cc) {
        const float v = flat[base + (int64_t)r * Cc + c];
        tile[ty + 8 * k][tx] = v;
        sn[base + (int64_t)r * Cc + c] = f2bf(v);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + ty + 8 * k, r = r0 + tx;
      if (r < R && c < Cc) st[base + (int64_t)c * R + r] = f2bf(tile[tx][ty + 8 * k]);
    }
  }
}

inline bool a16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <int BM, int BN>
void launch_rk(const satt_gemm_params& p, int nz, hipStream_t s) {
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  dim3 grid(ntm * ntn, 1, nz);
  static const int pd = [] { const char* e = getenv("SATT_RK_PREFETCH"); return e ? atoi(e) : RK_PREFETCH_DEFAULT; }();
#define SATT_RK(PDV)                                                                                                              \
  do {                                                                                                                          \
    if (p.a_mode == 2) hipLaunchKernelGGL((gemm_rk_k<BM, BN, true, PDV>), grid, dim3(TNT), 0, s, p, ntn, ntm * ntn);           \
    else hipLaunchKernelGGL((gemm_rk_k<BM, BN, false, PDV>), grid, dim3(TNT), 0, s, p, ntn, ntm * ntn);                         \
  } while (0)
  if (pd == 2) SATT_RK(2); else if (pd == 3) SATT_RK(3); else if (pd == 4) SATT_RK(4); else SATT_RK(1);
#undef SATT_RK
}
template <int BM>
void launch_dw(const satt_gemm_params& p, int64_t slab_rows, hipStream_t s) {
  const int ntn = (p.N + 127) / 128;
  int tiles = 0;
  if (p.bank_ng > 0) for (int g = 0; g < p.bank_ng; ++g) tiles += (((g + 1) * p.conv_C + BM - 1) / BM) * ntn;
  else tiles = ((p.M + BM - 1) / BM) * ntn;
  dim3 grid(tiles, 1, p.splitk);
  static const int pd = [] { const char* e = getenv("SATT_DW_PREFETCH"); return e ? atoi(e) : DW_PREFETCH_DEFAULT; }();
  if (pd == 2) hipLaunchKernelGGL((gemm_dw_k<BM, 2>), grid, dim3(TNT), 0, s, p, ntn, tiles, slab_rows);
  else if (pd == 3) hipLaunchKernelGGL((gemm_dw_k<BM, 3>), grid, dim3(TNT), 0, s, p, ntn, tiles, slab_rows);
  else hipLaunchKernelGGL((gemm_dw_k<BM, 1>), grid, dim3(TNT), 0, s, p, ntn, tiles, slab_rows);
}
void launch_reduce(const float* ws, int nslab, int64_t M, int N, float* C, int64_t ldc, int accumulate, hipStream_t s) {
  const int64_t total = M * (N / 4);
  const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(slab_reduce_k, dim3(blocks), dim3(256), 0, s, ws, nslab, M, N, C, ldc, accumulate);
}

}  // namespace

// Tile choice, from the sweep in profiles/r02_gemm_tile_sweep.txt (B = 32 shapes of the train step): with fp32
// activations these products are bound by memory latency and the output stream, not by MFMA issue, so MORE workgroups in
// flight beat operand reuse - 64 x 64 tiles for the short reductions (K < 512: 12800 x 1024 x 128 runs in 24 us against
// 43 us on 128 x 128 tiles), 64 x 128 for the long ones and for the convolutions (conv bank forward 61 us vs 74 / 78 us on
// 128 x 128 / 64 x 64).  128-row tiles stay available through the override for problems with far more tiles than CUs.
static void pick_tile(const satt_gemm_params& p, int nz, int& bm, int& bn) {
  (void)nz;
  bm = 64;
  bn = (p.a_mode == 2 || p.K >= 512) && p.N > 64 ? 128 : 64;
  static const int fbm = [] { const char* e = getenv("SATT_TILE_BM"); return e ? atoi(e) : 0; }();     // tuning overrides
  static const int fbn = [] { const char* e = getenv("SATT_TILE_BN"); return e ? atoi(e) : 0; }();
  if (fbm == 64 || fbm == 128) bm = fbm;
  if (fbn == 64 || fbn == 128) bn = fbn;
}

static bool rk_eligible(const satt_gemm_params& p) {
  if (p.precision != SATT_PREC_BF16 || !p.Bs) return false;
  if (p.a_mode != 0 && p.a_mode != 2) return false;
  if (!a16(p.A) || !a16(p.Bs) || p.lda % 4 || p.strideA_o % 4 || p.strideA_i % 4) return false;
  if (p.strideB_o || p.strideB_i) return false;                 // the shadow is shared by every batch
  if (p.K % 8 || p.sbs_n % 8 || p.sbs_tap % 8) return false;
  if (p.a_mode == 2) {
    if (p.conv_C % TBK || p.kin % TBK) return false;
    if (p.bank_ng > 0 && (p.bank_a_col % 4 || p.bank_b_unit % 8)) return false;
  } else {
    if (p.kin < p.K || p.bank_ng > 0) return false;
  }
  const int nz = p.bank_ng > 0 ? p.bank_ng : p.nb_outer * p.nb_inner * p.splitk;
  return nz <= 65535;
}
// a workspace is used only where slabs can be dense [M][N] float4 rows of ONE problem
static bool rk_ws_usable(const satt_gemm_params& p) {
  return p.ws && a16(p.ws) && p.N % 4 == 0 && p.nb_outer * p.nb_inner == 1 &&
         (p.splitk > 1 || (p.bank_ng > 0 && p.bank_c_col == 0));
}

// blockIdx.z extent of gemm_rk_k: batches x splits, or the groups of a bank (paired when they all add into one C)
static int rk_nz(const satt_gemm_params& p) {
  if (p.bank_ng > 0) return p.bank_c_col == 0 ? (p.bank_ng + 1) / 2 : p.bank_ng;
  return p.nb_outer * p.nb_inner * p.splitk;
}

// few rows per batch against a long reduction: the in-workgroup split (gemm_rows_k)
static bool rows_eligible(const satt_gemm_params& p) {
  static const int off = [] { const char* e = getenv("SATT_NO_ROWS_GEMM"); return e ? atoi(e) : 0; }();
  if (off) return false;
  if (p.a_mode != 0 || p.bank_ng > 0 || p.splitk != 1 || p.act || p.residual || p.drop_thresh) return false;
  // measured beside the recurrent cluster kernels (half of the CUs free): [16..32 rows x 32 samples] x 1024 -> 256 columns 24 -> 9-12 us,
  // x 768 -> 256 21 -> 12 us; wide outputs (544 / 1024 columns: 9-16 column tiles per sample, each fetching its own weight
  // slice) and short reductions (K = 256 / 544: 8-16 us on the 64-row tiles) are NOT faster here and stay on gemm_rk_k
  if (p.M > 32 || p.K < 768 || p.N > 256) return false;
  return (int64_t)p.nb_outer * p.nb_inner <= 65535 && (p.M + 15) / 16 <= 65535;
}

// the conv bank on its own kernel (conv_bank_fwd_k): shapes of the ZoneoutCBHG bank, nothing fused behind the product.
// 0: no; 1: the forward (bank_c_col > 0, one input for all widths); 2: the input gradient (bank_c_col == 0: the widths' own gradient
// columns as inputs, one slab per width, summed by slab_reduce_k - needs the workspace)
static int bank_kernel_form(const satt_gemm_params& p) {
  static const int off = [] { const char* e = getenv("SATT_NO_CONV_BANK_KERNEL"); return e ? atoi(e) : 0; }();
  if (off || p.a_mode != 2 || p.bank_ng <= 0 || p.bank_ng > 2 * CB_PAD || p.conv_off != 0 || p.conv_T < 1) return 0;
  if (p.conv_C != CB_C || p.kin != CB_C || p.N != CB_BN || p.alpha != 1.f || p.bias || p.act || p.residual || p.drop_thresh || p.splitk != 1) return 0;
  if (p.nb_outer * p.nb_inner != 1 || !a16(p.C) || p.ldc % 4 || p.sbs_n % 8 || p.sbs_n < CB_C) return 0;
  if ((int64_t)p.sbs_n * CB_BN + CB_C >= (int64_t)1 << 31 || cb_image_rows(p.conv_T, p.bank_ng) > CB_NR_MAX) return 0;
  if (p.bank_c_col > 0) return (p.conv_sgn == 1 && p.bank_a_col == 0 && !p.accumulate && p.bank_c_col % 4 == 0) ? 1 : 0;
  return (p.conv_sgn == -1 && p.bank_a_col > 0 && p.bank_a_col % 4 == 0 && p.accumulate) ? 2 : 0;
}

bool satt_gemm_tile_rk(const satt_gemm_params& pp, hipStream_t s) {
  if (!rk_eligible(pp)) return false;
  const int form = bank_kernel_form(pp);
  if (form == 1 || (form == 2 && rk_ws_usable(pp))) {
    const int ntm = (pp.M + CB_BM - 1) / CB_BM, nr = cb_image_rows(pp.conv_T, pp.bank_ng);
    const size_t lds = (size_t)(16 * nr * 8 + 2 * CB_BSTG) * sizeof(uint16_t);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bank_fwd_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)((16 * CB_NR_MAX * 8 + 2 * CB_BSTG) * sizeof(uint16_t)));
    if (attr == hipSuccess) {
      hipLaunchKernelGGL(conv_bank_fwd_k, dim3(ntm * pp.bank_ng), dim3(TNT), lds, s, pp, ntm, nr);
      if (form == 2) launch_reduce(pp.ws, pp.bank_ng, pp.M, pp.N, pp.C, pp.ldc, pp.accumulate, s);
      return true;
    }
  }
  if (rows_eligible(pp)) {
    const dim3 grid((pp.N + 63) / 64, pp.M > 16 ? (pp.M + 31) / 32 : 1, pp.nb_outer * pp.nb_inner);
    if (pp.M > 16) hipLaunchKernelGGL((gemm_rows_k<2>), grid, dim3(TNT), 0, s, pp);
    else hipLaunchKernelGGL((gemm_rows_k<1>), grid, dim3(TNT), 0, s, pp);
    return true;
  }
  satt_gemm_params p = pp;
  if (!rk_ws_usable(p)) p.ws = nullptr;
  if (p.splitk > 1 && !p.accumulate && !p.ws) return false;     // overwriting splits need the slab workspace
  const int nz = rk_nz(p);
  int bm, bn;
  pick_tile(p, nz, bm, bn);
  if (bm == 128 && bn == 128) launch_rk<128, 128>(p, nz, s);
  else if (bm == 64 && bn == 128) launch_rk<64, 128>(p, nz, s);
  else if (bm == 128 && bn == 64) launch_rk<128, 64>(p, nz, s);
  else launch_rk<64, 64>(p, nz, s);
  if (p.ws) launch_reduce(p.ws, nz, p.M, p.N, p.C, p.ldc, p.accumulate, s);
  return true;
}

static bool dw_eligible(const satt_gemm_params& p) {
  if (p.precision != SATT_PREC_BF16) return false;
  if (p.a_mode != 1 && p.a_mode != 3) return false;
  if (p.nb_outer * p.nb_inner != 1) return false;
  if (!p.accumulate || p.bias || p.act || p.residual || p.drop_thresh) return false;
  if (p.sb_n != 1 || p.sb_k % 4 || p.kin < p.K) return false;
  if (!a16(p.A) || !a16(p.B) || p.lda % 4 || p.M % 4 || p.N % 4 || p.N <= 64) return false;
  if (p.a_mode == 3 && (p.conv_C % 4 || p.conv_T < 4)) return false;
  if (p.bank_ng > 0 && (p.a_mode != 3 || p.bank_c_col % 4 || p.ldc != p.N || p.colsum)) return false;
  return p.splitk <= 65535;
}

bool satt_gemm_tile_dw(const satt_gemm_params& pp, hipStream_t s) {
  if (!dw_eligible(pp)) return false;
  satt_gemm_params p = pp;
  if (p.ws && (!a16(p.ws) || p.splitk == 1)) p.ws = nullptr;
  // rows of one slab: all groups of a bank share it (group g starts at row conv_C * g (g + 1) / 2)
  const int64_t slab_rows = p.bank_ng > 0 ? (int64_t)p.conv_C * (p.bank_ng * (p.bank_ng + 1) / 2) : p.M;
  const int mmax = p.bank_ng > 0 ? p.bank_ng * p.conv_C : p.M;
  static const int fbm = [] { const char* e = getenv("SATT_TILE_BM"); return e ? atoi(e) : 0; }();
  int bm = mmax <= 64 ? 64 : 128;
  if (bm == 128 && p.bank_ng == 0 && (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.splitk < 160) bm = 64;
  if (fbm == 64 || fbm == 128) bm = fbm;
  if (bm == 128) launch_dw<128>(p, slab_rows, s); else launch_dw<64>(p, slab_rows, s);
  if (p.ws) launch_reduce(p.ws, p.splitk, slab_rows, p.N, p.C, p.ldc, 1, s);
  return true;
}

int satt_gemm_tile_path(const satt_gemm_params& p) { return rk_eligible(p) ? (bank_kernel_form(p) ? 3 : 1) : dw_eligible(p) ? 2 : 0; }

// floats of workspace a split reduction of this problem wants (0: none - no split, or not a large-tile problem)
int64_t satt_gemm_tile_ws_floats(const satt_gemm_params& p) {
  if (rk_eligible(p)) {
    const bool sum = p.splitk > 1 || (p.bank_ng > 0 && p.bank_c_col == 0);
    const int nslab = bank_kernel_form(p) == 2 ? p.bank_ng : rk_nz(p);          // (one slab per width on conv_bank_fwd_k)
    return (sum && p.N % 4 == 0 && p.nb_outer * p.nb_inner == 1) ? (int64_t)nslab * p.M * p.N : 0;
  }
  if (dw_eligible(p) && p.splitk > 1) {
    const int64_t rows = p.bank_ng > 0 ? (int64_t)p.conv_C * (p.bank_ng * (p.bank_ng + 1) / 2) : p.M;
    return (int64_t)p.splitk * rows * p.N;
  }
  return 0;
}

extern "C" int satt_shadow_pack(const float* flat, const int64_t* table, int nweights, uint16_t* st, uint16_t* sn,
                                void* stream) {
  if (nweights <= 0) return SATT_OK;
  if (!flat || !table || !st || !sn) return SATT_E_BADARG;
  hipLaunchKernelGGL(shadow_pack_k, dim3(64, nweights), dim3(256), 0, (hipStream_t)stream, flat, table, st, sn);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}
