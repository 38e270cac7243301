// Generic MFMA GEMM for gfx950 with virtual-im2col operand modes and a fused epilogue.
// 64x64 output tile per 256-thread workgroup (2x2 waves, each wave 2x2 MFMA 16x16 tiles), register-prefetched
// global->LDS staging.  PREC_BF16: operands rounded to bf16 on the way into LDS, v_mfma_f32_16x16x32_bf16;
// PREC_F32: exact fp32 via v_mfma_f32_16x16x4_f32 (parity mode).
#include <cstdlib>
#include "common.h"
#include "gemm_tile.h"

namespace {

template <int PREC> struct Cfg;
template <> struct Cfg<SATT_PREC_BF16> { typedef uint16_t LT; static constexpr int BK = 32, STRIDE = 40; };
// DEEP: a launch that puts about one workgroup on a CU has nothing to hide the global-load round trip of a K tile
// behind (the next tile is requested one iteration ahead): stage SATT_GEMM_DEEP_BK elements per iteration instead.
#ifndef SATT_GEMM_DEEP_BK
#define SATT_GEMM_DEEP_BK 64
#endif
constexpr int PREC_BF16_DEEP = 2;
template <> struct Cfg<PREC_BF16_DEEP> { typedef uint16_t LT; static constexpr int BK = SATT_GEMM_DEEP_BK, STRIDE = SATT_GEMM_DEEP_BK + 8; };
template <> struct Cfg<SATT_PREC_F32> { typedef float LT; static constexpr int BK = 16, STRIDE = 17; };

template <int PREC> __device__ __forceinline__ typename Cfg<PREC>::LT cvt(float v);
template <> __device__ __forceinline__ uint16_t cvt<SATT_PREC_BF16>(float v) { return f2bf(v); }
template <> __device__ __forceinline__ uint16_t cvt<PREC_BF16_DEEP>(float v) { return f2bf(v); }
template <> __device__ __forceinline__ float cvt<SATT_PREC_F32>(float v) { return v; }

constexpr int BM = 64, BN = 64, NT = 256;

// VEC: every operand access is a 16-byte float4 along its contiguous dimension (requires lda/ldb/strides % 4 == 0,
// 16 B aligned bases, conv_C % 4 == 0); index math is done once per 4 elements.  !VEC: scalar generic loads.
// PLAIN (requires VEC): plain operand addressing (A_MODE 0/1, one B tap: kin >= K, no conv bank) - every staged group
// keeps a running source pointer, so the main loop has no index arithmetic (the generic path spends ~290 instructions
// per iteration on tap/row decomposition and 64-bit address math for 4 MFMAs: it is instruction-issue bound).
template <int PREC, int A_MODE, bool B_NCONTIG, bool VEC, bool PLAIN = false>
__global__ __launch_bounds__(NT) void gemm_kernel(const satt_gemm_params p) {
  typedef typename Cfg<PREC>::LT LT;
  constexpr int BK = Cfg<PREC>::BK, STRIDE = Cfg<PREC>::STRIDE;
  constexpr int NE = BM * BK / NT;  // elements per thread per operand tile
  constexpr int NV = NE / 4;        // float4 groups per thread per operand tile
  constexpr bool A_KCONTIG = (A_MODE == 0 || A_MODE == 2);
  __shared__ __attribute__((aligned(16))) LT As[BM * STRIDE];
  __shared__ __attribute__((aligned(16))) LT Bs[BN * STRIDE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int z = blockIdx.z;
  const int zk = z % p.splitk; z /= p.splitk;
  const int zo = z / p.nb_inner, zi = z - zo * p.nb_inner;
  const float* __restrict__ A = p.A + zo * p.strideA_o + zi * p.strideA_i;
  const float* __restrict__ B = p.B + zo * p.strideB_o + zi * p.strideB_i;
  float* __restrict__ C = p.C + zo * p.strideC_o + zi * p.strideC_i;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  int Kz = p.K, conv_off = p.conv_off;
  bool atomic_out = p.splitk > 1;
  if (A_MODE == 2 && p.bank_ng > 0) {        // conv bank: blockIdx.z is the group, widest (longest K) first
    const int g = p.bank_ng - 1 - (int)blockIdx.z;
    Kz = (g + 1) * p.conv_C;
    conv_off = -p.conv_sgn * (g / 2);
    A = p.A + (int64_t)g * p.bank_a_col;
    B = p.B + p.bank_b_unit * (int64_t)(g * (g + 1) / 2);
    C = p.C + (int64_t)g * p.bank_c_col;
    atomic_out = p.bank_c_col == 0;
  }
  int kbeg = 0, kend = Kz;
  if (p.splitk > 1) {
    int chunk = (p.K + p.splitk - 1) / p.splitk;
    chunk = (chunk + BK - 1) / BK * BK;
    kbeg = zk * chunk;
    kend = min(p.K, kbeg + chunk);
  }

  // thread -> (row/col, k) of staged group g: k-contiguous sources put 4 consecutive k in one group,
  // m/n-contiguous sources put 4 consecutive rows / columns in one group
  constexpr int GK = BK / 4;        // groups along k for k-contiguous sources
  constexpr int GM = BM / 4;        // groups along m (or n) for m/n-contiguous sources
  // TRA / TRB: a bf16 operand whose source is contiguous along m / n is transposed on its way into LDS.  Writing the
  // four rows of a float4 group as 2-byte stores at a 4-row lane pitch hits 2 of the 32 write banks (8-way conflict:
  // SQ_LDS_BANK_CONFLICT was ~80 % of the LDS-active cycles of these kernels).  Instead a thread owns one float4 at
  // TWO adjacent k (groups 2h, 2h+1), 16 consecutive lanes walk the 16 k pairs of a tile, and each row gets one
  // 4-byte store of the (k, k+1) pair: a 32-lane group then covers all 32 banks.
  constexpr bool TRA = VEC && !A_KCONTIG && (PREC != SATT_PREC_F32);
  constexpr bool TRB = VEC && B_NCONTIG && (PREC != SATT_PREC_F32);
  auto a_pos = [&](int g, int& mm, int& kk) {
    if (TRA) {
      mm = (tid >> 4) * 4; kk = 2 * (tid & 15) + (g & 1) + 32 * (g >> 1);
    } else if (VEC) {
      const int e = tid + g * NT;
      if (A_KCONTIG) { kk = (e % GK) * 4; mm = e / GK; } else { mm = (e % GM) * 4; kk = e / GM; }
    } else {
      const int e = tid + g * NT;
      if (A_KCONTIG) { kk = e % BK; mm = e / BK; } else { mm = e % BM; kk = e / BM; }
    }
  };
  auto b_pos = [&](int g, int& nn, int& kk) {
    if (TRB) {
      nn = (tid >> 4) * 4; kk = 2 * (tid & 15) + (g & 1) + 32 * (g >> 1);
    } else if (VEC) {
      const int e = tid + g * NT;
      if (B_NCONTIG) { nn = (e % GM) * 4; kk = e / GM; } else { kk = (e % GK) * 4; nn = e / GK; }
    } else {
      const int e = tid + g * NT;
      if (B_NCONTIG) { nn = e % BN; kk = e / BN; } else { kk = e % BK; nn = e / BK; }
    }
  };
  constexpr int NG = VEC ? NV : NE;   // staged groups per thread per operand
  constexpr int GW = VEC ? 4 : 1;     // elements per group

  // per-thread fixed decomposition of the rows this thread stages
  int rowb[NG], rowt[NG];
  int mtap = 0, mc = 0;
  if (A_MODE == 2) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      int mm, kk; a_pos(g, mm, kk);
      const int m = m0 + mm;
      rowb[g] = m / p.conv_T; rowt[g] = m - rowb[g] * p.conv_T;
    }
  }
  if (A_MODE == 3) {
    int mm, kk; a_pos(0, mm, kk);          // mm is the same for every group of this thread
    const int m = m0 + mm;
    mtap = m / p.conv_C; mc = m - mtap * p.conv_C;
  }

  float ra[NG * GW], rb[NG * GW];
  // PLAIN: running pointers of the staged groups (at k0 = kbeg) and their loop-invariant validity
  const float* pa[PLAIN ? NG : 1]; const float* pb[PLAIN ? NG : 1];
  bool aok[PLAIN ? NG : 1], bok[PLAIN ? NG : 1];
  int64_t a_step = 0, b_step = 0;
  // conv fast path (PLAIN with A_MODE 2; conv_C % BK == 0, kin % BK == 0): the tap of a whole K tile is uniform, so
  // the tap / channel split is scalar bookkeeping once per iteration and every group adds one scalar offset to its
  // row pointer.  (atap, ac0) / (btap, br0): tap and first channel / B row of the tile being fetched.
  int atap = 0, ac0 = 0, btap = 0, br0 = 0;
  if constexpr (PLAIN && A_MODE == 2) {
    atap = kbeg / p.conv_C; ac0 = kbeg - atap * p.conv_C;
    btap = kbeg / p.kin; br0 = kbeg - btap * p.kin;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      int mm, kk; a_pos(g, mm, kk);
      aok[g] = m0 + mm < p.M;
      pa[g] = A + ((int64_t)rowb[g] * p.conv_T + rowt[g]) * p.lda + kk;
      int nn; b_pos(g, nn, kk);
      bok[g] = n0 + nn < p.N;
      pb[g] = B + (int64_t)kk * p.sb_k + (int64_t)(n0 + nn) * p.sb_n;
    }
  }
  if constexpr (PLAIN && A_MODE <= 1) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      int mm, kk; a_pos(g, mm, kk);
      aok[g] = m0 + mm < p.M;
      pa[g] = A_MODE == 0 ? A + (int64_t)(m0 + mm) * p.lda + (kbeg + kk) : A + (int64_t)(kbeg + kk) * p.lda + (m0 + mm);
      int nn; b_pos(g, nn, kk);
      bok[g] = n0 + nn < p.N;
      pb[g] = B + (int64_t)(kbeg + kk) * p.sb_k + (int64_t)(n0 + nn) * p.sb_n;
    }
    a_step = A_MODE == 0 ? (int64_t)BK : (int64_t)BK * p.lda;
    b_step = (int64_t)BK * p.sb_k;
  }
  auto fetch = [&](int k0) {
    if constexpr (PLAIN && A_MODE == 2) {       // called with k0 = kbeg, kbeg + BK, ... in order
      const int shift = p.conv_sgn * atap + conv_off;
      const int64_t aoff = (int64_t)shift * p.lda + ac0;
      const int64_t boff = (int64_t)btap * p.sb_tap + (int64_t)br0 * p.sb_k;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        int mm, kk; a_pos(g, mm, kk);
        const int tt = rowt[g] + shift;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (aok[g] && k0 + kk < kend && tt >= 0 && tt < p.conv_T) v = *reinterpret_cast<const float4*>(pa[g] + aoff);
        ra[g * 4 + 0] = v.x; ra[g * 4 + 1] = v.y; ra[g * 4 + 2] = v.z; ra[g * 4 + 3] = v.w;
        int nn; b_pos(g, nn, kk);
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bok[g] && k0 + kk < kend) w = *reinterpret_cast<const float4*>(pb[g] + boff);
        rb[g * 4 + 0] = w.x; rb[g * 4 + 1] = w.y; rb[g * 4 + 2] = w.z; rb[g * 4 + 3] = w.w;
      }
      ac0 += BK; if (ac0 >= p.conv_C) { ac0 = 0; ++atap; }
      br0 += BK; if (br0 >= p.kin) { br0 = 0; ++btap; }
      return;
    }
    if constexpr (PLAIN && A_MODE <= 1) {       // called with k0 = kbeg, kbeg + BK, ... in order
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        int mm, kk; a_pos(g, mm, kk);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (aok[g] && k0 + kk < kend) v = *reinterpret_cast<const float4*>(pa[g]);
        ra[g * 4 + 0] = v.x; ra[g * 4 + 1] = v.y; ra[g * 4 + 2] = v.z; ra[g * 4 + 3] = v.w;
        pa[g] += a_step;
        int nn; b_pos(g, nn, kk);
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bok[g] && k0 + kk < kend) w = *reinterpret_cast<const float4*>(pb[g]);
        rb[g * 4 + 0] = w.x; rb[g * 4 + 1] = w.y; rb[g * 4 + 2] = w.z; rb[g * 4 + 3] = w.w;
        pb[g] += b_step;
      }
      return;
    }
    int atap0 = 0, ac0 = 0, ab0 = 0, at0 = 0;
    if (A_MODE == 2) { atap0 = k0 / p.conv_C; ac0 = k0 - atap0 * p.conv_C; }
    if (A_MODE == 3) { ab0 = k0 / p.conv_T; at0 = k0 - ab0 * p.conv_T; }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      int mm, kk; a_pos(g, mm, kk);
      const int m = m0 + mm, k = k0 + kk;
#pragma unroll
      for (int j = 0; j < GW; ++j) ra[g * GW + j] = 0.f;
      if (m < p.M && k < kend) {
        const float* src = nullptr;
        if (A_MODE == 0) src = A + (int64_t)m * p.lda + k;
        if (A_MODE == 1) src = A + (int64_t)k * p.lda + m;
        if (A_MODE == 2) {
          int c = ac0 + kk, tap = atap0;
          while (c >= p.conv_C) { c -= p.conv_C; ++tap; }
          const int tt = rowt[g] + p.conv_sgn * tap + conv_off;
          if (tt >= 0 && tt < p.conv_T) src = A + ((int64_t)rowb[g] * p.conv_T + tt) * p.lda + c;
        }
        if (A_MODE == 3) {
          int t = at0 + kk, b = ab0;
          while (t >= p.conv_T) { t -= p.conv_T; ++b; }
          const int tt = t + p.conv_sgn * mtap + p.conv_off;
          if (tt >= 0 && tt < p.conv_T) src = A + ((int64_t)b * p.conv_T + tt) * p.lda + mc;
        }
        if (src) {
          if (VEC) {   // host guarantees M % 4 == 0 / K % 4 == 0 along the vector dimension
            const float4 v = *reinterpret_cast<const float4*>(src);
            ra[g * 4 + 0] = v.x; ra[g * 4 + 1] = v.y; ra[g * 4 + 2] = v.z; ra[g * 4 + 3] = v.w;
          } else {
            ra[g] = *src;
          }
        }
      }
    }
    const int btap0 = k0 / p.kin, br0 = k0 - btap0 * p.kin;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      int nn, kk; b_pos(g, nn, kk);
      const int n = n0 + nn, k = k0 + kk;
#pragma unroll
      for (int j = 0; j < GW; ++j) rb[g * GW + j] = 0.f;
      if (n < p.N && k < kend) {
        int r = br0 + kk, tap = btap0;
        while (r >= p.kin) { r -= p.kin; ++tap; }
        const float* src = B + (int64_t)tap * p.sb_tap + (int64_t)r * p.sb_k + (int64_t)n * p.sb_n;
        if (VEC) {
          const float4 v = *reinterpret_cast<const float4*>(src);
          rb[g * 4 + 0] = v.x; rb[g * 4 + 1] = v.y; rb[g * 4 + 2] = v.z; rb[g * 4 + 3] = v.w;
        } else {
          rb[g] = *src;
        }
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      int mm, kk; a_pos(g, mm, kk);
      if (VEC && A_KCONTIG) {
        if constexpr (PREC != SATT_PREC_F32) {
          uint2 w;
          w.x = pack_bf16x2(ra[g * 4 + 0], ra[g * 4 + 1]);
          w.y = pack_bf16x2(ra[g * 4 + 2], ra[g * 4 + 3]);
          *reinterpret_cast<uint2*>(&As[mm * STRIDE + kk]) = w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) As[mm * STRIDE + kk + j] = cvt<PREC>(ra[g * 4 + j]);
        }
      } else if constexpr (TRA) {
        if ((g & 1) == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            *reinterpret_cast<uint32_t*>(&As[(mm + j) * STRIDE + kk]) = pack_bf16x2(ra[g * 4 + j], ra[(g + 1) * 4 + j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < GW; ++j) As[(mm + j) * STRIDE + kk] = cvt<PREC>(ra[g * GW + j]);
      }
      int nn; b_pos(g, nn, kk);
      if (VEC && !B_NCONTIG) {
        if constexpr (PREC != SATT_PREC_F32) {
          uint2 w;
          w.x = pack_bf16x2(rb[g * 4 + 0], rb[g * 4 + 1]);
          w.y = pack_bf16x2(rb[g * 4 + 2], rb[g * 4 + 3]);
          *reinterpret_cast<uint2*>(&Bs[nn * STRIDE + kk]) = w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) Bs[nn * STRIDE + kk + j] = cvt<PREC>(rb[g * 4 + j]);
        }
      } else if constexpr (TRB) {
        if ((g & 1) == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            *reinterpret_cast<uint32_t*>(&Bs[(nn + j) * STRIDE + kk]) = pack_bf16x2(rb[g * 4 + j], rb[(g + 1) * 4 + j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < GW; ++j) Bs[(nn + j) * STRIDE + kk] = cvt<PREC>(rb[g * GW + j]);
      }
    }
  };

  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  if (kbeg < kend) fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    stage();
    __syncthreads();
    if (k0 + BK < kend) fetch(k0 + BK);
    if constexpr (PREC != SATT_PREC_F32) {
#pragma unroll
      for (int ks = 0; ks < BK; ks += 32) {
        bf16x8_t a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a[i] = *reinterpret_cast<const bf16x8_t*>(&As[(wm * 32 + i * 16 + (lane & 15)) * STRIDE + ks + (lane >> 4) * 8]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          b[j] = *reinterpret_cast<const bf16x8_t*>(&Bs[(wn * 32 + j * 16 + (lane & 15)) * STRIDE + ks + (lane >> 4) * 8]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        float a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = As[(wm * 32 + i * 16 + (lane & 15)) * STRIDE + kk * 4 + (lane >> 4)];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = Bs[(wn * 32 + j * 16 + (lane & 15)) * STRIDE + kk * 4 + (lane >> 4)];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  const uint32_t seed = (p.drop_thresh != 0 && p.seed) ? *p.seed : 0u;
  // Epilogue operands first (bias per column, residual / previous C per element, clamped addresses), then arithmetic and
  // stores with no wait in between: a load beside each store compiles to load - s_waitcnt vmcnt(0) - store per element, and
  // vmcnt also counts the stores in flight (16 serial round trips per thread; csrc/gemm_tile.hip has the same note)
  float bcol[2] = {0.f, 0.f}, exr[2][2][4], exa[2][2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { exr[i][j][r] = 0.f; exa[i][j][r] = 0.f; }
  if (!atomic_out && p.bias) {
#pragma unroll
    for (int j = 0; j < 2; ++j) bcol[j] = p.bias[min(n0 + wn * 32 + j * 16 + (lane & 15), p.N - 1)];
  }
  if (!atomic_out && p.residual) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = min(m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r, p.M - 1);
          const int col = min(n0 + wn * 32 + j * 16 + (lane & 15), p.N - 1);
          exr[i][j][r] = p.residual[(int64_t)row * p.ldr + col];
        }
  }
  if (!atomic_out && p.accumulate) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = min(m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r, p.M - 1);
          const int col = min(n0 + wn * 32 + j * 16 + (lane & 15), p.N - 1);
          exa[i][j][r] = C[(int64_t)row * p.ldc + col];
        }
  }
  if (atomic_out) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
          const int col = n0 + wn * 32 + j * 16 + (lane & 15);
          if (row < p.M && col < p.N) atomicAdd(C + (int64_t)row * p.ldc + col, p.alpha * acc[i][j][r]);
        }
    return;
  }
  // every value final in registers (all loads consumed) before the first store
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
        const int col = n0 + wn * 32 + j * 16 + (lane & 15);
        float v = p.alpha * acc[i][j][r] + bcol[j];
        if (p.act == SATT_ACT_RELU) v = fmaxf(v, 0.f);
        else if (p.act == SATT_ACT_TANH) v = tanhf(v);
        else if (p.act == SATT_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
        else if (p.act == SATT_ACT_SOFTSIGN) v = v / (1.f + fabsf(v));
        if (p.drop_thresh != 0)
          v = satt_keep(seed, p.drop_stream, (uint32_t)row * (uint32_t)p.N + (uint32_t)col, p.drop_thresh)
                  ? v * p.drop_scale : 0.f;
        acc[i][j][r] = (v + exr[i][j][r]) + exa[i][j][r];
      }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
        const int col = n0 + wn * 32 + j * 16 + (lane & 15);
        if (row < p.M && col < p.N) C[(int64_t)row * p.ldc + col] = acc[i][j][r];
      }
}

// host-side check that every access of the problem can be a 16-byte vector
inline bool can_vec(const satt_gemm_params& p) {
  auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!a16(p.A) || !a16(p.B)) return false;
  if (p.lda % 4 || p.strideA_o % 4 || p.strideA_i % 4 || p.strideB_o % 4 || p.strideB_i % 4) return false;
  const bool a_k = (p.a_mode == 0 || p.a_mode == 2);
  if (a_k) { if (p.K % 4) return false; } else { if (p.M % 4) return false; }
  if (p.a_mode == 2 && p.conv_C % 4) return false;
  if (p.bank_ng > 0 && (p.bank_a_col % 4 || p.bank_b_unit % 4)) return false;
  if (p.a_mode == 3 && p.conv_C % 4 && p.conv_C != 1) return false;
  if (p.a_mode == 3 && p.conv_C == 1) return false;        // 1-channel weight-gradient form: scalar path
  if (p.splitk > 1) {                                      // split boundaries are multiples of BK -> fine
  }
  if (p.sb_n == 1) {               // n-contiguous B: vectors along n
    if (p.N % 4 || p.sb_k % 4 || p.sb_tap % 4) return false;
  } else if (p.sb_k == 1) {        // k-contiguous B: vectors along k
    if (p.K % 4 || p.kin % 4 || p.sb_n % 4 || p.sb_tap % 4) return false;
  } else {
    return false;
  }
  return true;
}

template <int PREC, int A_MODE>
void launch2(const satt_gemm_params& p, dim3 grid, hipStream_t s) {
  const bool v = can_vec(p);
  if constexpr (A_MODE <= 1) {
    if (v && p.kin >= p.K && p.bank_ng == 0) {
      if constexpr (PREC == SATT_PREC_BF16) {
        static const int deep_max = [] { const char* e = getenv("SATT_GEMM_DEEP_MAX"); return e ? atoi(e) : 384; }();
        const int kper = (p.K + p.splitk - 1) / p.splitk;
        if ((int64_t)grid.x * grid.y * grid.z <= deep_max && kper >= 2 * SATT_GEMM_DEEP_BK) {
          if (p.sb_n == 1) hipLaunchKernelGGL((gemm_kernel<PREC_BF16_DEEP, A_MODE, true, true, true>), grid, dim3(NT), 0, s, p);
          else hipLaunchKernelGGL((gemm_kernel<PREC_BF16_DEEP, A_MODE, false, true, true>), grid, dim3(NT), 0, s, p);
          return;
        }
      }
      if (p.sb_n == 1) hipLaunchKernelGGL((gemm_kernel<PREC, A_MODE, true, true, true>), grid, dim3(NT), 0, s, p);
      else hipLaunchKernelGGL((gemm_kernel<PREC, A_MODE, false, true, true>), grid, dim3(NT), 0, s, p);
      return;
    }
  }
  if constexpr (A_MODE == 2) {
    constexpr int BKc = Cfg<PREC>::BK;
    if (v && p.conv_C % BKc == 0 && p.kin % BKc == 0) {
      if (p.sb_n == 1) hipLaunchKernelGGL((gemm_kernel<PREC, 2, true, true, true>), grid, dim3(NT), 0, s, p);
      else hipLaunchKernelGGL((gemm_kernel<PREC, 2, false, true, true>), grid, dim3(NT), 0, s, p);
      return;
    }
  }
  if (p.sb_n == 1) {
    if (v) hipLaunchKernelGGL((gemm_kernel<PREC, A_MODE, true, true>), grid, dim3(NT), 0, s, p);
    else hipLaunchKernelGGL((gemm_kernel<PREC, A_MODE, true, false>), grid, dim3(NT), 0, s, p);
  } else {
    if (v) hipLaunchKernelGGL((gemm_kernel<PREC, A_MODE, false, true>), grid, dim3(NT), 0, s, p);
    else hipLaunchKernelGGL((gemm_kernel<PREC, A_MODE, false, false>), grid, dim3(NT), 0, s, p);
  }
}
template <int PREC>
void launch1(const satt_gemm_params& p, dim3 grid, hipStream_t s) {
  switch (p.a_mode) {
    case 0: launch2<PREC, 0>(p, grid, s); break;
    case 1: launch2<PREC, 1>(p, grid, s); break;
    case 2: launch2<PREC, 2>(p, grid, s); break;
    default: launch2<PREC, 3>(p, grid, s); break;
  }
}

}  // namespace

// argument checks + defaults shared by satt_gemm and satt_gemm_path
static int gemm_prepare(const satt_gemm_params* pp, satt_gemm_params& p) {
  if (!pp) return SATT_E_BADARG;
  p = *pp;
  if (p.M <= 0 || p.N <= 0) return 1;          // nothing to do
  if (p.K < 0 || !p.A || !p.B || !p.C) return SATT_E_BADARG;
  if (p.a_mode < 0 || p.a_mode > 3) return SATT_E_BADARG;
  if ((p.a_mode >= 2) && (p.conv_T <= 0 || p.conv_C <= 0)) return SATT_E_BADARG;
  if (p.nb_outer <= 0) p.nb_outer = 1;
  if (p.nb_inner <= 0) p.nb_inner = 1;
  if (p.splitk <= 0) p.splitk = 1;
  if (p.kin <= 0) p.kin = p.K > 0 ? p.K : 1;
  if (p.splitk > 1 && (p.bias || p.act || p.residual || p.drop_thresh)) return SATT_E_BADARG;
  if (p.precision != SATT_PREC_F32 && p.precision != SATT_PREC_BF16) return SATT_E_BADARG;
  if (p.bank_ng < 0) return SATT_E_BADARG;
  if (p.bank_ng > 0) {
    if (p.a_mode == 3) {               // weight gradients of the whole bank in one launch: large-tile kernel only
      if (p.nb_outer * p.nb_inner != 1 || !p.accumulate) return SATT_E_BADARG;
    } else {
      if (p.a_mode != 2 || p.nb_outer * p.nb_inner != 1 || p.splitk != 1 || p.bank_b_unit < 0) return SATT_E_BADARG;
      if (p.bank_c_col == 0 && (!p.accumulate || p.bias || p.act || p.residual || p.drop_thresh)) return SATT_E_BADARG;
      p.K = p.bank_ng * p.conv_C;          // longest group; used by the vector-path check only
    }
  }
  if (p.colsum && ((p.a_mode != 1 && p.a_mode != 3) || p.sb_n != 1 || p.nb_outer * p.nb_inner != 1)) return SATT_E_BADARG;
  return SATT_OK;
}

extern "C" int satt_gemm(const satt_gemm_params* pp, void* stream) {
  satt_gemm_params p;
  const int prc = gemm_prepare(pp, p);
  if (prc) return prc > 0 ? SATT_OK : prc;
  hipStream_t s = (hipStream_t)stream;
  // large-tile bf16 families first (gemm_tile.hip); the generic kernel below takes whatever they decline
  if (satt_gemm_tile_rk(p, s) || satt_gemm_tile_dw(p, s)) {
    SATT_LAUNCH_CHECK();
    return SATT_OK;
  }
  if (p.bank_ng > 0 && p.a_mode == 3) return SATT_E_UNSUPPORTED;     // callers fall back to one call per width
  if (p.splitk > 1 && !p.accumulate) return SATT_E_BADARG;           // overwriting splits need the slab workspace
  if (p.colsum) {       // the generic kernel has no fused column sum: the bias gradient is its own launch
    const int rc = satt_colsum(p.B, p.sb_k, p.colsum, p.K, p.N, 1, stream);
    if (rc) return rc;
  }
  dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.bank_ng > 0 ? p.bank_ng : p.nb_outer * p.nb_inner * p.splitk);
  if (grid.y > 65535 || grid.z > 65535) return SATT_E_UNSUPPORTED;
  if (p.precision == SATT_PREC_BF16) launch1<SATT_PREC_BF16>(p, grid, s);
  else launch1<SATT_PREC_F32>(p, grid, s);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}

/* host-only: floats of workspace (p->ws) with which a split reduction of this problem avoids atomics; 0 if none applies */
extern "C" int64_t satt_gemm_ws_floats(const satt_gemm_params* pp) {
  satt_gemm_params p;
  const int prc = gemm_prepare(pp, p);
  if (prc) return 0;
  return satt_gemm_tile_ws_floats(p);
}

/* host-only: which kernel family satt_gemm would run this problem on (0 generic, 1 large-tile forward / dX, 2 large-tile
 * weight gradient); negative on bad arguments */
extern "C" int satt_gemm_path(const satt_gemm_params* pp) {
  satt_gemm_params p;
  const int prc = gemm_prepare(pp, p);
  if (prc) return prc > 0 ? 0 : prc;
  return satt_gemm_tile_path(p);
}
