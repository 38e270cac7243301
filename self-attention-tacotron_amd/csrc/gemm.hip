// Generic MFMA GEMM for gfx950 with virtual-im2col operand modes and a fused epilogue.
// 64x64 output tile per 256-thread workgroup (2x2 waves, each wave 2x2 MFMA 16x16 tiles), register-prefetched
// global->LDS staging.  PREC_BF16: operands rounded to bf16 on the way into LDS, v_mfma_f32_16x16x32_bf16;
// PREC_F32: exact fp32 via v_mfma_f32_16x16x4_f32 (parity mode).
#include "common.h"

namespace {

template <int PREC> struct Cfg;
template <> struct Cfg<SATT_PREC_BF16> { typedef uint16_t LT; static constexpr int BK = 32, STRIDE = 40; };
template <> struct Cfg<SATT_PREC_F32> { typedef float LT; static constexpr int BK = 16, STRIDE = 17; };

template <int PREC> __device__ __forceinline__ typename Cfg<PREC>::LT cvt(float v);
template <> __device__ __forceinline__ uint16_t cvt<SATT_PREC_BF16>(float v) { return f2bf(v); }
template <> __device__ __forceinline__ float cvt<SATT_PREC_F32>(float v) { return v; }

constexpr int BM = 64, BN = 64, NT = 256;

template <int PREC, int A_MODE, bool B_NCONTIG>
__global__ __launch_bounds__(NT) void gemm_kernel(const satt_gemm_params p) {
  typedef typename Cfg<PREC>::LT LT;
  constexpr int BK = Cfg<PREC>::BK, STRIDE = Cfg<PREC>::STRIDE;
  constexpr int NE = BM * BK / NT;  // elements per thread per operand tile
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
  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    int chunk = (p.K + p.splitk - 1) / p.splitk;
    chunk = (chunk + BK - 1) / BK * BK;
    kbeg = zk * chunk;
    kend = min(p.K, kbeg + chunk);
  }

  // per-thread fixed decomposition of the rows this thread stages
  int rowb[NE], rowt[NE];
  int mtap = 0, mc = 0;
  if (A_MODE == 2) {
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      int m = m0 + (tid + i * NT) / BK;
      rowb[i] = m / p.conv_T; rowt[i] = m - rowb[i] * p.conv_T;
    }
  }
  if (A_MODE == 3) {
    int m = m0 + (tid % BM);
    mtap = m / p.conv_C; mc = m - mtap * p.conv_C;
  }

  float ra[NE], rb[NE];
  auto fetch = [&](int k0) {
    int atap0 = 0, ac0 = 0, ab0 = 0, at0 = 0;
    if (A_MODE == 2) { atap0 = k0 / p.conv_C; ac0 = k0 - atap0 * p.conv_C; }
    if (A_MODE == 3) { ab0 = k0 / p.conv_T; at0 = k0 - ab0 * p.conv_T; }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + i * NT;
      int mm, kk;
      if (A_KCONTIG) { kk = e % BK; mm = e / BK; } else { mm = e % BM; kk = e / BM; }
      const int m = m0 + mm, k = k0 + kk;
      float v = 0.f;
      if (m < p.M && k < kend) {
        if (A_MODE == 0) v = A[(int64_t)m * p.lda + k];
        if (A_MODE == 1) v = A[(int64_t)k * p.lda + m];
        if (A_MODE == 2) {
          int c = ac0 + kk, tap = atap0;
          while (c >= p.conv_C) { c -= p.conv_C; ++tap; }
          const int tt = rowt[i] + p.conv_sgn * tap + p.conv_off;
          if (tt >= 0 && tt < p.conv_T) v = A[((int64_t)rowb[i] * p.conv_T + tt) * p.lda + c];
        }
        if (A_MODE == 3) {
          int t = at0 + kk, b = ab0;
          while (t >= p.conv_T) { t -= p.conv_T; ++b; }
          const int tt = t + p.conv_sgn * mtap + p.conv_off;
          if (tt >= 0 && tt < p.conv_T) v = A[((int64_t)b * p.conv_T + tt) * p.lda + mc];
        }
      }
      ra[i] = v;
    }
    const int btap0 = k0 / p.kin, br0 = k0 - btap0 * p.kin;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + i * NT;
      int nn, kk;
      if (B_NCONTIG) { nn = e % BN; kk = e / BN; } else { kk = e % BK; nn = e / BK; }
      const int n = n0 + nn, k = k0 + kk;
      float v = 0.f;
      if (n < p.N && k < kend) {
        int r = br0 + kk, tap = btap0;
        while (r >= p.kin) { r -= p.kin; ++tap; }
        v = B[(int64_t)tap * p.sb_tap + (int64_t)r * p.sb_k + (int64_t)n * p.sb_n];
      }
      rb[i] = v;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + i * NT;
      int mm, kk;
      if (A_KCONTIG) { kk = e % BK; mm = e / BK; } else { mm = e % BM; kk = e / BM; }
      As[mm * STRIDE + kk] = cvt<PREC>(ra[i]);
      int nn;
      if (B_NCONTIG) { nn = e % BN; kk = e / BN; } else { kk = e % BK; nn = e / BK; }
      Bs[nn * STRIDE + kk] = cvt<PREC>(rb[i]);
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
    if constexpr (PREC == SATT_PREC_BF16) {
      bf16x8_t a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const bf16x8_t*>(&As[(wm * 32 + i * 16 + (lane & 15)) * STRIDE + (lane >> 4) * 8]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[j] = *reinterpret_cast<const bf16x8_t*>(&Bs[(wn * 32 + j * 16 + (lane & 15)) * STRIDE + (lane >> 4) * 8]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
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
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
        const int col = n0 + wn * 32 + j * 16 + (lane & 15);
        if (row < p.M && col < p.N) {
          float v = p.alpha * acc[i][j][r];
          float* dst = C + (int64_t)row * p.ldc + col;
          if (p.splitk > 1) {
            atomicAdd(dst, v);
          } else {
            if (p.bias) v += p.bias[col];
            if (p.act == SATT_ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.act == SATT_ACT_TANH) v = tanhf(v);
            else if (p.act == SATT_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
            if (p.drop_thresh != 0)
              v = satt_keep(seed, p.drop_stream, (uint32_t)row * (uint32_t)p.N + (uint32_t)col, p.drop_thresh)
                      ? v * p.drop_scale : 0.f;
            if (p.residual) v += p.residual[(int64_t)row * p.ldr + col];
            if (p.accumulate) v += *dst;
            *dst = v;
          }
        }
      }
}

template <int PREC, int A_MODE>
void launch2(const satt_gemm_params& p, dim3 grid, hipStream_t s) {
  if (p.sb_n == 1) hipLaunchKernelGGL((gemm_kernel<PREC, A_MODE, true>), grid, dim3(NT), 0, s, p);
  else hipLaunchKernelGGL((gemm_kernel<PREC, A_MODE, false>), grid, dim3(NT), 0, s, p);
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

extern "C" int satt_gemm(const satt_gemm_params* pp, void* stream) {
  if (!pp) return SATT_E_BADARG;
  satt_gemm_params p = *pp;
  if (p.M <= 0 || p.N <= 0) return SATT_OK;
  if (p.K < 0 || !p.A || !p.B || !p.C) return SATT_E_BADARG;
  if (p.a_mode < 0 || p.a_mode > 3) return SATT_E_BADARG;
  if ((p.a_mode >= 2) && (p.conv_T <= 0 || p.conv_C <= 0)) return SATT_E_BADARG;
  if (p.nb_outer <= 0) p.nb_outer = 1;
  if (p.nb_inner <= 0) p.nb_inner = 1;
  if (p.splitk <= 0) p.splitk = 1;
  if (p.kin <= 0) p.kin = p.K > 0 ? p.K : 1;
  if (p.splitk > 1 && (!p.accumulate || p.bias || p.act || p.residual || p.drop_thresh)) return SATT_E_BADARG;
  if (p.precision != SATT_PREC_F32 && p.precision != SATT_PREC_BF16) return SATT_E_BADARG;
  dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.nb_outer * p.nb_inner * p.splitk);
  if (grid.y > 65535 || grid.z > 65535) return SATT_E_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (p.precision == SATT_PREC_BF16) launch1<SATT_PREC_BF16>(p, grid, s);
  else launch1<SATT_PREC_F32>(p, grid, s);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}
