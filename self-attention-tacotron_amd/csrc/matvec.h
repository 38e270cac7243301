// Workgroup-wide bf16 mat-vec used by the persistent recurrent kernels:
//   y[n] = sum_k x[k] * W[k][n]      W: bf16 row-major [K][N] in global memory (L2-resident), N % 8 == 0
// Thread (ks, cg) owns 8 consecutive columns cg*8.. and rows ks, ks+KS, ...; each wave-load moves 64 x 16 B
// contiguous bytes of one weight row.  Partials go through LDS.  All NT threads must call it.
#pragma once
#include "common.h"

template <int NT>
__device__ __forceinline__ void matvec_bf16(const float* __restrict__ x_lds, const uint16_t* __restrict__ W, int K,
                                            int N, float* __restrict__ partial, float* __restrict__ y_lds) {
  const int tid = threadIdx.x;
  const int CG = N >> 3;
  const int KS = NT / CG;  // >= 1 (N <= 8*NT)
  const int cg = tid % CG, ks = tid / CG;
  if (ks < KS) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    const uint16_t* wp = W + (size_t)cg * 8;
    int k = ks;
    for (; k + 3 * KS < K; k += 4 * KS) {
      uint4 w0 = *reinterpret_cast<const uint4*>(wp + (size_t)k * N);
      uint4 w1 = *reinterpret_cast<const uint4*>(wp + (size_t)(k + KS) * N);
      uint4 w2 = *reinterpret_cast<const uint4*>(wp + (size_t)(k + 2 * KS) * N);
      uint4 w3 = *reinterpret_cast<const uint4*>(wp + (size_t)(k + 3 * KS) * N);
      const float x0 = x_lds[k], x1 = x_lds[k + KS], x2 = x_lds[k + 2 * KS], x3 = x_lds[k + 3 * KS];
#define SATT_MV_FMA(WV_, XV_)                                                                      \
  acc[0] += XV_ * __uint_as_float((WV_).x << 16); acc[1] += XV_ * __uint_as_float((WV_).x & 0xFFFF0000u); \
  acc[2] += XV_ * __uint_as_float((WV_).y << 16); acc[3] += XV_ * __uint_as_float((WV_).y & 0xFFFF0000u); \
  acc[4] += XV_ * __uint_as_float((WV_).z << 16); acc[5] += XV_ * __uint_as_float((WV_).z & 0xFFFF0000u); \
  acc[6] += XV_ * __uint_as_float((WV_).w << 16); acc[7] += XV_ * __uint_as_float((WV_).w & 0xFFFF0000u);
      SATT_MV_FMA(w0, x0) SATT_MV_FMA(w1, x1) SATT_MV_FMA(w2, x2) SATT_MV_FMA(w3, x3)
    }
    for (; k < K; k += KS) {
      uint4 w0 = *reinterpret_cast<const uint4*>(wp + (size_t)k * N);
      const float x0 = x_lds[k];
      SATT_MV_FMA(w0, x0)
    }
#undef SATT_MV_FMA
    float4* pp = reinterpret_cast<float4*>(partial + (size_t)ks * N + cg * 8);
    pp[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    pp[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
  __syncthreads();
  for (int n = tid; n < N; n += NT) {
    float s = 0.f;
    for (int q = 0; q < KS; ++q) s += partial[(size_t)q * N + n];
    y_lds[n] = s;
  }
  __syncthreads();
}
