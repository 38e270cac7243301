// Workgroup-wide bf16 mat-vec used by the persistent recurrent kernels:
//   y[n] = sum_k x[k] * W[k][n]      W: bf16 row-major [K][N] in global memory (L2-resident), N % 8 == 0
// Thread (ks, cg) owns 8 consecutive columns cg*8.. and rows ks, ks+KS, ...; each wave-load moves 64 x 16 B
// contiguous bytes of one weight row.  UN row-loads are kept in flight per thread and the next batch is issued
// before the current one is consumed (register double buffering), so the loop is bound by per-CU L2 bandwidth
// rather than by load latency.  Partials go through LDS.  All NT threads must call it.
#pragma once
#include "common.h"

__device__ __forceinline__ void mv_fma8(float (&acc)[8], const uint4& w, float xv) {
  acc[0] += xv * __uint_as_float(w.x << 16); acc[1] += xv * __uint_as_float(w.x & 0xFFFF0000u);
  acc[2] += xv * __uint_as_float(w.y << 16); acc[3] += xv * __uint_as_float(w.y & 0xFFFF0000u);
  acc[4] += xv * __uint_as_float(w.z << 16); acc[5] += xv * __uint_as_float(w.z & 0xFFFF0000u);
  acc[6] += xv * __uint_as_float(w.w << 16); acc[7] += xv * __uint_as_float(w.w & 0xFFFF0000u);
}

template <int NT, int UN>
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
    const int nrow = (K - ks + KS - 1) / KS;      // rows owned by this thread: ks + r*KS
    const int nfull = nrow / UN;
    uint4 cur[UN], nxt[UN];
    if (nfull > 0) {
#pragma unroll
      for (int u = 0; u < UN; ++u) cur[u] = *reinterpret_cast<const uint4*>(wp + (size_t)(ks + u * KS) * N);
    }
    for (int bt = 0; bt < nfull; ++bt) {
      const int r0 = bt * UN;
      if (bt + 1 < nfull) {
#pragma unroll
        for (int u = 0; u < UN; ++u)
          nxt[u] = *reinterpret_cast<const uint4*>(wp + (size_t)(ks + (r0 + UN + u) * KS) * N);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) mv_fma8(acc, cur[u], x_lds[ks + (r0 + u) * KS]);
      if (bt + 1 < nfull) {
#pragma unroll
        for (int u = 0; u < UN; ++u) cur[u] = nxt[u];
      }
    }
    for (int r = nfull * UN; r < nrow; ++r) {
      const uint4 w0 = *reinterpret_cast<const uint4*>(wp + (size_t)(ks + r * KS) * N);
      mv_fma8(acc, w0, x_lds[ks + r * KS]);
    }
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
