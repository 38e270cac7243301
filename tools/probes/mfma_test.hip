// micro-test: y[n] = sum_k x[k] W[k][n] via v_mfma_f32_16x16x32_bf16 with B pinned in AGPRs and x split 3-way
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
__host__ __device__ inline uint16_t f2bf(float f) { union { float f; uint32_t u; } c_; c_.f = f; uint32_t u = c_.u; u += 0x7FFFu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
__host__ __device__ inline float bf2f(uint16_t h) { union { float f; uint32_t u; } c_; c_.u = ((uint32_t)h) << 16; return c_.f; }
constexpr int K = 64, N = 16;
__global__ void k(const float* x, const uint16_t* W, float* y, int mode) {
  __shared__ uint16_t xs[4][K];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4 * K; i += 64) (&xs[0][0])[i] = 0;
  __syncthreads();
  if (lane < K) {
    float v = x[lane]; uint16_t h = f2bf(v); float r1 = v - bf2f(h); uint16_t m = f2bf(r1); float r2 = r1 - bf2f(m);
    xs[0][lane] = h; xs[1][lane] = m; xs[2][lane] = f2bf(r2);
  }
  __syncthreads();
  i32x4_t w[2];
  for (int kt = 0; kt < 2; ++kt) {
    i32x4_t t = {0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) {
      const int kk = kt * 32 + (lane >> 4) * 8 + i, n = lane & 15;
      t[i >> 1] |= (int)((uint32_t)W[kk * N + n] << ((i & 1) * 16));
    }
    asm volatile("" : "+a"(t));
    w[kt] = t;
  }
  f32x4_t acc = {0, 0, 0, 0};
  if (mode == 3) {
    const uint16_t* xr = &xs[0][0] + min(lane & 15, 3) * K + (lane >> 4) * 8;
    const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(xr), a1 = *reinterpret_cast<const bf16x8_t*>(xr + 32);
    asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %4, %0\n\ts_nop 7\n\ts_nop 7"
                 : "+v"(acc) : "v"(a0), "v"(a1), "a"(w[0]), "a"(w[1]));
    if (lane < 16) { y[lane] = acc[0] + acc[1] + acc[2]; y[16 + lane] = acc[0]; y[32 + lane] = acc[1]; y[48 + lane] = acc[2]; }
    return;
  }
  const uint16_t* xrow = &xs[0][0] + min(lane & 15, 3) * K + (lane >> 4) * 8;
  for (int kt = 0; kt < 2; ++kt) {
    const bf16x8_t av = *reinterpret_cast<const bf16x8_t*>(xrow + kt * 32);
    if (mode == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7" : "+v"(acc) : "v"(av), "a"(w[kt]));
    else if (mode == 2) {   // B produced by VALU immediately before the asm MFMA
      i32x4_t t = {0, 0, 0, 0};
      for (int i = 0; i < 8; ++i) {
        const int kk = kt * 32 + (lane >> 4) * 8 + i, n = lane & 15;
        t[i >> 1] |= (int)((uint32_t)W[kk * N + n] << ((i & 1) * 16));
      }
      t[0] ^= mode; t[0] ^= 2;
      asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7" : "+v"(acc) : "v"(av), "v"(t));
    } else {
      bf16x8_t bv = __builtin_bit_cast(bf16x8_t, w[kt]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0, 0, 0);
    }
  }
  asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
  if (lane < 16) { y[lane] = acc[0] + acc[1] + acc[2]; y[16 + lane] = acc[0]; y[32 + lane] = acc[1]; y[48 + lane] = acc[2]; }
}
int main() {
  std::vector<float> x(K); std::vector<uint16_t> W(K * N);
  for (int i = 0; i < K; ++i) x[i] = sinf(i * 1.37f) * 1.234567f;
  for (int i = 0; i < K * N; ++i) W[i] = f2bf(cosf(i * 0.77f));
  float *dx, *dy; uint16_t* dW;
  hipMalloc(&dx, K * 4); hipMalloc(&dy, 64 * 4); hipMalloc(&dW, K * N * 2);
  hipMemcpy(dx, x.data(), K * 4, hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), K * N * 2, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dW, dy, mode);
    float y[64]; hipMemcpy(y, dy, 256, hipMemcpyDeviceToHost);
    double maxe = 0, maxhi = 0;
    for (int n = 0; n < N; ++n) {
      double r = 0, rhi = 0; for (int kk = 0; kk < K; ++kk) { r += (double)x[kk] * bf2f(W[kk * N + n]); rhi += (double)bf2f(f2bf(x[kk])) * bf2f(W[kk * N + n]); }
      maxe = fmax(maxe, fabs(y[n] - r)); maxhi = fmax(maxhi, fabs(y[16 + n] - rhi));
    }
    printf("mode %d: max |y - ref| = %.3e   hi-row err = %.3e   y0=%f mid=%g lo=%g\n", mode, maxe, maxhi, y[0], y[32], y[48]);
  }
  return 0;
}
