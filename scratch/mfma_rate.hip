// issue-rate microbenchmark: v_mfma_f32_16x16x32_bf16 vs v_mfma_f32_4x4x4_16b_bf16 (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* tm, int iters) {
  f32x4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  bf16x8_t a8 = {1, 2, 3, 4, 5, 6, 7, 8}, b8 = {1, 1, 1, 1, 1, 1, 1, 1};
  bf16x4_t a4 = {1, 2, 3, 4}, b4 = {1, 1, 1, 1};
  a8[0] += threadIdx.x; a4[0] += threadIdx.x;
  __syncthreads();
  unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c3, 0, 0, 0);
    } else if (MODE == 2) {      // B operand in AGPRs, zero SrcC, like the recurrent kernels
      typedef __attribute__((ext_vector_type(4))) int i32x4_t;
      i32x4_t ba = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};
      asm volatile("" : "+a"(ba));
      asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %4, %5, 0\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %5, 0\n\t"
                   "v_mfma_f32_16x16x32_bf16 %2, %4, %5, 0\n\tv_mfma_f32_16x16x32_bf16 %3, %4, %5, 0\n\ts_nop 7\n\ts_nop 7"
                   : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3) : "v"(a8), "a"(ba));
    } else {
      c0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, b4, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, b4, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, b4, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, b4, c3, 0, 0, 0);
    }
  }
  __syncthreads();
  unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) tm[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
  float* out; unsigned long long* tm; hipMalloc(&out, 512 * 4 * 4); hipMalloc(&tm, 64);
  const int iters = 2000;
  for (int threads : {64, 256, 512}) {
    for (int mode = 0; mode < 3; ++mode) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, out, tm, iters);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, out, tm, iters);
      else hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, out, tm, iters);
      hipDeviceSynchronize();
      unsigned long long t; hipMemcpy(&t, tm, 8, hipMemcpyDeviceToHost);
      const double ns = t * 10.0, per = ns / (iters * 4.0);
      printf("%s threads=%3d : %.2f ns per MFMA per wave-slot (%.1f cycles @2.4GHz); waves/SIMD=%d\n",
             mode == 0 ? "16x16x32_bf16" : (mode == 1 ? "4x4x4_16b_bf16" : "16x16x32 B=AGPR asm blk"), threads, per, per * 2.4, (threads / 64 + 3) / 4);
    }
  }
  return 0;
}
