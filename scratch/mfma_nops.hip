// how many wait states does v_mfma_f32_16x16x32_bf16 -> VALU read need on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <int N> __global__ void k(float* out) {
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x3F80; b[i] = 0x3F80; }   // bf16 1.0 -> every output element = 32
  f32x4_t acc = {-1.f, -1.f, -1.f, -1.f};
  float r;
  asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  if (N > 0) asm volatile("s_nop %0" ::"n"(N > 0 ? N - 1 : 0));
  asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(acc[1]));       // VALU read of the 2nd result register
  out[threadIdx.x] = r;
}
template <int N> void run(float* out) {
  hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, out);
  float h[64]; hipMemcpy(h, out, 256, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; ++i) bad += (h[i] != 31.f);
  printf("explicit wait states after MFMA = %2d : %s (lane0 = %g)\n", N, bad ? "WRONG" : "ok", h[0]);
}
int main() {
  float* out; hipMalloc(&out, 256);
  run<0>(out); run<1>(out); run<2>(out); run<3>(out); run<4>(out); run<5>(out); run<6>(out); run<7>(out); run<8>(out);
  run<10>(out); run<12>(out); run<16>(out);
  return 0;
}
