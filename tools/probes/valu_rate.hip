// Issue rate of the instruction kinds the attention energy rows are made of (gfx950): cycles per wave instruction for
// v_fma_f32, v_pk_fma_f32, v_exp_f32, v_rcp_f32 with 1 and 2 waves per SIMD (the cluster kernels run 2).  Answers whether a
// packed polynomial tanh could beat exp2 + rcp.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/probes/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND> __global__ void rate_k(float* out, unsigned long long* cyc, int iters, float seed) {
  float a[8];
  v2f p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed + 0.01f * i + 1e-3f * threadIdx.x; p[i] = (v2f){a[i], a[i] + 0.5f}; }
  const v2f c2 = (v2f){0.999f, 1.001f}, d2 = (v2f){1e-3f, -1e-3f};
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {          // 8 independent chains: no dependency stalls
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(0.999f), "v"(1e-3f));
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(c2), "v"(d2));
        if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if (KIND == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
        if (KIND == 5) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(-9.f), "v"(9.f));
      }
    }
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND> void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 2000;
  for (int threads : {256, 512}) {           // 1 and 2 waves per SIMD
    rate_k<KIND><<<1, threads>>>(out, cyc, iters, 0.3f);
    hipDeviceSynchronize();
    rate_k<KIND><<<1, threads>>>(out, cyc, iters, 0.3f);
    hipDeviceSynchronize();
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    const double per = (double)c / (iters * 32.0);
    printf("%-14s %d waves/SIMD: %6.2f clock64 ticks per wave instruction (per SIMD: %6.2f)\n", name, threads / 256, per,
           per / (threads / 256));
  }
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4096 * sizeof(float)); hipMalloc(&cyc, sizeof(unsigned long long));
  int khz = 0, wall = 0;
  hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
  printf("shader clock %d kHz, wall clock %d kHz (clock64 = s_memtime: constant-rate counter; compare kinds, not absolute)\n", khz, wall);
  run<0>("v_fma_f32", out, cyc);
  run<1>("v_pk_fma_f32", out, cyc);
  run<4>("v_pk_mul_f32", out, cyc);
  run<5>("v_med3_f32", out, cyc);
  run<2>("v_exp_f32", out, cyc);
  run<3>("v_rcp_f32", out, cyc);
  return 0;
}
