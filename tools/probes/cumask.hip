// Which XCD / CU does a workgroup land on under a stream CU mask?  (hipExtStreamCreateWithCUMask bit layout on gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void where_k(int* out) {
  if (threadIdx.x == 0) {
    int xcc = (int)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;       // HW_REG_XCC_ID[3:0]
    int hw = (int)__builtin_amdgcn_s_getreg((31 << 11) | 4);              // HW_REG_HW_ID (wave, simd, cu, sh, se ...)
    out[blockIdx.x * 2] = xcc; out[blockIdx.x * 2 + 1] = hw;
  }
  // stay resident long enough that all blocks are co-resident (one per CU because of the LDS below)
  __shared__ float pad[36 * 1024];
  pad[threadIdx.x] = 1.f;
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(100);
  if (pad[threadIdx.x] < 0) out[0] = 0;
}
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("CUs %d\n", pr.multiProcessorCount);
  int* d; hipMalloc(&d, 4096 * 2 * sizeof(int));
  const uint32_t patterns[][8] = {
    {0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0, 0, 0, 0},          // first 128 bits
    {0x55555555, 0x55555555, 0x55555555, 0x55555555, 0x55555555, 0x55555555, 0x55555555, 0x55555555},   // even bits
    {0x0F0F0F0F, 0x0F0F0F0F, 0x0F0F0F0F, 0x0F0F0F0F, 0x0F0F0F0F, 0x0F0F0F0F, 0x0F0F0F0F, 0x0F0F0F0F},   // low nibbles
    {0xFFFFFFFF, 0, 0xFFFFFFFF, 0, 0xFFFFFFFF, 0, 0xFFFFFFFF, 0},
  };
  for (int pi = 0; pi < 4; ++pi) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, patterns[pi]);
    if (e != hipSuccess) { printf("pattern %d: create failed %d\n", pi, (int)e); continue; }
    hipMemsetAsync(d, 0xFF, 4096 * 2 * sizeof(int), st);
    hipLaunchKernelGGL(where_k, dim3(128), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    std::vector<int> h(256);
    hipMemcpy(h.data(), d, 256 * sizeof(int), hipMemcpyDeviceToHost);
    int cnt[16] = {0};
    for (int b = 0; b < 128; ++b) cnt[h[b * 2] & 15]++;
    printf("pattern %d: blocks per XCC:", pi);
    for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
    printf("   first blocks xcc:");
    for (int b = 0; b < 16; ++b) printf(" %d", h[b * 2]);
    printf("\n");
    hipStreamDestroy(st);
  }
  return 0;
}
