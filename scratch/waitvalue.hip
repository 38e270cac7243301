// Can a stream wait (hipStreamWaitValue32) on a counter that a RUNNING kernel increments?  (producer/consumer between a
// persistent kernel and later stream work without ending the kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void producer(unsigned* ctr, unsigned long long* ts, int steps) {
  for (int s = 0; s < steps; ++s) {
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(100);        // ~"one chunk" of work
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      atomicAdd(ctr, 1u);
      if (blockIdx.x == 0) ts[s] = wall_clock64();
    }
  }
}
__global__ void consumer(const unsigned* ctr, unsigned* seen, unsigned long long* ts, int k) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { seen[k] = *ctr; ts[k] = wall_clock64(); }
}
int main() {
  const int NB = 128, STEPS = 6;
  unsigned *ctr, *seen; unsigned long long *tp, *tc;
  bool sig = true;
  if (hipExtMallocWithFlags((void**)&ctr, 64, hipMallocSignalMemory) != hipSuccess) { sig = false; hipMalloc(&ctr, 64); }
  printf("signal memory: %d\n", (int)sig);
  hipMalloc(&seen, 64 * 4); hipMalloc(&tp, 64 * 8); hipMalloc(&tc, 64 * 8);
  hipMemset(ctr, 0, 64); hipMemset(seen, 0, 256);
  hipStream_t a, b; hipStreamCreate(&a); hipStreamCreate(&b);
  hipLaunchKernelGGL(producer, dim3(NB), dim3(256), 0, a, ctr, tp, STEPS);
  for (int k = 0; k < STEPS; ++k) {
    hipError_t e = hipStreamWaitValue32(b, ctr, (unsigned)((k + 1) * NB), hipStreamWaitValueGte, 0xFFFFFFFFu);
    if (e != hipSuccess) { printf("hipStreamWaitValue32 failed: %d %s\n", (int)e, hipGetErrorString(e)); return 1; }
    hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, ctr, seen, tc, k);
  }
  hipDeviceSynchronize();
  unsigned hs[64]; unsigned long long hp[64], hc[64];
  hipMemcpy(hs, seen, 256, hipMemcpyDeviceToHost); hipMemcpy(hp, tp, 512, hipMemcpyDeviceToHost); hipMemcpy(hc, tc, 512, hipMemcpyDeviceToHost);
  for (int k = 0; k < STEPS; ++k)
    printf("chunk %d: consumer saw counter %u (needs >= %d), started %.1f us after the producer's block 0 signalled\n", k, hs[k],
           (k + 1) * NB, ((double)hc[k] - (double)hp[k]) / 100.0);
  // second try with plain hipMalloc memory
  return 0;
}
