// ping-pong latency between two workgroups through global memory with different cache-scope bits (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
template <int SM> __device__ __forceinline__ void st(u64* p, u64 v) {
  if (SM == 0) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  if (SM == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
  if (SM == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  if (SM == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
template <int LM> __device__ __forceinline__ u64 ld(u64* p) {
  u64 v;
  if (LM == 0) asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LM == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LM == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LM == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LM == 4) asm volatile("buffer_inv sc0\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LM == 5) asm volatile("buffer_inv sc1\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LM == 6) asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int SM, int LM>
__global__ void pp(u64* flags, int partner, int N, unsigned long long* out, int* fail) {
  const int b = blockIdx.x;
  if (b != 0 && b != partner) return;
  if (threadIdx.x != 0) return;
  u64* mine = flags + (b == 0 ? 0 : 64);
  u64* theirs = flags + (b == 0 ? 64 : 0);
  const unsigned long long t0 = wall_clock64();
  for (int i = 1; i <= N; ++i) {
    if (b == 0) st<SM>(mine, (u64)i);
    unsigned spins = 0;
    while (ld<LM>(theirs) != (u64)i) { if (++spins > (1u << 22)) { *fail = 1; return; } }
    if (b != 0) st<SM>(mine, (u64)i);
  }
  if (b == 0) *out = wall_clock64() - t0;
}
template <int SM, int LM> void run(const char* name, u64* flags, unsigned long long* out, int* fail, int partner) {
  const int N = 2000;
  hipMemset(flags, 0, 1024); hipMemset(fail, 0, 4); hipMemset(out, 0, 8);
  hipLaunchKernelGGL((pp<SM, LM>), dim3(16), dim3(64), 0, 0, flags, partner, N, out, fail);
  hipDeviceSynchronize();
  unsigned long long t; int f; hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
  printf("%-34s partner=%2d : %s round trip %.3f us\n", name, partner, f ? "FAILED (stale)" : "ok", t / 100.0 / N);
}
int main() {
  u64* flags; unsigned long long* out; int* fail;
  hipMalloc(&flags, 1024); hipMalloc(&out, 8); hipMalloc(&fail, 4);
  for (int partner : {8, 1}) {
    run<2, 2>("store sc1 / load sc1 (agent)", flags, out, fail, partner);
    run<0, 2>("store plain / load sc1", flags, out, fail, partner);
    run<1, 1>("store sc0 / load sc0 (workgroup)", flags, out, fail, partner);
    run<0, 1>("store plain / load sc0", flags, out, fail, partner);
    run<0, 4>("store plain / inv sc0 + load", flags, out, fail, partner);
    run<0, 5>("store plain / inv sc1 + load", flags, out, fail, partner);
    run<0, 6>("store plain / load nt", flags, out, fail, partner);
    run<3, 3>("store sc0sc1 / load sc0sc1 (system)", flags, out, fail, partner);
  }
  return 0;
}
