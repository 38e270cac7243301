// Library-level entry points of libsatt_hip.so.
#include "common.h"
#include <string.h>

extern "C" int satt_version(void) { return 100; }

extern "C" const char* satt_strerror(int code) {
  switch (code) {
    case SATT_OK: return "ok";
    case SATT_E_BADARG: return "bad argument (shape / pointer / flag combination)";
    case SATT_E_UNSUPPORTED: return "unsupported size for this kernel";
    case SATT_E_LAUNCH: return "kernel launch failed";
    case SATT_E_ARCH: return "device is not gfx950";
    default: return "unknown error";
  }
}

extern "C" int satt_arch_supported(int device) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SATT_E_BADARG;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

// ---- diagnostics: LDS poison.  LDS is not cleared between kernels (nor between processes): a kernel that reads an LDS word before
// writing it sees whatever the previous workgroup on that CU left there - its own previous launch, as a rule, which is why such a read
// can hide for a whole session and show only in the first launch of a process.  This launch leaves `pattern` in every LDS word of every
// CU (512 workgroups x 160 KB: each takes a whole CU, so the first 256 cover an idle chip; `linger` x ~3.4 us keeps a workgroup on its
// CU while the rest of the grid spreads - 0 for the per-launch sweep of SATT_DEBUG_POISON_LDS (_lib.py), ~100 for a single shot);
// tools/decode_cold.py runs the cold decode behind it with a NaN pattern and with plausible finite ones.
namespace {
__global__ __launch_bounds__(512) void poison_lds_k(uint32_t pattern, int words, int linger) {
  extern __shared__ uint32_t lds_words[];
  for (int i = threadIdx.x; i < words; i += 512) lds_words[i] = pattern;
  __syncthreads();
  for (int i = 0; i < linger; ++i) __builtin_amdgcn_s_sleep(127);
  if (lds_words[(threadIdx.x * 97) % words] != pattern) __builtin_trap();
}
}  // namespace
extern "C" int satt_debug_poison_lds(uint32_t pattern, int linger, void* stream) {
  const int bytes = 160 * 1024;
  if (hipFuncSetAttribute((const void*)poison_lds_k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return SATT_E_LAUNCH;
  }
  hipLaunchKernelGGL(poison_lds_k, dim3(512), dim3(512), bytes, (hipStream_t)stream, pattern, bytes / 4, linger < 0 ? 0 : linger);
  SATT_LAUNCH_CHECK();
  return SATT_OK;
}
