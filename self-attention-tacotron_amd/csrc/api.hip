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
