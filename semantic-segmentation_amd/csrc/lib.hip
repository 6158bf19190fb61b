// Library-wide entry points: version, per-thread error text, device probe.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace segk {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace segk

extern "C" int segk_version(void) { return 100; }

extern "C" const char* segk_last_error(void) { return segk::g_err; }

extern "C" int segk_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    segk::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return SEGK_HIP_ERROR;
  }
  return n;
}
