// Error plumbing and version entry of the C ABI (include/urse.h).
#include <stdarg.h>

#include "urse_common.h"

namespace urse {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int device_cu_count() {
  static thread_local int cached_dev = -1, cached = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev != cached_dev) {
    hipDeviceProp_t prop;
    cached = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cached_dev = dev;
  }
  return cached;
}
}  // namespace urse

extern "C" int urse_version(void) { return 1; }
extern "C" const char* urse_last_error(void) { return urse::g_err; }
