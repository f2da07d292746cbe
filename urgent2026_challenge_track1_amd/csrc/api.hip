// Error plumbing and version entry of the C ABI (include/urse.h).
#include <stdarg.h>

#include <atomic>

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
static std::atomic<int> g_launches[URSE_KV_COUNT];
void note_launch(int variant) {
  if (variant >= 0 && variant < URSE_KV_COUNT) g_launches[variant].fetch_add(1, std::memory_order_relaxed);
}
}  // namespace urse

extern "C" int urse_launch_count(int variant) {
  if (variant < 0 || variant >= URSE_KV_COUNT) return -1;
  return urse::g_launches[variant].load(std::memory_order_relaxed);
}
extern "C" int urse_launch_counts_reset(void) {
  for (int i = 0; i < URSE_KV_COUNT; ++i) urse::g_launches[i].store(0, std::memory_order_relaxed);
  return URSE_OK;
}

extern "C" int urse_version(void) { return 1; }
extern "C" const char* urse_last_error(void) { return urse::g_err; }
