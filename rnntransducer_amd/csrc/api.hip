// Error plumbing + version of librnnt_hip (C ABI in include/rnnt_hip.h).
#include "common.hpp"

namespace rnnt {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace rnnt

extern "C" int rnnt_hip_version(void) { return RNNT_HIP_ABI_VERSION; }
extern "C" const char* rnnt_hip_last_error(void) { return rnnt::g_err; }
extern "C" int rnnt_hip_device_cus(void) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return cus;
}
