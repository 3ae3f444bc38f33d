// api.cpp -- library-wide entry points: version, error string, device query.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace tfrs {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace tfrs

extern "C" int tfrs_version(void) { return 100; /* 0.1.0 */ }

extern "C" const char *tfrs_last_error(void) { return tfrs::g_err; }

extern "C" int tfrs_device_info(int dev, int *cu_count_h, int *lds_bytes_h, char *arch_h,
                                int arch_len) {
  hipDeviceProp_t prop;
  TFRS_HIP(hipGetDeviceProperties(&prop, dev));
  if (cu_count_h) *cu_count_h = prop.multiProcessorCount;
  if (lds_bytes_h) *lds_bytes_h = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (arch_h && arch_len > 0) {
    strncpy(arch_h, prop.gcnArchName, (size_t)arch_len - 1);
    arch_h[arch_len - 1] = '\0';
  }
  return TFRS_OK;
}
