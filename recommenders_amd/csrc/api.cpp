// api.cpp -- library-wide entry points: version, error string, device query.
#include <stdarg.h>
#include <string.h>

#include <stdlib.h>

#include <map>
#include <mutex>
#include <set>
#include <string>
#include <utility>

#include "common.h"

namespace tfrs {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {
std::mutex g_opt_mu;
std::map<std::string, const char *> g_opt;      // overrides set through tfrs_set_option
std::set<std::string> g_opt_values;             // interned values: pointers handed out stay valid
}  // namespace

const char *option(const char *name) {
  {
    std::lock_guard<std::mutex> lock(g_opt_mu);
    if (!g_opt.empty()) {
      auto it = g_opt.find(name);
      if (it != g_opt.end()) return it->second;
    }
  }
  return getenv(name);
}

hipError_t ensure_dynamic_lds(const void *kernel, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  // last (kernel, device, bytes) this thread set or confirmed: the steady state of a launch loop
  static thread_local const void *t_kernel = nullptr;
  static thread_local int t_dev = -1, t_bytes = 0;
  if (t_kernel == kernel && t_dev == dev && t_bytes >= bytes) return hipSuccess;
  static std::mutex mu;
  static std::map<std::pair<const void *, int>, int> done;   // (kernel, device) -> bytes granted
  std::lock_guard<std::mutex> lock(mu);
  int &have = done[std::make_pair(kernel, dev)];
  if (have < bytes) {
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    have = bytes;
  }
  t_kernel = kernel;
  t_dev = dev;
  t_bytes = have;
  return hipSuccess;
}

}  // namespace tfrs

extern "C" int tfrs_set_option(const char *name, const char *value) {
  TFRS_CHECK_ARG(name && strncmp(name, "TFRS_", 5) == 0, "set_option: option names start with TFRS_");
  std::lock_guard<std::mutex> lock(tfrs::g_opt_mu);
  if (!value) {
    tfrs::g_opt.erase(name);          // back to the environment (or unset)
  } else {
    tfrs::g_opt[name] = tfrs::g_opt_values.insert(value).first->c_str();
  }
  return TFRS_OK;
}

extern "C" int tfrs_get_option(const char *name, char *value_h, int value_len) {
  TFRS_CHECK_ARG(name && value_h && value_len > 0, "get_option: bad argument");
  const char *v = tfrs::option(name);
  if (!v) {
    value_h[0] = '\0';
    return 0;                          // unset
  }
  strncpy(value_h, v, (size_t)value_len - 1);
  value_h[value_len - 1] = '\0';
  return 1;
}

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
