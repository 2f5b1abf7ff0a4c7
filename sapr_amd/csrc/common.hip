// Error plumbing and device queries of libsapr_hip.so.
#include "sapr_common.h"

namespace sapr {

std::string &last_error() {
  static thread_local std::string e;
  return e;
}

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

int hip_fail(hipError_t e, const char *what) {
  last_error() = std::string(what) + ": " + hipGetErrorString(e);
  (void)hipGetLastError();  // clear the sticky error
  return static_cast<int>(e) > 0 ? static_cast<int>(e) : 1;
}

}  // namespace sapr

extern "C" int sapr_abi_version(void) { return SAPR_ABI_VERSION; }

extern "C" const char *sapr_last_error(void) { return sapr::last_error().c_str(); }

extern "C" int sapr_device_info(int dev, int *cu_count, int *wave_size, char *arch, size_t arch_len) {
  hipDeviceProp_t p;
  SAPR_HIP_TRY(hipGetDeviceProperties(&p, dev));
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (wave_size) *wave_size = p.warpSize;
  if (arch && arch_len) snprintf(arch, arch_len, "%s", p.gcnArchName);
  return 0;
}
