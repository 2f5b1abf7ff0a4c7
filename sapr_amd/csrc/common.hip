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

namespace {
#include "lse_unit.h"
// one thread per argument: the device's own evaluation of csrc/lse_unit.h (v_rcp_f64 estimate, v_ldexp_f64, v_rndne_f64)
__global__ void lse_selftest_kernel(const double *__restrict__ d, int64_t n, double *__restrict__ out) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  double e, inv, l1p;
  lse2_terms(d[i], &e, &inv, &l1p);
  out[i] = e;
  out[n + i] = inv;
  out[2 * n + i] = l1p;
  out[3 * n + i] = exp_unit(-d[i]);
}
}  // namespace

}  // namespace sapr

extern "C" int sapr_selftest_lse(const double *d, int64_t n, double *out, void *stream) {
  SAPR_REQUIRE(d && out && n >= 0, "bad arguments");
  if (n > 0)
    SAPR_LAUNCH(sapr::lse_selftest_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                sapr::as_stream(stream), d, n, out);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

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
