// Shared host/device helpers for libsapr_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <string>

#include "../../include/sapr_hip.h"

namespace sapr {

constexpr int kWave = 64;  // CDNA wavefront

// ---- error plumbing -------------------------------------------------------------
std::string &last_error();
int fail(int code, const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);

#define SAPR_HIP_TRY(expr)                                      \
  do {                                                          \
    hipError_t _e = (expr);                                     \
    if (_e != hipSuccess) return ::sapr::hip_fail(_e, #expr);   \
  } while (0)

#define SAPR_REQUIRE(cond, ...)                                         \
  do {                                                                  \
    if (!(cond)) return ::sapr::fail(SAPR_ERR_ARG, __VA_ARGS__);        \
  } while (0)

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// hipGetLastError() reports (and clears) the last error of ANY earlier runtime call on this thread —
// including ones made by the host framework sharing the runtime — so it is cleared right before
// each launch whose status is then read back with it.
#define SAPR_LAUNCH(...)             \
  do {                               \
    (void)hipGetLastError();         \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)

// ---- device helpers -------------------------------------------------------------
__device__ __forceinline__ constexpr double neg_inf() { return -__builtin_huge_val(); }

// numpy's pair-wise float reduction over a contiguous axis of length N <= 128
// (numpy/core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum): N < 8 is a plain
// left-to-right loop from 0.0; otherwise eight running sums over blocks of eight,
// combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the N%8 tail added in order.
// hmmlearn's log-density calls .sum(axis=-1) over D contiguous doubles, so matching it
// term for term makes our emission scores bit-identical to the CPU evaluation.
template <int N>
__device__ __forceinline__ double np_pairwise_sum(const double (&a)[N]) {
  static_assert(N >= 1 && N <= 128, "numpy switches to recursive halving above 128 terms");
  if constexpr (N < 8) {
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) r += a[i];
    return r;
  } else {
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    constexpr int kFull = N - (N % 8);
#pragma unroll
    for (int i = 8; i < kFull; i += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
    for (int i = kFull; i < N; ++i) res += a[i];
    return res;
  }
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    int other = __shfl_xor(v, o, 64);
    v = other > v ? other : v;
  }
  return v;
}

}  // namespace sapr
