// micro-benchmark: can a SIMD issue fp32 VALU work from one wavefront while another wavefront's
// v_mfma_f32_16x16x4_f32 instructions execute?  (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// mode bit 0: waves 0-3 run MFMAs; bit 1: waves 4-7 run VALU FMAs (waves w and w+4 share a SIMD)
__global__ __launch_bounds__(512) void k(float *out, int iters, int mode) {
  const int wave = threadIdx.x / 64;
  float r = 0.f;
  if (wave < 4) {
    if (mode & 1) {
      f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
      const float x = threadIdx.x * 0.001f, y = 1.0001f;
      for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
      }
      r = a0[0] + a1[1] + a2[2] + a3[3];
    }
  } else if (mode & 2) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = __builtin_fmaf(a[j], 1.0001f, 0.0001f);
    }
    for (int i = 0; i < 8; ++i) r += a[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
  float *out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  for (int mode : {1, 2, 3}) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<<<256, 512>>>(out, 10, mode);
    (void)hipEventRecord(e0);
    k<<<256, 512>>>(out, iters, mode);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d (%s): %.3f ms  [per iteration: 4 MFMA 16x16x4 f32 on one wave, 16 v_fma_f32 on the other wave of the SIMD]\n", mode,
           mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", ms);
  }
  return 0;
}
