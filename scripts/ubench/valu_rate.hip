// micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_pk_add_f32 (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float *out, int iters) {
  float a[8]; f2 b[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = f2{a[i], a[i] + 1.f}; }
  const float c = 1.0001f, d = 0.0001f; const f2 c2 = {c, c}, d2 = {d, d};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) a[i] = __builtin_fmaf(a[i], c, d);
        if (MODE == 1) b[i] = __builtin_elementwise_fma(b[i], c2, d2);
        if (MODE == 2) b[i] = b[i] + d2;
        if (MODE == 3) a[i] = a[i] + d;
      }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + b[i].x + b[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, int wpb, int blocks) {
  float *out; hipMalloc(&out, blocks * wpb * 64 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, wpb * 64>>>(out, 10);
  hipEventRecord(e0); k<MODE><<<blocks, wpb * 64>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_wave = double(iters) * 32;
  double cyc = ms * 1e-3 * 2.4e9;  // nominal clock
  printf("%-14s waves/CU=%2d  %.3f ms  -> %.2f cycles/instr/wave (nominal 2.4 GHz), %.2f wave-instr/cycle/SIMD\n", name,
         wpb * blocks / 256, ms, cyc / instr_per_wave, instr_per_wave * (wpb * blocks / 256 / 4.0) / cyc);
  hipFree(out);
}
int main() {
  for (int wpb : {4, 8, 16}) {
    run<0>("v_fma_f32", wpb, 256); run<1>("v_pk_fma_f32", wpb, 256); run<2>("v_pk_add_f32", wpb, 256); run<3>("v_add_f32", wpb, 256);
  }
  return 0;
}
