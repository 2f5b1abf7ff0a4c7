// micro-benchmark: dependent-issue latency of fp32 VALU ops and throughput of the permlane swaps / ds_bpermute (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(float *out, int iters) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
  const float d = 0.0001f;
  const int addr = ((threadIdx.x ^ 16) & 63) * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (MODE == 0) {  // ILP 1
#pragma unroll
        for (int i = 0; i < 8; ++i) a[0] = a[0] + d;
      }
      if (MODE == 1) {  // ILP 2
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[0] = a[0] + d; a[1] = a[1] + d; }
      }
      if (MODE == 2) {  // ILP 4
#pragma unroll
        for (int i = 0; i < 2; ++i) { a[0] += d; a[1] += d; a[2] += d; a[3] += d; }
      }
      if (MODE == 3) {  // permlane32 swaps on 4 independent pairs (8 instr)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[2 * j]), __float_as_uint(a[2 * j + 1]), false, false);
            a[2 * j] = __uint_as_float(q[0]);
            a[2 * j + 1] = __uint_as_float(q[1]);
          }
      }
      if (MODE == 4) {  // bpermute, 8 independent
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a[i])));
      }
      if (MODE == 5) {  // v_cndmask ILP 8
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (threadIdx.x & 16) ? a[i] : a[(i + 1) & 7];
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, int wpb, int blocks) {
  float *out;
  (void)hipMalloc(&out, blocks * wpb * 64 * 4);
  const int iters = 10000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<MODE><<<blocks, wpb * 64>>>(out, 10);
  (void)hipEventRecord(e0);
  k<MODE><<<blocks, wpb * 64>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = double(iters) * 32, cyc = ms * 1e-3 * 2.4e9;
  printf("%-22s waves/SIMD=%d  %.2f cycles/instr/wave  %.2f wave-instr/cycle/SIMD\n", name, wpb * blocks / 1024,
         cyc / instr_per_wave, instr_per_wave * (wpb * blocks / 1024.0) / cyc);
  (void)hipFree(out);
}
int main() {
  for (int wpb : {4, 8, 12}) {
    run<0>("add chain ILP1", wpb, 256);
    run<1>("add chain ILP2", wpb, 256);
    run<2>("add chain ILP4", wpb, 256);
    run<3>("permlane32_swap x8", wpb, 256);
    run<4>("ds_bpermute x8", wpb, 256);
    run<5>("cndmask ILP8", wpb, 256);
  }
  return 0;
}
