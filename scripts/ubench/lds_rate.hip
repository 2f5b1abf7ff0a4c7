// micro-benchmark: LDS bytes per clock of the read / write forms the MFCC core uses, by explicit instruction (dev tool).
// Each wavefront issues the instruction back to back (8 in flight, one s_waitcnt per 8) on conflict-free addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(float *out, int iters) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // per-wave window of 2 KB; lane-contiguous
  const unsigned a8 = (w * 2048 + lane * 8) & 32767, a4 = (w * 2048 + lane * 4) & 32767, a16 = (w * 4096 + lane * 16) & 32767;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // 8 x ds_read_b64
      v2f v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[i]) : "v"(a8), "n"(i * 512));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(v[i]));
      acc += v[0].x;
    }
    if (MODE == 1) {  // 4 x ds_read2_b64 (same bytes as MODE 0)
      v4f v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v[i]) : "v"(a8), "n"((i & 1) * 128), "n"((i & 1) * 128 + 64));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(v[i]));
      acc += v[0].x;
    }
    if (MODE == 2) {  // 8 x ds_read_b32
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[i]) : "v"(a4), "n"(i * 256));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(v[i]));
      acc += v[0];
    }
    if (MODE == 3) {  // 4 x ds_read2_b32
      v2f v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v[i]) : "v"(a4), "n"((i & 1) * 128), "n"((i & 1) * 128 + 64));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(v[i]));
      acc += v[0].x;
    }
    if (MODE == 4) {  // 4 x ds_read_b128
      v4f v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(a16), "n"(i * 1024));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(v[i]));
      acc += v[0].x;
    }
    if (MODE == 5) {  // 8 x ds_write_b32
      float d = acc;
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(a4), "v"(d), "n"(i * 256));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (MODE == 6) {  // 4 x ds_write2_b32
      float d = acc;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        asm volatile("ds_write2_b32 %0, %1, %1 offset0:%2 offset1:%3" ::"v"(a4), "v"(d), "n"((i & 1) * 128), "n"((i & 1) * 128 + 64));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (MODE == 7) {  // 4 x ds_write_b64
      v2f d = {acc, acc};
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a8), "v"(d), "n"(i * 512));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + lds[threadIdx.x];
}
template <int MODE>
void run(const char *name, int bytes_per_iter_lane, int wpb, int blocks) {
  float *out;
  (void)hipMalloc(&out, blocks * wpb * 64 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<MODE><<<blocks, wpb * 64, 32768>>>(out, 10);
  (void)hipEventRecord(e0);
  k<MODE><<<blocks, wpb * 64, 32768>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double cyc = ms * 1e-3 * 2.4e9;  // nominal clock
  const double bytes_per_cu = double(iters) * bytes_per_iter_lane * 64 * wpb * (blocks / 256.0);
  printf("%-18s waves/CU=%2d  %.1f B/clk/CU  (%.3f ms)\n", name, wpb * blocks / 256, bytes_per_cu / cyc, ms);
  (void)hipFree(out);
}
int main() {
  for (int wpb : {4, 8, 16}) {
    run<0>("ds_read_b64 x8", 64, wpb, 256);
    run<1>("ds_read2_b64 x4", 64, wpb, 256);
    run<2>("ds_read_b32 x8", 32, wpb, 256);
    run<3>("ds_read2_b32 x4", 32, wpb, 256);
    run<4>("ds_read_b128 x4", 64, wpb, 256);
    run<5>("ds_write_b32 x8", 32, wpb, 256);
    run<6>("ds_write2_b32 x4", 32, wpb, 256);
    run<7>("ds_write_b64 x4", 32, wpb, 256);
  }
  return 0;
}
