// probe: raw buffer load with a NEGATIVE vector offset plus a positive immediate offset whose sum is in range —
// does the hardware return the element (32-bit wrap) or 0 (out of range)?  (dev tool; decides how mfcc_wave.h forms offsets)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *src, float *out, int n) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, n * 4, 0x00020000);
  int voff = -128 + 4 * (int)threadIdx.x;  // lanes 0..31 negative, 32.. non-negative
  float a, b;
  asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:128\n s_waitcnt vmcnt(0)" : "=v"(a) : "v"(voff), "s"(rsrc));
  int full = voff + 128;
  asm volatile("buffer_load_dword %0, %1, %2, 0 offen\n s_waitcnt vmcnt(0)" : "=v"(b) : "v"(full), "s"(rsrc));
  out[threadIdx.x] = a;
  out[64 + threadIdx.x] = b;
}
int main() {
  float h[256], *d, *o, r[128];
  for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
  (void)hipMalloc(&d, sizeof h);
  (void)hipMalloc(&o, sizeof r);
  (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o, 256);
  (void)hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  printf("neg voffset + imm: lane0 %.0f lane31 %.0f lane32 %.0f lane63 %.0f | full offset in VGPR: lane0 %.0f lane31 %.0f lane32 %.0f\n",
         r[0], r[31], r[32], r[63], r[64], r[95], r[96]);
  return 0;
}
