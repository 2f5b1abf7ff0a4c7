// dev probe: out-of-range behaviour of raw buffer_load_dwordx4 on gfx950 (negative offsets, partial chunks)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *x, int n, float *out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, n * 4, 0x00020000);
  const int off = (int(threadIdx.x) * 4 - 8) * 4;
  const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = f[i];
}
int main() {
  float h[64], *dx, *dout, ho[64];
  for (int i = 0; i < 64; ++i) h[i] = 100 + i;
  (void)hipMalloc(&dx, sizeof(h));
  (void)hipMalloc(&dout, sizeof(ho));
  (void)hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 8>>>(dx + 16, 10, dout);  // "utterance" of 10 samples starting at element 16 of the allocation
  (void)hipMemcpy(ho, dout, 8 * 16, hipMemcpyDeviceToHost);
  for (int t = 0; t < 8; ++t)
    printf("offset %3d: %6.1f %6.1f %6.1f %6.1f\n", t * 4 - 8, ho[t * 4], ho[t * 4 + 1], ho[t * 4 + 2], ho[t * 4 + 3]);
  return 0;
}
