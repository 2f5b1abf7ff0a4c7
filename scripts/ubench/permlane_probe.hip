// dev probe: lane semantics of the gfx950 v_permlane{16,32}_swap instructions
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
  unsigned a = threadIdx.x, b = threadIdx.x + 1000;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[threadIdx.x * 4 + 0] = r[0];
  out[threadIdx.x * 4 + 1] = r[1];
  out[threadIdx.x * 4 + 2] = q[0];
  out[threadIdx.x * 4 + 3] = q[1];
}
int main() {
  unsigned *d, h[256];
  hipMalloc(&d, sizeof(h));
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 8)
    printf("lane %2d: swap32 -> (%4u, %4u)   swap16 -> (%4u, %4u)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
