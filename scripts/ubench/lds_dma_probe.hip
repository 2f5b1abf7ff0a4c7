// Probe (gfx950): where does global_load_lds_dwordx4 put each lane's 16 bytes?  Expected: base + 16 * lane.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *g, float *out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *s = reinterpret_cast<float *>(smem);
  for (int i = threadIdx.x; i < 1024; i += 64) s[i] = -1.f;
  __syncthreads();
  // lane l fetches 4 floats starting at g[8 * (63 - l)]: a lane-dependent, non-contiguous source
  const float *p = g + 8 * (63 - threadIdx.x);
  __builtin_amdgcn_global_load_lds(p, (__attribute__((address_space(3))) void *)(s + 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = s[i];
}
int main() {
  float h[512], *g, *o, r[1024];
  for (int i = 0; i < 512; ++i) h[i] = i;
  hipMalloc(&g, sizeof h); hipMalloc(&o, sizeof r);
  hipMemcpy(g, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 64, 4096>>>(g, o);
  hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j)
      if (r[256 + 4 * l + j] != 8 * (63 - l) + j) ++bad;
  for (int i = 0; i < 256; ++i) if (r[i] != -1.f) ++bad;
  for (int i = 512; i < 1024; ++i) if (r[i] != -1.f) ++bad;
  printf("lane-contiguous 16-byte slots at the given base: %s (%d mismatches); first slots: %g %g %g %g | %g %g\n",
         bad ? "NO" : "yes", bad, r[256], r[257], r[258], r[259], r[260], r[261]);
  return bad != 0;
}
