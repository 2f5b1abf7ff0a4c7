// Operand / result layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, found by experiment (dev tool): for every pair of
// lanes (la, lb) the instruction runs with A = 1 in lane la only and B = 1 in lane lb only; the lanes of D that
// become 1 tell which (block, i, k) lane la feeds and which (block, k, j) lane lb feeds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(double *out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      out[(la * 64 + lb) * 64 + lane] = d;
    }
}
int main() {
  double *out;
  (void)hipMalloc(&out, 64 * 64 * 64 * 8);
  probe<<<1, 64>>>(out);
  std::vector<double> h(64 * 64 * 64);
  (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  // for each A lane: the set of B lanes it pairs with and the output lanes it reaches
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d pairs with B lanes -> D lanes:", la);
    for (int lb = 0; lb < 64; ++lb)
      for (int l = 0; l < 64; ++l)
        if (h[(la * 64 + lb) * 64 + l] != 0.0) printf(" (%d->%d)", lb, l);
    printf("\n");
  }
  return 0;
}
