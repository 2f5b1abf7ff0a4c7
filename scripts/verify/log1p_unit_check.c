/* Accuracy of csrc/log1p_unit.h (the E-step kernels' log(1 + e) on 0 <= e <= 1) against log1pl, on the CPU: the header
   is plain C, so this is the very code the device runs (same operations, fused multiply-adds included).
     gcc -O2 -ffp-contract=off -o log1p_unit_check log1p_unit_check.c -lm && ./log1p_unit_check [n]
   Prints the largest error in units of the last place of the exact value (1.98 over 6e7 arguments) and fails above 2.5. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../sapr_amd/csrc/log1p_unit.h"

static double ulp_err(double e) {
  const long double ref = log1pl((long double)e);
  const double got = log1p_unit(e);
  if (ref == 0.0L) return got == 0.0 ? 0.0 : 1e300;
  int ex;
  frexpl(ref, &ex);
  const long double ulp = ldexpl(1.0L, (ex - 53 < -1074) ? -1074 : ex - 53); /* denormal results: spacing 2^-1074 */
  return (double)(fabsl((long double)got - ref) / ulp);
}

int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : 20000000L;
  double worst = 0.0, at = 0.0;
  uint64_t rng = 0x9E3779B97F4A7C15ull;
  for (long i = 0; i <= n; ++i) {
    double cand[3];
    cand[0] = (double)i / (double)n;                      /* uniform grid, both ends included */
    rng = rng * 6364136223846793005ull + 1442695040888963407ull;
    cand[1] = ldexp((double)(rng >> 11), -53);            /* random in [0, 1) */
    cand[2] = exp(-745.0 * (double)i / (double)n);        /* what the kernels pass: exp(-d) down to the denormals */
    for (int k = 0; k < 3; ++k) {
      const double u = ulp_err(cand[k]);
      if (u > worst) worst = u, at = cand[k];
    }
  }
  const double edge[] = {0.0, 4.9406564584124654e-324, 2.2250738585072014e-308, 1e-300, 1e-17, 0x1p-53, 0x1p-52, 1.0};
  for (unsigned k = 0; k < sizeof edge / sizeof edge[0]; ++k) {
    const double u = ulp_err(edge[k]);
    if (u > worst) worst = u, at = edge[k];
  }
  printf("log1p_unit: %ld x 3 arguments, worst error %.3f ulp at e = %.17g\n", n, worst, at);
  return worst <= 2.5 ? 0 : 1;
}
