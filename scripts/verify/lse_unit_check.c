/* Accuracy of csrc/lse_unit.h (e = exp(-d), 1 / (1 + e), log(1 + e) for d >= 0: the two-term log-sum-exp of the E-step
   recursions) against expl / log1pl, on the CPU: the header is plain C, so this is the code the device runs, with the
   reciprocal estimate replaced by a float32 one (no better than the hardware's).
     gcc -O2 -ffp-contract=off -o lse_unit_check lse_unit_check.c -lm && ./lse_unit_check [n]
   Prints the largest errors in units of the last place of the exact values and fails above 6. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../sapr_amd/csrc/lse_unit.h"

static double ulps(long double ref, double got) {
  if (ref == 0.0L) return got == 0.0 ? 0.0 : 1e300;
  int ex;
  frexpl(ref, &ex);
  const long double ulp = ldexpl(1.0L, (ex - 53 < -1074) ? -1074 : ex - 53);
  return (double)(fabsl((long double)got - ref) / ulp);
}

int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : 10000000L;
  double worst[3] = {0, 0, 0}, at[3] = {0, 0, 0};
  uint64_t rng = 0x9E3779B97F4A7C15ull;
  for (long i = 0; i <= n; ++i) {
    double cand[4];
    cand[0] = 760.0 * (double)i / (double)n;                 /* uniform grid over the whole range, underflow included */
    rng = rng * 6364136223846793005ull + 1442695040888963407ull;
    cand[1] = 40.0 * ldexp((double)(rng >> 11), -53);        /* where the result still moves a log-likelihood */
    cand[2] = ldexp((double)(rng >> 11), -53 - (int)(i % 60)); /* small arguments: e close to 1 */
    cand[3] = 0.34657359027997264 * (double)(2 * (i % 2000) + 1) + ldexp((double)(rng >> 40), -30); /* reduction boundaries */
    for (int c = 0; c < 4; ++c) {
      double e, inv, l1p;
      lse2_terms(cand[c], &e, &inv, &l1p);
      const long double re = expl(-(long double)cand[c]);
      const double u[3] = {ulps(re, e), ulps(1.0L / (1.0L + re), inv), ulps(log1pl(re), l1p)};
      for (int q = 0; q < 3; ++q)
        if (u[q] > worst[q]) worst[q] = u[q], at[q] = cand[c];
    }
  }
  /* exp_unit over the whole double range of results, both signs */
  double worst_x = 0.0, at_x = 0.0;
  for (long i = 0; i <= n; ++i) {
    const double x = -760.0 + 1469.7 * (double)i / (double)n;   /* up to 709.7: exp stays below DBL_MAX */
    const double u = ulps(expl((long double)x), exp_unit(x));
    if (u > worst_x) worst_x = u, at_x = x;
  }
  int okx = exp_unit(INFINITY) == INFINITY && exp_unit(-INFINITY) == 0.0 && exp_unit(1000.0) == INFINITY &&
            exp_unit(-1000.0) == 0.0 && exp_unit(0.0) == 1.0 && exp_unit(NAN) != exp_unit(NAN);
  printf("exp_unit: worst error %.3f ulp (x = %.17g); edges %s\n", worst_x, at_x, okx ? "ok" : "WRONG");
  if (!okx || worst_x > 2.0) return 1;
  /* edges: 0 (e = 1), the clamp, infinity, NaN */
  double e, inv, l1p;
  lse2_terms(0.0, &e, &inv, &l1p);
  int ok = e == 1.0 && inv == 0.5 && fabs(l1p - 0.6931471805599453) < 3e-16;
  lse2_terms(1e9, &e, &inv, &l1p);
  ok = ok && e == 0.0 && inv == 1.0 && l1p == 0.0;
  lse2_terms(INFINITY, &e, &inv, &l1p);
  ok = ok && e == 0.0 && inv == 1.0 && l1p == 0.0;
  lse2_terms(NAN, &e, &inv, &l1p);
  ok = ok && e != e && inv != inv && l1p != l1p;
  printf("lse_unit: %ld x 4 arguments, worst error exp %.3f ulp (d = %.17g), 1/(1+e) %.3f ulp (d = %.17g), log1p %.3f ulp "
         "(d = %.17g); edges %s\n", n, worst[0], at[0], worst[1], at[1], worst[2], at[2], ok ? "ok" : "WRONG");
  return (ok && worst[0] <= 6.0 && worst[1] <= 6.0 && worst[2] <= 6.0) ? 0 : 1;
}
