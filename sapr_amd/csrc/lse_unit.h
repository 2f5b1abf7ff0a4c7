// The three values a two-term log-sum-exp of the lattice recursions needs, from d = |a - b| >= 0:
//     e = exp(-d),   inv = 1 / (1 + e),   l1p = log(1 + e)
// (estep.hip lse2 / lse2_share / logaddexp: log(exp a + exp b) = max + l1p, the share of the larger term in the sum =
// inv, of the smaller = e inv; custom.hip np_logaddexp*).  Round 3 composed them from the device library's exp (about
// 32 instructions with its special cases and coefficient moves), two IEEE divisions (1 / (1 + e), and e / (2 + e) in
// front of the atanh series: 13 instructions each) and that series — ~105 float64 instructions per call, nine calls per
// frame.  Here, for the one range the recursions use:
//   * exp(-d) = 2^k exp(r), k = rint(-d log2 e), |r| <= ln 2 / 2, Taylor polynomial of degree 13 (remainder 4e-18 of
//     the value), coefficients in constant memory so that they reach v_fma_f64 as scalar operands; arguments above 800
//     are clamped (the result is 0 from 745.2 on either way); NaN propagates;
//   * ONE reciprocal y = 1 / ((1 + e)(2 + e)) — hardware estimate and two Newton steps; the denominator lies in [2, 6],
//     so none of the scaling an IEEE division needs — gives both quotients: inv = (2 + e) y, s = e (1 + e) y;
//   * log(1 + e) = 2 atanh(s), s = e / (2 + e) <= 1/3, as the odd series 2 s (1 + z/3 + z^2/5 + ...), z = s^2 <= 1/9, cut
//     after z^16 / 33 (remainder below 2e-18 of the sum).
// 58 instructions.  Errors (scripts/verify/lse_unit_check.c compiles THIS header on the CPU and compares with expl / log1pl
// over the whole range; tests/test_build_guards_cpu.py): e below 1 ulp, inv below 3, l1p below 4.5 ulp of a value <= 0.693 that is added
// to log-likelihoods of magnitude 10^2 .. 10^4.  Plain C, shared by the device code and that check.
#pragma once

#ifdef __HIPCC__
#define SAPR_LSE_FN __device__ __forceinline__
#define SAPR_LSE_TAB __constant__
#define SAPR_LSE_RCP(x) __builtin_amdgcn_rcp(x)
#define SAPR_LSE_RINT(x) __builtin_rint(x)
#define SAPR_LSE_FMA(a, b, c) __builtin_fma(a, b, c)
#define SAPR_LSE_LDEXP(x, k) __builtin_amdgcn_ldexp(x, k)
#else
#include <math.h>
#define SAPR_LSE_FN static inline
#define SAPR_LSE_TAB static const
/* the CPU check starts Newton's iteration from a float32-accurate estimate: no better than v_rcp_f64 */
#define SAPR_LSE_RCP(x) ((double)(1.0f / (float)(x)))
#define SAPR_LSE_RINT(x) rint(x)
#define SAPR_LSE_FMA(a, b, c) fma(a, b, c)
#define SAPR_LSE_LDEXP(x, k) ldexp(x, k)
#endif

// 1/13!, 1/12!, ..., 1/2!
SAPR_LSE_TAB double kExpNegCoef[12] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0,
                                       1.0 / 362880.0,     1.0 / 40320.0,     1.0 / 5040.0,     1.0 / 720.0,
                                       1.0 / 120.0,        1.0 / 24.0,        1.0 / 6.0,        0.5};
// 1/33, 1/31, ..., 1/3 (the atanh series)
SAPR_LSE_TAB double kLseAtanhCoef[16] = {1.0 / 33.0, 1.0 / 31.0, 1.0 / 29.0, 1.0 / 27.0, 1.0 / 25.0, 1.0 / 23.0,
                                         1.0 / 21.0, 1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0,
                                         1.0 / 9.0,  1.0 / 7.0,  1.0 / 5.0,  1.0 / 3.0};

// exp(-d) for d >= 0
SAPR_LSE_FN double exp_neg_unit(double d) {
  const double dc = d > 800.0 ? 800.0 : d;  // (NaN stays NaN)
  const double t = -dc;
  const double k = SAPR_LSE_RINT(t * 1.4426950408889634074);                 // log2(e)
  double r = SAPR_LSE_FMA(k, -6.93147180369123816490e-01, t);               // ln 2, high part (32 significant bits)
  r = SAPR_LSE_FMA(k, -1.90821492927058770002e-10, r);                       // ln 2, low part
  double p = kExpNegCoef[0];
#ifdef __HIPCC__
#pragma unroll
#endif
  for (int i = 1; i < 12; ++i) p = SAPR_LSE_FMA(p, r, kExpNegCoef[i]);
  p = SAPR_LSE_FMA(p, r, 1.0);
  p = SAPR_LSE_FMA(p, r, 1.0);
  return SAPR_LSE_LDEXP(p, (int)k);
}

// exp(x) for any x by the same chain (arguments beyond +-800 are clamped: the result is 0 resp. +inf from 745.2 /
// 709.8 on either way; NaN propagates) — the softmax rows and the xi terms of the recursions
SAPR_LSE_FN double exp_unit(double x) {
  double xc = x > 800.0 ? 800.0 : x;
  xc = xc < -800.0 ? -800.0 : xc;
  const double k = SAPR_LSE_RINT(xc * 1.4426950408889634074);
  double r = SAPR_LSE_FMA(k, -6.93147180369123816490e-01, xc);
  r = SAPR_LSE_FMA(k, -1.90821492927058770002e-10, r);
  double p = kExpNegCoef[0];
#ifdef __HIPCC__
#pragma unroll
#endif
  for (int i = 1; i < 12; ++i) p = SAPR_LSE_FMA(p, r, kExpNegCoef[i]);
  p = SAPR_LSE_FMA(p, r, 1.0);
  p = SAPR_LSE_FMA(p, r, 1.0);
  return SAPR_LSE_LDEXP(p, (int)k);
}

SAPR_LSE_FN void lse2_terms(double d, double *e_out, double *inv_out, double *l1p_out) {
  const double e = exp_neg_unit(d);
  const double d1 = 1.0 + e, d2 = 2.0 + e, dd = d1 * d2;
  double y = SAPR_LSE_RCP(dd);
  double c = SAPR_LSE_FMA(-dd, y, 1.0);
  y = SAPR_LSE_FMA(y, c, y);
  c = SAPR_LSE_FMA(-dd, y, 1.0);
  y = SAPR_LSE_FMA(y, c, y);
  const double s = (e * d1) * y, z = s * s;  // s = e / (2 + e) <= 1/3
  double p = kLseAtanhCoef[0];
#ifdef __HIPCC__
#pragma unroll
#endif
  for (int i = 1; i < 16; ++i) p = SAPR_LSE_FMA(p, z, kLseAtanhCoef[i]);
  const double s2 = s + s;
  *e_out = e;
  *inv_out = d2 * y;
  *l1p_out = SAPR_LSE_FMA(s2 * z, p, s2);
}
