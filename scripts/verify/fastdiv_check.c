/* CPU check of the exactly-rounded division used by the emission kernels (emission.h):
 *   yh = RN(1/v), yl = RN(RN(1 - v*yh) * yh)            (model preparation, once per (state, dim))
 *   t = RN(a*yl); q0 = RN(a*yh + t); r = RN(a - v*q0); q = RN(q0 + r*yh)      (4 fp64 VALU ops)
 * against the IEEE quotient a / v, on random and adversarial operands in the domain sapr_diag_pack
 * admits (v in [1e-30, 1e30]; a = d*d with |d| in {0} U [1e-46, 1e32]).
 *   gcc -O2 -ffp-contract=off -fopenmp -o /tmp/fastdiv_check scripts/verify/fastdiv_check.c -lm && /tmp/fastdiv_check 2000000000
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t rng(uint64_t *s) {
  uint64_t x = *s;
  x ^= x << 13; x ^= x >> 7; x ^= x << 17;
  return *s = x;
}
static inline double from_bits(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static inline uint64_t to_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

static inline double fast_div(double a, double v, double yh, double yl) {
  const double t = a * yl;
  const double q0 = fma(a, yh, t);
  const double r = fma(-v, q0, a);
  return fma(r, yh, q0);
}
static inline void recip(double v, double *yh, double *yl) {
  *yh = 1.0 / v;
  const double e = fma(-v, *yh, 1.0);
  *yl = e * *yh;
}

int main(int argc, char **argv) {
  const long long n = argc > 1 ? atoll(argv[1]) : 200000000LL;
  long long bad = 0, corrected = 0;
#pragma omp parallel for reduction(+ : bad, corrected) schedule(static)
  for (int th = 0; th < 64; ++th) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (th + 1);
    for (long long i = 0; i < n / 64; ++i) {
      const uint64_t r1 = rng(&s), r2 = rng(&s), r3 = rng(&s);
      /* v: random mantissa (every 8th: few set bits / all ones minus a few), exponent in [-99, 99] */
      uint64_t mv = r1 & 0xFFFFFFFFFFFFFull;
      if ((r3 & 7) == 0) mv = (r3 & 8) ? (0xFFFFFFFFFFFFFull ^ (1ull << (r3 >> 8) % 52)) : (1ull << (r3 >> 8) % 52);
      if ((r3 & 0xFF0) == 0) mv = 0xFFFFFFFFFFFFFull;
      const int ev = (int)((r1 >> 52) % 199) - 99;
      const double v = from_bits(((uint64_t)(1023 + ev) << 52) | mv);
      /* a = d*d, d from a float32-like x minus a double mean, or a raw random double */
      double a;
      if (r3 & 0x1000) {
        const float x = (float)from_bits(((uint64_t)(1023 + (int)((r2 >> 52) % 40) - 20) << 52) | (r2 & 0xFFFFFFFFFFFFFull));
        const double mu = from_bits(((uint64_t)(1023 + (int)((r3 >> 20) % 40) - 20) << 52) | (rng(&s) & 0xFFFFFFFFFFFFFull));
        const double d = (double)x - mu;
        a = d * d;
      } else {
        a = from_bits(((uint64_t)(1023 + (int)((r2 >> 52) % 400) - 200) << 52) | (r2 & 0xFFFFFFFFFFFFFull));
      }
      double yh, yl;
      recip(v, &yh, &yl);
      const double q = fast_div(a, v, yh, yl), ref = a / v;
      if (to_bits(q) != to_bits(ref)) {
        if (bad < 5) fprintf(stderr, "MISMATCH a=%a v=%a got %a want %a\n", a, v, q, ref);
        ++bad;
      }
      if (to_bits(fma(a, yh, a * yl)) != to_bits(ref)) ++corrected;
      /* the correction step itself: q0 is only ever guaranteed FAITHFUL (one of the two neighbours of
       * a/v); feed it the wrong neighbour on either side and demand the rounded quotient back */
      if (ref != 0.0 && isfinite(ref)) {
        for (int side = 0; side < 2; ++side) {
          const double qw = nextafter(ref, side ? INFINITY : -INFINITY);
          /* faithful means the exact quotient lies strictly between qw and ref */
          const double rr = fma(-v, ref, a);
          if ((side && rr <= 0) || (!side && rr >= 0)) continue;
          const double rw = fma(-v, qw, a);
          if (to_bits(fma(rw, yh, qw)) != to_bits(ref)) {
            if (bad < 5) fprintf(stderr, "CORRECTION FAILED a=%a v=%a from %a want %a\n", a, v, qw, ref);
            ++bad;
          }
        }
      }
    }
  }
  printf("cases %lld  mismatches %lld  (q0 alone wrong in %lld)\n", n / 64 * 64, bad, corrected);
  return bad != 0;
}
