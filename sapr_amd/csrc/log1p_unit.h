// log(1 + e) for 0 <= e <= 1 — the only range the two-term log-sum-exp of the lattice recursions needs
// (e = exp(-|a - b|): estep.hip lse2 / lse2_share / logaddexp, custom.hip np_logaddexp*).
// The device library's log1p covers every argument with double-double arithmetic (~110 float64 instructions per
// call, 80 of them additions); here log(1 + e) = 2 atanh(s), s = e / (2 + e) <= 1/3, as the odd series
// 2 s (1 + z/3 + z^2/5 + ...), z = s^2 <= 1/9, cut after z^16 / 33 (remainder below 2e-18 of the sum): one division and
// 18 multiply-adds.  Error < 2.5 ulp (1.98 measured) of a value <= 0.693 — scripts/verify/log1p_unit_check.c compiles THIS header on the
// CPU and compares with log1pl over the whole range (tests/test_build_guards_cpu.py) — which the sums it is added to,
// log-likelihoods of magnitude 10^2 .. 10^4, do not see.  Plain C, shared by the device code and that check.
#pragma once

#ifdef __HIPCC__
#define SAPR_LOG1P_FN __device__ __forceinline__
// (constant memory, not literals: the coefficients arrive as scalar loads hoisted out of the frame loop and feed
// v_fma_f64 as SGPR operands; as literals each costs a v_mov_b64 into the accumulator in front of its v_fmac_f64)
#define SAPR_LOG1P_TAB __constant__
#else
#define SAPR_LOG1P_FN static inline
#define SAPR_LOG1P_TAB static const
#endif

SAPR_LOG1P_TAB double kLog1pUnitCoef[16] = {1.0 / 33.0, 1.0 / 31.0, 1.0 / 29.0, 1.0 / 27.0, 1.0 / 25.0, 1.0 / 23.0, 1.0 / 21.0, 1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0, 1.0 / 7.0, 1.0 / 5.0, 1.0 / 3.0};

SAPR_LOG1P_FN double log1p_unit(double e) {
  const double s = e / (2.0 + e), z = s * s;
  double p = kLog1pUnitCoef[0];
#ifdef __HIPCC__
#pragma unroll
#endif
  for (int k = 1; k < 16; ++k) p = __builtin_fma(p, z, kLog1pUnitCoef[k]);
  const double s2 = s + s;
  return __builtin_fma(s2 * z, p, s2);
}
