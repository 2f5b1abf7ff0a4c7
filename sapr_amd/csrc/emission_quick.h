// Quick-form log-densities of the Baum-Welch E-step (estep.hip only: fb_emit_kernel, fb_forward_kernel<..., QEMIT>).
// Split from emission.h so that an edit here rebuilds one translation unit, not the thirteen that include emission.h.
#pragma once

#include "emission.h"

namespace sapr {
namespace emission {

// Baum-Welch E-step emission (estep.hip fb_emit_kernel): NF frames of one utterance per walk over the
// parameters, (x - mean)^2 * RN(1/var) accumulated by FMA — three fp64 instructions per (state, dim, frame)
// instead of the seven of the exactly rounded quotient in numpy's summation order.  The E-step's outputs are
// sums of exponentials compared at 1e-9 (north star: 1e-5 on log-likelihoods), not bit for bit; one rounding of
// the reciprocal and a different association cost ~1e-15 relative.  Scalar loads as in EmitLoop.
// (as in pair_terms_asm the arithmetic is inline assembly so that it stays next to the scalar loads it consumes)
template <bool FIRST>
__device__ __forceinline__ void quick_term_asm(double x, const double4 &p, double &acc) {
  double a;
  if constexpr (FIRST)
    asm volatile(
        "v_add_f64 %[a], %[x], -%[mu]\n\t"
        "v_mul_f64 %[a], %[a], %[a]\n\t"
        "v_mul_f64 %[acc], %[a], %[y]"
        : [a] "=&v"(a), [acc] "=&v"(acc)
        : [x] "v"(x), [mu] "s"(p.x), [y] "s"(p.z));
  else
    asm volatile(
        "v_add_f64 %[a], %[x], -%[mu]\n\t"
        "v_mul_f64 %[a], %[a], %[a]\n\t"
        "v_fma_f64 %[acc], %[a], %[y], %[acc]"
        : [a] "=&v"(a), [acc] "+v"(acc)
        : [x] "v"(x), [mu] "s"(p.x), [y] "s"(p.z));
}
template <bool FIRST>
__device__ __forceinline__ void quick_term_asm(float x, const double4 &p, double &acc) {
  double a;
  if constexpr (FIRST)
    asm volatile(
        "v_cvt_f64_f32 %[a], %[x]\n\t"
        "v_add_f64 %[a], %[a], -%[mu]\n\t"
        "v_mul_f64 %[a], %[a], %[a]\n\t"
        "v_mul_f64 %[acc], %[a], %[y]"
        : [a] "=&v"(a), [acc] "=&v"(acc)
        : [x] "v"(x), [mu] "s"(p.x), [y] "s"(p.z));
  else
    asm volatile(
        "v_cvt_f64_f32 %[a], %[x]\n\t"
        "v_add_f64 %[a], %[a], -%[mu]\n\t"
        "v_mul_f64 %[a], %[a], %[a]\n\t"
        "v_fma_f64 %[acc], %[a], %[y], %[acc]"
        : [a] "=&v"(a), [acc] "+v"(acc)
        : [x] "v"(x), [mu] "s"(p.x), [y] "s"(p.z));
}

// NF frames of one (state, dimension) element in ONE asm statement, stage by stage — NF independent chains of three
// instructions (round 4).  One statement per frame reused one temporary: a dependent chain of three per frame, one
// frame after the other, and an s_nop between every two statements (the hazard recogniser does not look inside) — at
// the one or two wavefronts per SIMD of an E-step grid nothing else filled the gaps.
template <bool FIRST>
__device__ __forceinline__ void quick_terms_asm(const double (&x)[4], const double4 &p, double (&acc)[4]) {
  double a0, a1, a2, a3;
  if constexpr (FIRST)
    asm volatile(
        "v_add_f64 %[a0], %[x0], -%[mu]\n\tv_add_f64 %[a1], %[x1], -%[mu]\n\t"
        "v_add_f64 %[a2], %[x2], -%[mu]\n\tv_add_f64 %[a3], %[x3], -%[mu]\n\t"
        "v_mul_f64 %[a0], %[a0], %[a0]\n\tv_mul_f64 %[a1], %[a1], %[a1]\n\t"
        "v_mul_f64 %[a2], %[a2], %[a2]\n\tv_mul_f64 %[a3], %[a3], %[a3]\n\t"
        "v_mul_f64 %[c0], %[a0], %[y]\n\tv_mul_f64 %[c1], %[a1], %[y]\n\t"
        "v_mul_f64 %[c2], %[a2], %[y]\n\tv_mul_f64 %[c3], %[a3], %[y]"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [c0] "=&v"(acc[0]), [c1] "=&v"(acc[1]),
          [c2] "=&v"(acc[2]), [c3] "=&v"(acc[3])
        : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [mu] "s"(p.x), [y] "s"(p.z));
  else
    asm volatile(
        "v_add_f64 %[a0], %[x0], -%[mu]\n\tv_add_f64 %[a1], %[x1], -%[mu]\n\t"
        "v_add_f64 %[a2], %[x2], -%[mu]\n\tv_add_f64 %[a3], %[x3], -%[mu]\n\t"
        "v_mul_f64 %[a0], %[a0], %[a0]\n\tv_mul_f64 %[a1], %[a1], %[a1]\n\t"
        "v_mul_f64 %[a2], %[a2], %[a2]\n\tv_mul_f64 %[a3], %[a3], %[a3]\n\t"
        "v_fma_f64 %[c0], %[a0], %[y], %[c0]\n\tv_fma_f64 %[c1], %[a1], %[y], %[c1]\n\t"
        "v_fma_f64 %[c2], %[a2], %[y], %[c2]\n\tv_fma_f64 %[c3], %[a3], %[y], %[c3]"
        : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [c0] "+v"(acc[0]), [c1] "+v"(acc[1]),
          [c2] "+v"(acc[2]), [c3] "+v"(acc[3])
        : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [mu] "s"(p.x), [y] "s"(p.z));
}

// the state's constant gconst[j] travels the same way: one s_load_dwordx2 when the state's first element is
// reached, complete at the next pair's wait (which names it as an operand), used D - 1 elements later
template <int BYTE_OFF>
__device__ __forceinline__ long long sload2(const void *base) {
  long long v;
  asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(v) : "s"(base), "n"(BYTE_OFF));
  return v;
}

// all outstanding scalar loads (they return out of order: the count can only be waited down to 0)
__device__ __forceinline__ void swait_step(i32x8 &a, i32x8 &b, i32x8 &c, i32x8 &d, long long &g) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(g));
}

// kEmitQStep (state, dimension) elements per step; the NEXT step's parameters are loaded right behind this step's wait,
// so a load has the arithmetic of a whole step to complete (scalar loads return out of order: a wait can only be for
// all of them, the prefetch distance is one step whatever the depth).  Round 4: three elements per step instead of
// two — with the one or two wavefronts per SIMD of an E-step grid nobody else covers the scalar cache's latency, and
// two elements (24 float64 instructions at four frames, 96 cycles) did not: 27 % of fb_forward_kernel's wave-cycles
// were s_waitcnt.  fb_forward_kernel<13,10> 0.638 -> 0.575 ms; four elements (0.584) spill more SGPRs than they gain.
#ifndef SAPR_EMITQ_STEP
#define SAPR_EMITQ_STEP 3
#endif
constexpr int kEmitQStep = SAPR_EMITQ_STEP;

template <int D, int S, int NF, int E>
struct EmitLoopQ {
  static constexpr int kN = S * D;
  static constexpr int G = (kN - E) >= kEmitQStep ? kEmitQStep : (kN - E);            // elements of this step
  static constexpr int GN = (kN - E - G) >= kEmitQStep ? kEmitQStep : (kN - E - G);  // of the next one (0: none)

  template <int I, class X, class Sink>
  static __device__ __forceinline__ void element(const X (&x)[NF][D], const double *gc, const i32x8 &n, long long &g,
                                                 double (&acc)[NF], Sink &sink) {
    constexpr int j = (E + I) / D, d = (E + I) % D;
    const double4 p = as_params(n);
    // the state's constant: loaded at its first element, complete at the next step's wait, used D - 1 elements later
    if constexpr (d == 0) g = sload2<8 * j>(gc);
    if constexpr (NF == 4 && std::is_same_v<X, double>) {
      const double xs[4] = {x[0][d], x[1][d], x[2][d], x[3][d]};
      quick_terms_asm<d == 0>(xs, p, acc);
    } else {
#pragma unroll
      for (int f = 0; f < NF; ++f) quick_term_asm<d == 0>(x[f][d], p, acc[f]);
    }
    if constexpr (d == D - 1) {
      const double gcj = __builtin_bit_cast(double, g);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        double b = -0.5 * (gcj + acc[f]);
        asm volatile("" : "+v"(b));
        sink(std::integral_constant<int, j>{}, f, b);
      }
    }
  }

  template <class X, class Sink>
  static __device__ __forceinline__ void run(const X (&x)[NF][D], const void *prm, const double *gc, i32x8 n0,
                                             i32x8 n1, i32x8 n2, i32x8 n3, long long g, double (&acc)[NF],
                                             Sink &sink) {
    static_assert(kEmitQStep >= 2 && kEmitQStep <= 4, "two to four elements per step");
    static_assert(D > kEmitQStep, "the constant's load needs a later step's wait");
    if constexpr (G >= 4)
      swait_step(n0, n1, n2, n3, g);
    else if constexpr (G == 3)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(n0), "+s"(n1), "+s"(n2), "+s"(g));
    else if constexpr (G == 2)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(n0), "+s"(n1), "+s"(g));
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(n0), "+s"(g));
    i32x8 m0 = n0, m1 = n1, m2 = n2, m3 = n3;
    if constexpr (GN >= 1) m0 = sload8<32 * (E + G)>(prm);
    if constexpr (GN >= 2) m1 = sload8<32 * (E + G + 1)>(prm);
    if constexpr (GN >= 3) m2 = sload8<32 * (E + G + 2)>(prm);
    if constexpr (GN >= 4) m3 = sload8<32 * (E + G + 3)>(prm);
    element<0>(x, gc, n0, g, acc, sink);
    if constexpr (G >= 2) element<1>(x, gc, n1, g, acc, sink);
    if constexpr (G >= 3) element<2>(x, gc, n2, g, acc, sink);
    if constexpr (G >= 4) element<3>(x, gc, n3, g, acc, sink);
    if constexpr (GN > 0) EmitLoopQ<D, S, NF, E + G>::run(x, prm, gc, m0, m1, m2, m3, g, acc, sink);
  }
};

template <int D, int S, int NF, class X, class Sink>
__device__ __forceinline__ void frame_log_densities_quick(const X (&x)[NF][D], const double4 *__restrict__ prm,
                                                          const double *__restrict__ gc, Sink &&sink) {
  double acc[NF];
  const i32x8 f0 = sload8<0>(prm), f1 = sload8<32>(prm);
  i32x8 f2 = f0, f3 = f1;
  if constexpr (kEmitQStep >= 3) f2 = sload8<64>(prm);
  if constexpr (kEmitQStep >= 4) f3 = sload8<96>(prm);
  EmitLoopQ<D, S, NF, 0>::run(x, prm, gc, f0, f1, f2, f3, 0ll, acc, sink);
}


}  // namespace emission
}  // namespace sapr
