// Device code shared by the trellis kernels (viterbi.hip, estep.hip): the diagonal-Gaussian
// log-density of hmmlearn (stats.py _log_multivariate_normal_density_diag) evaluated in numpy's
// operation order on SGPR-resident parameters, and the parameter blob built by sapr_diag_pack.
#pragma once

#include <type_traits>

#include "sapr_common.h"

namespace sapr {
namespace emission {

// log-density of one frame under state (w, s): -0.5 * (gconst + sum_d (x_d - mu_d)^2 / var_d)
// with numpy's evaluation order.  The order of the sum over d depends on the memory layout of
// the X array hmmlearn is handed: a C-contiguous (T,D) array (fit/score: hmmlearn_hmm.py:80-81
// concatenates) reduces pair-wise; the transposed VIEW of a (D,T) array that decoder.py:59
// passes to decode() makes numpy allocate the (T,S,D) temporary t-fastest and accumulate the D
// slices one after another (left-to-right sum) — unless T == 1, where the view is C-contiguous
// again.  `seq` selects the second behaviour (tests/test_oracle_hmmlearn.py pins the rule
// against numpy itself).
template <bool FASTDIV>
__device__ __forceinline__ double quad_term(double x, const double4 &p) {
  const double df = x - p.x;
  const double a = df * df;
  if constexpr (FASTDIV) {
    // exactly rounded a / var in four operations.  The reciprocal is carried in two words,
    // yh = RN(1/var) and yl = RN(1/var - yh): q0 = RN(a*yh + RN(a*yl)) rounds a value within 2^-104
    // (relative) of the true quotient, so it is one of its two floating-point neighbours ("faithful");
    // Markstein's correction — r = a - var*q0 is exact under FMA, q = RN(q0 + r*yh) — then yields
    // RN(a/var) (Markstein 1990; Muller et al., Handbook of Floating-Point Arithmetic, §4.7).
    // sapr_diag_pack restricts the operands to a range without under/overflow in any step;
    // scripts/verify/fastdiv_check.c checks the chain against IEEE division on 2e9 cases.
    const double t = a * p.w;
    const double q0 = __builtin_fma(a, p.z, t);
    const double r = __builtin_fma(-p.y, q0, a);
    return __builtin_fma(r, p.z, q0);
  } else {
    return a / p.y;
  }
}

// Streaming form of the two reduction orders: terms arrive for d = 0 .. D-1 and only the running
// state is kept (1 accumulator left-to-right, 8 + 1 for numpy's pair-wise scheme) instead of all D
// quotients — for D = 39 that is 60 VGPRs fewer and one more wavefront per SIMD.
//   SEQ       quad = q0; quad += q1; ...
//   pairwise  numpy's sum over a contiguous axis of n >= 8 terms: r[j] = q[j] (j < 8); r[j] += q[8i + j]
//             for the full groups of 8; ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)); then the n % 8 trailing
//             terms one by one.  n < 8: 0.0 + q0 + q1 + ...   (np_pairwise_sum in sapr_common.h is the
//             array form; tests/test_oracle_hmmlearn.py pins both against numpy itself)
template <int D, bool SEQ>
struct TermSum {
  static constexpr int kFull = D - (D % 8);
  double r[(SEQ || D < 8) ? 1 : 8];
  double res;
  template <int d>
  __device__ __forceinline__ void add(double t) {
    if constexpr (SEQ) {
      res = d == 0 ? t : res + t;
    } else if constexpr (D < 8) {
      res = d == 0 ? 0.0 + t : res + t;
    } else {
      if constexpr (d < 8) {
        r[d] = t;
      } else if constexpr (d < kFull) {
        r[d % 8] += t;
      }
      if constexpr (d == kFull - 1) res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
      if constexpr (d >= kFull) res += t;
    }
  }
};

// log-densities of one frame under all S states of the block's word model.  The parameters
// {mean, var, RN(1/var)} are wavefront-uniform, so they are fetched with explicit scalar loads
// (s_load_dwordx8, one element AHEAD of its use) and feed the fp64 VALU as SGPR operands: no
// vector-memory or LDS traffic and no VALU work for parameters.  Written as inline asm because
// hipcc otherwise merges the S*D loads into s_load_dwordx16 batches, hoists them all, overflows
// the 102 SGPRs and spills through v_writelane/v_readlane on the (saturated) vector ALU.
// The load and its s_waitcnt are separate asm statements: nothing but register allocation keeps the compiler
// from copying the destination SGPRs in between.  scripts/verify/check_sload_hazard.py scans the generated ISA
// for exactly that (every hand-written s_load, up to its wait); run it after a ROCm upgrade or an edit here.
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int BYTE_OFF>
__device__ __forceinline__ i32x8 sload8(const void *base) {
  i32x8 v;
  asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(v) : "s"(base), "n"(BYTE_OFF));
  return v;
}
__device__ __forceinline__ void swait(i32x8 &v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v)); }

__device__ __forceinline__ double as_f64(int lo, int hi) {
  return __builtin_bit_cast(double, (static_cast<unsigned long long>(static_cast<unsigned>(hi)) << 32) |
                                        static_cast<unsigned>(lo));
}

__device__ __forceinline__ double4 as_params(const i32x8 &v) {
  double4 p;
  p.x = as_f64(v[0], v[1]);
  p.y = as_f64(v[2], v[3]);
  p.z = as_f64(v[4], v[5]);
  p.w = as_f64(v[6], v[7]);
  return p;
}

// Elements (state j, dim d) are walked in pairs.  For the fast-division build the whole pair —
// wait for its parameters, start the loads of the NEXT pair, two interleaved 6-instruction fp64
// chains reading {mean, var, yh, yl} straight from SGPRs — is one run of inline assembly:
// hipcc's IR-level code motion otherwise separates the arithmetic from the loads it depends on
// and spills hundreds of SGPRs per frame.  Same IEEE operations as quad_term<true>:
//   a = (x - mean)^2; t = a*yl; q = fma(a,yh,t); r = fma(-var,q,a); q = fma(r,yh,q)
__device__ __forceinline__ void pair_terms_asm(double x0, double x1, const double4 &p0, const double4 &p1,
                                               double &t0, double &t1) {
  double a0, a1, r0, r1, q0, q1;
  asm volatile(
      "v_add_f64 %[a0], %[x0], -%[mu0]\n\t"
      "v_add_f64 %[a1], %[x1], -%[mu1]\n\t"
      "v_mul_f64 %[a0], %[a0], %[a0]\n\t"
      "v_mul_f64 %[a1], %[a1], %[a1]\n\t"
      "v_mul_f64 %[r0], %[a0], %[l0]\n\t"
      "v_mul_f64 %[r1], %[a1], %[l1]\n\t"
      "v_fma_f64 %[q0], %[a0], %[y0], %[r0]\n\t"
      "v_fma_f64 %[q1], %[a1], %[y1], %[r1]\n\t"
      "v_fma_f64 %[r0], -%[b0], %[q0], %[a0]\n\t"
      "v_fma_f64 %[r1], -%[b1], %[q1], %[a1]\n\t"
      "v_fma_f64 %[q0], %[r0], %[y0], %[q0]\n\t"
      "v_fma_f64 %[q1], %[r1], %[y1], %[q1]"
      : [a0] "=&v"(a0), [a1] "=&v"(a1), [r0] "=&v"(r0), [r1] "=&v"(r1), [q0] "=&v"(q0), [q1] "=&v"(q1)
      : [x0] "v"(x0), [x1] "v"(x1), [mu0] "s"(p0.x), [b0] "s"(p0.y), [y0] "s"(p0.z), [l0] "s"(p0.w),
        [mu1] "s"(p1.x), [b1] "s"(p1.y), [y1] "s"(p1.z), [l1] "s"(p1.w));
  t0 = q0;
  t1 = q1;
}

// `sink(std::integral_constant<int, j>, b_j)` receives each state's log-density as soon as its D terms
// are summed, so a consumer (the Viterbi column update) can use it without an S-element array.
template <int D, int S, bool FASTDIV, bool SEQ, int E>
struct EmitLoop {
  template <class Sink>
  static __device__ __forceinline__ void run(const double (&x)[D], const void *prm, const double *gc,
                                             i32x8 n0, i32x8 n1, TermSum<D, SEQ> &q, Sink &sink) {
    static_assert((S * D) % 2 == 0, "pairs");
    constexpr int j0 = E / D, d0 = E % D, j1 = (E + 1) / D, d1 = (E + 1) % D;
    swait(n0);
    swait(n1);
    const double4 p0 = as_params(n0), p1 = as_params(n1);
    i32x8 m0 = n0, m1 = n1;
    if constexpr (E + 2 < S * D) {
      m0 = sload8<32 * (E + 2)>(prm);
      m1 = sload8<32 * (E + 3)>(prm);
    }
    double t0, t1;
    if constexpr (FASTDIV) {
      pair_terms_asm(x[d0], x[d1], p0, p1, t0, t1);
    } else {
      t0 = quad_term<false>(x[d0], p0);
      t1 = quad_term<false>(x[d1], p1);
    }
    // the empty asm pins each state's log-density between the surrounding (ordered) asm runs
    q.template add<d0>(t0);
    if constexpr (d0 == D - 1) {
      double bj = -0.5 * (gc[j0] + q.res);
      asm volatile("" : "+v"(bj));
      sink(std::integral_constant<int, j0>{}, bj);
    }
    q.template add<d1>(t1);
    if constexpr (d1 == D - 1) {
      double bj = -0.5 * (gc[j1] + q.res);
      asm volatile("" : "+v"(bj));
      sink(std::integral_constant<int, j1>{}, bj);
    }
    if constexpr (E + 2 < S * D) EmitLoop<D, S, FASTDIV, SEQ, E + 2>::run(x, prm, gc, m0, m1, q, sink);
  }
};

template <int D, int S, bool FASTDIV, bool SEQ, class Sink>
__device__ __forceinline__ void frame_log_densities_each(const double (&x)[D], const double4 *__restrict__ prm,
                                                         const double *__restrict__ gc, Sink &&sink) {
  TermSum<D, SEQ> q;
  const i32x8 f0 = sload8<0>(prm), f1 = sload8<32>(prm);
  EmitLoop<D, S, FASTDIV, SEQ, 0>::run(x, prm, gc, f0, f1, q, sink);
}

template <int D, int S, bool FASTDIV, bool SEQ>
__device__ __forceinline__ void frame_log_densities(const double (&x)[D], const double4 *__restrict__ prm,
                                                    const double *__restrict__ gc, double (&b)[S]) {
  frame_log_densities_each<D, S, FASTDIV, SEQ>(x, prm, gc, [&](auto jc, double bj) { b[decltype(jc)::value] = bj; });
}

// TWO frames of one utterance against the same parameters: each s_load pair now feeds 24 fp64 instructions
// instead of 12 before the next wait.  SMEM returns out of order, so every wait is lgkmcnt(0) and the prefetch
// distance is one pair whatever the depth; doubling the work per pair is the way to lengthen it that costs no
// SGPRs (fetching three or four elements ahead does, and spilled).  It matters when a SIMD holds one or two
// wavefronts — the pruned decoder's exact pass over the surviving words spent 41 % of its wave-cycles in
// s_waitcnt — and halves the scalar-load traffic everywhere.  Only the emission is shared: the two lattice
// columns are still updated one after the other.
// float32 features are promoted inside the chain (exact) — one more instruction per element, half the registers
// for the two frames: what the 39-dimensional instantiations can afford
__device__ __forceinline__ void pair_terms_asm(float x0, float x1, const double4 &p0, const double4 &p1, double &t0,
                                               double &t1) {
  double a0, a1, r0, r1, q0, q1;
  asm volatile(
      "v_cvt_f64_f32 %[a0], %[x0]\n\t"
      "v_cvt_f64_f32 %[a1], %[x1]\n\t"
      "v_add_f64 %[a0], %[a0], -%[mu0]\n\t"
      "v_add_f64 %[a1], %[a1], -%[mu1]\n\t"
      "v_mul_f64 %[a0], %[a0], %[a0]\n\t"
      "v_mul_f64 %[a1], %[a1], %[a1]\n\t"
      "v_mul_f64 %[r0], %[a0], %[l0]\n\t"
      "v_mul_f64 %[r1], %[a1], %[l1]\n\t"
      "v_fma_f64 %[q0], %[a0], %[y0], %[r0]\n\t"
      "v_fma_f64 %[q1], %[a1], %[y1], %[r1]\n\t"
      "v_fma_f64 %[r0], -%[b0], %[q0], %[a0]\n\t"
      "v_fma_f64 %[r1], -%[b1], %[q1], %[a1]\n\t"
      "v_fma_f64 %[q0], %[r0], %[y0], %[q0]\n\t"
      "v_fma_f64 %[q1], %[r1], %[y1], %[q1]"
      : [a0] "=&v"(a0), [a1] "=&v"(a1), [r0] "=&v"(r0), [r1] "=&v"(r1), [q0] "=&v"(q0), [q1] "=&v"(q1)
      : [x0] "v"(x0), [x1] "v"(x1), [mu0] "s"(p0.x), [b0] "s"(p0.y), [y0] "s"(p0.z), [l0] "s"(p0.w),
        [mu1] "s"(p1.x), [b1] "s"(p1.y), [y1] "s"(p1.z), [l1] "s"(p1.w));
  t0 = q0;
  t1 = q1;
}

template <int D, int S, bool SEQ, int E>
struct EmitLoop2 {
  template <class X, class Sink>
  static __device__ __forceinline__ void run(const X (&xa)[D], const X (&xb)[D], const void *prm,
                                             const double *gc, i32x8 n0, i32x8 n1, TermSum<D, SEQ> &qa,
                                             TermSum<D, SEQ> &qb, Sink &sink) {
    static_assert((S * D) % 2 == 0, "pairs");
    constexpr int j0 = E / D, d0 = E % D, j1 = (E + 1) / D, d1 = (E + 1) % D;
    swait(n0);
    swait(n1);
    const double4 p0 = as_params(n0), p1 = as_params(n1);
    i32x8 m0 = n0, m1 = n1;
    if constexpr (E + 2 < S * D) {
      m0 = sload8<32 * (E + 2)>(prm);
      m1 = sload8<32 * (E + 3)>(prm);
    }
    double ta0, ta1, tb0, tb1;
    pair_terms_asm(xa[d0], xa[d1], p0, p1, ta0, ta1);
    pair_terms_asm(xb[d0], xb[d1], p0, p1, tb0, tb1);
    qa.template add<d0>(ta0);
    qb.template add<d0>(tb0);
    if constexpr (d0 == D - 1) {
      double ba = -0.5 * (gc[j0] + qa.res), bb = -0.5 * (gc[j0] + qb.res);
      asm volatile("" : "+v"(ba), "+v"(bb));
      sink(std::integral_constant<int, j0>{}, ba, bb);
    }
    qa.template add<d1>(ta1);
    qb.template add<d1>(tb1);
    if constexpr (d1 == D - 1) {
      double ba = -0.5 * (gc[j1] + qa.res), bb = -0.5 * (gc[j1] + qb.res);
      asm volatile("" : "+v"(ba), "+v"(bb));
      sink(std::integral_constant<int, j1>{}, ba, bb);
    }
    if constexpr (E + 2 < S * D) EmitLoop2<D, S, SEQ, E + 2>::run(xa, xb, prm, gc, m0, m1, qa, qb, sink);
  }
};

// NF frames per walk, each state's NF log-densities handed to the sink together as soon as their D terms are
// summed (the Viterbi column updates of the NF frames can then run state by state, frame after frame, on ONE
// lattice column with a carried predecessor per frame — no b[NF][S] arrays)
// (round 4: the state's constant gconst[j] travels like the parameters — one s_load_dwordx2 when the state's first
// element is reached, complete at the next pair's wait, used D - 1 elements later.  Left to the compiler it was a scalar
// load with its own s_waitcnt lgkmcnt(0) in the middle of a pair: a full wait for the parameter loads just issued,
// once per state.)
template <int BYTE_OFF>
__device__ __forceinline__ long long sload2_gc(const void *base) {
  long long v;
  asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(v) : "s"(base), "n"(BYTE_OFF));
  return v;
}

template <int D, int S, bool SEQ, int NF, int E>
struct EmitLoopN {
  template <class X, class Sink>
  static __device__ __forceinline__ void run(const X (&x)[NF][D], const void *prm, const double *gc, i32x8 n0,
                                             i32x8 n1, long long g, TermSum<D, SEQ> (&q)[NF], Sink &sink) {
    static_assert((S * D) % 2 == 0 && D >= 4, "pairs; the constant's load needs a later pair's wait");
    constexpr int j0 = E / D, d0 = E % D, j1 = (E + 1) / D, d1 = (E + 1) % D;
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(n0), "+s"(n1), "+s"(g));
    const double4 p0 = as_params(n0), p1 = as_params(n1);
    i32x8 m0 = n0, m1 = n1;
    if constexpr (E + 2 < S * D) {
      m0 = sload8<32 * (E + 2)>(prm);
      m1 = sload8<32 * (E + 3)>(prm);
    }
    double t0[NF], t1[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) pair_terms_asm(x[f][d0], x[f][d1], p0, p1, t0[f], t1[f]);
    if constexpr (d0 == 0) g = sload2_gc<8 * j0>(gc);
#pragma unroll
    for (int f = 0; f < NF; ++f) q[f].template add<d0>(t0[f]);
    if constexpr (d0 == D - 1) {
      const double gcj = __builtin_bit_cast(double, g);
      double b[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        b[f] = -0.5 * (gcj + q[f].res);
        asm volatile("" : "+v"(b[f]));
      }
      sink(std::integral_constant<int, j0>{}, b);
    }
    if constexpr (d1 == 0) g = sload2_gc<8 * j1>(gc);
#pragma unroll
    for (int f = 0; f < NF; ++f) q[f].template add<d1>(t1[f]);
    if constexpr (d1 == D - 1) {
      const double gcj = __builtin_bit_cast(double, g);
      double b[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        b[f] = -0.5 * (gcj + q[f].res);
        asm volatile("" : "+v"(b[f]));
      }
      sink(std::integral_constant<int, j1>{}, b);
    }
    if constexpr (E + 2 < S * D) EmitLoopN<D, S, SEQ, NF, E + 2>::run(x, prm, gc, m0, m1, g, q, sink);
  }
};

template <int D, int S, bool SEQ, int NF, class X, class Sink>
__device__ __forceinline__ void frame_log_densities_n_each(const X (&x)[NF][D], const double4 *__restrict__ prm,
                                                           const double *__restrict__ gc, Sink &&sink) {
  TermSum<D, SEQ> q[NF];
  const i32x8 f0 = sload8<0>(prm), f1 = sload8<32>(prm);
  EmitLoopN<D, S, SEQ, NF, 0>::run(x, prm, gc, f0, f1, 0ll, q, sink);
}

// log-densities of two frames (fast-division build only: the caller falls back to two single-frame walks)
template <int D, int S, bool SEQ, class X>
__device__ __forceinline__ void frame_log_densities2(const X (&xa)[D], const X (&xb)[D],
                                                     const double4 *__restrict__ prm, const double *__restrict__ gc,
                                                     double (&ba)[S], double (&bb)[S]) {
  TermSum<D, SEQ> qa, qb;
  const i32x8 f0 = sload8<0>(prm), f1 = sload8<32>(prm);
  auto sink = [&](auto jc, double a, double b) {
    ba[decltype(jc)::value] = a;
    bb[decltype(jc)::value] = b;
  };
  EmitLoop2<D, S, SEQ, 0>::run(xa, xb, prm, gc, f0, f1, qa, qb, sink);
}

// one frame of features -> float64 registers.  A frame starts at a multiple of 4*D bytes, i.e. only
// dword-aligned: the 16-byte pieces are loaded through a packed struct (unaligned dwordx4 is legal on
// gfx9) so that D = 39 costs 10 vector-memory instructions instead of 39.
struct __attribute__((packed, aligned(4))) FeatQuad {
  float a, b, c, d;
};
template <int D>
__device__ __forceinline__ void load_frame(const float *__restrict__ p, double (&x)[D]) {
  float f[D];
#pragma unroll
  for (int d = 0; d + 4 <= D; d += 4) {
    const FeatQuad v = *reinterpret_cast<const FeatQuad *>(p + d);
    f[d] = v.a;
    f[d + 1] = v.b;
    f[d + 2] = v.c;
    f[d + 3] = v.d;
  }
#pragma unroll
  for (int d = D - D % 4; d < D; ++d) f[d] = p[d];
#pragma unroll
  for (int d = 0; d < D; ++d) x[d] = static_cast<double>(f[d]);  // float32 -> float64 is exact
}

// device blob built by sapr_diag_pack:
//   prm[W][S][D][4] = {mean, var, yh = RN(1/var), yl = RN(1/var - yh)}   gconst[W][S]   log_start[W][S]   log_trans[W][S][S]
// and, for the pruned decoder's float32 bounding pass (viterbi.hip):
//   hgc[W][S] = -0.5 * gconst        wconst[W][4] = {Cmax, sum|gconst|, sum|finite log_trans|, sum|finite log_start|}
//   prm32[W][P32] pairs {float(mean), float(yh)}, P32 = S*D rounded up to a multiple of 4 (one s_load_dwordx8
//   fetches four pairs), where Cmax = max_s sum_d mean^2 / var;
// then, 16-byte aligned, the operands of the matrix-core bounding pass (gfrag, gctr, gkw below).
struct PackView {
  const double4 *prm;
  const double *gconst, *log_start, *log_trans;
  const double *hgc, *wconst;
  const double *prm32;  // [W][P32] 8-byte slots {float mu, float y}
  // operands of the matrix-core bounding pass (viterbi_bound.hip: viterbi_bound_lds_kernel)
  const uint4 *gfrag;   // [W][RT][KC][hi, lo][64 lanes] 8 halves each: A fragments of v_mfma_f32_16x16x32_f16
  const float *gctr;    // [3][8 G]: centre subtracted from the features, then the power-of-two factors that bring
                        // x' into half range for the squared and for the linear slots (zeros / ones past D)
  const double *gkw;    // [W] per-word constant of the bound, [W] max_j sum_k |P 2^g|, then {2^g, 2^-g, scratch}
  const double *gR;     // [2][W][S]: R_j = sum_(i<=j) (lt_(i-1)i - sg_(i-1)), the forward weights the bounding lattice
                        // has divided out (v_j = u_j - R_j), then the same with -inf for the unreachable tail states
};
__host__ __device__ inline int pack_p32(int S, int D) { return (S * D + 3) / 4 * 4; }
// expanded feature vector phi = [x'^2 (D slots), 1, 0.. | x' (D slots), 0..] in groups of 8 slots:
// G groups per half, KC chunks of 32 slots (4 groups), RT tiles of 16 states
__host__ __device__ constexpr int gemm_groups(int D) { return (D + 1 + 7) / 8; }
__host__ __device__ constexpr int gemm_kchunks(int D) { return (2 * gemm_groups(D) + 3) / 4; }
__host__ __device__ constexpr int gemm_rtiles(int S) { return (S + 15) / 16; }
__host__ __device__ inline size_t gemm_frag_doubles(int W, int S, int D) {
  return static_cast<size_t>(W) * gemm_rtiles(S) * gemm_kchunks(D) * 2 * 128;  // 1 KiB per fragment
}
__host__ __device__ inline size_t pack_gemm_offset(int W, int S, int D) {
  const size_t before = static_cast<size_t>(W) * S * D * 4 + static_cast<size_t>(W) * S * 2 +
                        static_cast<size_t>(W) * S * S + static_cast<size_t>(W) * S + static_cast<size_t>(W) * 4 +
                        static_cast<size_t>(W) * pack_p32(S, D);
  return (before + 1) & ~static_cast<size_t>(1);  // 16-byte aligned fragments
}
__host__ __device__ inline size_t pack_doubles(int W, int S, int D) {
  return pack_gemm_offset(W, S, D) + gemm_frag_doubles(W, S, D) + static_cast<size_t>(gemm_groups(D)) * 12 +
         2 * static_cast<size_t>(W) + 3 + 2 * static_cast<size_t>(W) * S;
}
__host__ __device__ inline PackView pack_view(const void *pack, int W, int S, int D) {
  const double *b = static_cast<const double *>(pack);
  PackView v;
  v.prm = reinterpret_cast<const double4 *>(b);
  v.gconst = b + static_cast<size_t>(W) * S * D * 4;
  v.log_start = v.gconst + static_cast<size_t>(W) * S;
  v.log_trans = v.log_start + static_cast<size_t>(W) * S;
  v.hgc = v.log_trans + static_cast<size_t>(W) * S * S;
  v.wconst = v.hgc + static_cast<size_t>(W) * S;
  v.prm32 = v.wconst + static_cast<size_t>(W) * 4;
  const double *g = b + pack_gemm_offset(W, S, D);
  v.gfrag = reinterpret_cast<const uint4 *>(g);
  v.gctr = reinterpret_cast<const float *>(g + gemm_frag_doubles(W, S, D));
  v.gkw = g + gemm_frag_doubles(W, S, D) + static_cast<size_t>(gemm_groups(D)) * 12;
  v.gR = v.gkw + 2 * static_cast<size_t>(W) + 3;
  return v;
}

// ---- float32 bounding pass --------------------------------------------------------------------------
// q32_j = sum_d (x_d - float(mean))^2 * float(1/var) in float32, three VALU instructions per (state, dim)
// at the float32 rate, parameters again straight from SGPRs (four {mean, 1/var} pairs per s_load_dwordx8,
// one group ahead).  NOT the reference's arithmetic: viterbi.hip turns it into an interval that contains the
// exact score, and the exact kernel then runs only where intervals overlap.
template <int D>
__device__ __forceinline__ void load_frame_f32(const float *__restrict__ p, float (&f)[D]) {
#pragma unroll
  for (int d = 0; d + 4 <= D; d += 4) {
    const FeatQuad v = *reinterpret_cast<const FeatQuad *>(p + d);
    f[d] = v.a;
    f[d + 1] = v.b;
    f[d + 2] = v.c;
    f[d + 3] = v.d;
  }
#pragma unroll
  for (int d = D - D % 4; d < D; ++d) f[d] = p[d];
}

__device__ __forceinline__ float as_f32(int v) { return __builtin_bit_cast(float, v); }

template <int D, int S, int E>
struct ApproxLoop {
  // `cur` holds the parameters of the group of four elements E belongs to; `nxt` is the next group's load,
  // in flight since this group started
  template <class Sink>
  static __device__ __forceinline__ void run(const float (&x)[D], const void *prm, i32x8 cur, i32x8 nxt, float acc,
                                             Sink &sink) {
    constexpr int j = E / D, d = E % D, slot = E % 4;
    if constexpr (slot == 0) {
      swait(nxt);
      cur = nxt;
      if constexpr (E + 4 < S * D) nxt = sload8<8 * (E + 4)>(prm);
    }
    const float mu = as_f32(cur[2 * slot]), y = as_f32(cur[2 * slot + 1]);
    float a;
    if constexpr (d == 0) {
      asm volatile(
          "v_sub_f32 %[a], %[x], %[mu]\n\t"
          "v_mul_f32 %[a], %[a], %[a]\n\t"
          "v_mul_f32 %[acc], %[a], %[y]"
          : [a] "=&v"(a), [acc] "=&v"(acc)
          : [x] "v"(x[d]), [mu] "s"(mu), [y] "s"(y));
    } else {
      asm volatile(
          "v_sub_f32 %[a], %[x], %[mu]\n\t"
          "v_mul_f32 %[a], %[a], %[a]\n\t"
          "v_fmac_f32 %[acc], %[a], %[y]"
          : [a] "=&v"(a), [acc] "+v"(acc)
          : [x] "v"(x[d]), [mu] "s"(mu), [y] "s"(y));
    }
    if constexpr (d == D - 1) sink(std::integral_constant<int, j>{}, acc);
    if constexpr (E + 1 < S * D) ApproxLoop<D, S, E + 1>::run(x, prm, cur, nxt, acc, sink);
  }
};

template <int D, int S, class Sink>
__device__ __forceinline__ void frame_quads_f32_each(const float (&x)[D], const void *prm32, Sink &&sink) {
  const i32x8 f0 = sload8<0>(prm32);
  ApproxLoop<D, S, 0>::run(x, prm32, f0, f0, 0.0f, sink);
}

}  // namespace emission
}  // namespace sapr
