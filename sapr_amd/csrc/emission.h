// Device code shared by the trellis kernels (viterbi.hip, estep.hip): the diagonal-Gaussian
// log-density of hmmlearn (stats.py _log_multivariate_normal_density_diag) evaluated in numpy's
// operation order on SGPR-resident parameters, and the parameter blob built by sapr_diag_pack.
#pragma once

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

template <int D, bool SEQ>
__device__ __forceinline__ double sum_terms(const double (&q)[D]) {
  if constexpr (SEQ) {
    double quad = q[0];
#pragma unroll
    for (int d = 1; d < D; ++d) quad += q[d];
    return quad;
  }
  return np_pairwise_sum<D>(q);
}

// log-densities of one frame under all S states of the block's word model.  The parameters
// {mean, var, RN(1/var)} are wavefront-uniform, so they are fetched with explicit scalar loads
// (s_load_dwordx8, one element AHEAD of its use) and feed the fp64 VALU as SGPR operands: no
// vector-memory or LDS traffic and no VALU work for parameters.  Written as inline asm because
// hipcc otherwise merges the S*D loads into s_load_dwordx16 batches, hoists them all, overflows
// the 102 SGPRs and spills through v_writelane/v_readlane on the (saturated) vector ALU.
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

template <int D, int S, bool FASTDIV, bool SEQ, int E>
struct EmitLoop {
  static __device__ __forceinline__ void run(const double (&x)[D], const void *prm, const double *gc,
                                             i32x8 n0, i32x8 n1, double (&q)[D], double (&b)[S]) {
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
    // the empty asm pins each state's sum between the surrounding (ordered) asm runs; otherwise
    // instruction selection defers all S sums and keeps S*D quotients alive (256 VGPRs)
    q[d0] = t0;
    if constexpr (d0 == D - 1) {
      b[j0] = -0.5 * (gc[j0] + sum_terms<D, SEQ>(q));
      asm volatile("" : "+v"(b[j0]));
    }
    q[d1] = t1;
    if constexpr (d1 == D - 1) {
      b[j1] = -0.5 * (gc[j1] + sum_terms<D, SEQ>(q));
      asm volatile("" : "+v"(b[j1]));
    }
    if constexpr (E + 2 < S * D) EmitLoop<D, S, FASTDIV, SEQ, E + 2>::run(x, prm, gc, m0, m1, q, b);
  }
};

template <int D, int S, bool FASTDIV, bool SEQ>
__device__ __forceinline__ void frame_log_densities(const double (&x)[D], const double4 *__restrict__ prm,
                                                    const double *__restrict__ gc, double (&b)[S]) {
  double q[D];
  const i32x8 f0 = sload8<0>(prm), f1 = sload8<32>(prm);
  EmitLoop<D, S, FASTDIV, SEQ, 0>::run(x, prm, gc, f0, f1, q, b);
}

template <int D>
__device__ __forceinline__ void load_frame(const float *__restrict__ p, double (&x)[D]) {
  float f[D];
#pragma unroll
  for (int d = 0; d < D; ++d) f[d] = p[d];
#pragma unroll
  for (int d = 0; d < D; ++d) x[d] = static_cast<double>(f[d]);  // float32 -> float64 is exact
}

// device blob built by sapr_diag_pack (all float64):
//   prm[W][S][D][4] = {mean, var, yh = RN(1/var), yl = RN(1/var - yh)}   gconst[W][S]   log_start[W][S]   log_trans[W][S][S]
struct PackView {
  const double4 *prm;
  const double *gconst, *log_start, *log_trans;
};
__host__ __device__ inline size_t pack_doubles(int W, int S, int D) {
  return static_cast<size_t>(W) * S * D * 4 + static_cast<size_t>(W) * S * 2 + static_cast<size_t>(W) * S * S;
}
__host__ __device__ inline PackView pack_view(const void *pack, int W, int S, int D) {
  const double *b = static_cast<const double *>(pack);
  PackView v;
  v.prm = reinterpret_cast<const double4 *>(b);
  v.gconst = b + static_cast<size_t>(W) * S * D * 4;
  v.log_start = v.gconst + static_cast<size_t>(W) * S;
  v.log_trans = v.log_start + static_cast<size_t>(W) * S;
  return v;
}


}  // namespace emission
}  // namespace sapr
