// Kernels for the reference's from-scratch HMM (assignment2/custom_hmm.py) on gfx950.
//
// This is the SECONDARY path of the recogniser (decoder.py defaults to the hmmlearn models); it is
// implemented for parity with every quirk of the reference, not tuned: one lane owns one
// (utterance, model) problem, S and D are run-time values (S <= 20, D <= 40), lattices live in HBM.
// CPU restatement: oracle/custom_hmm_oracle.py (pinned to the imported reference).
//
//   custom_estep_kernel   custom_hmm.py:146-322 — emission (the Gram-matrix ROW-SUM "Mahalanobis"
//                         term of :168-172, evaluated as d_t . (C^-1 sum_s d_s)), forward with
//                         non-emitting entry/exit states and the global max-shift (:176-211),
//                         backward (:213-246), gamma (:248-257), per-frame renormalised xi
//                         (:259-322) and the per-utterance sums baum_welch accumulates (:434-439).
//   custom_decode_kernel  custom_hmm.py:462-514 — Viterbi over the first `Tq` frames
//                         (Tq = features.shape[0] = D, the reference's quirk), strict '>' from -inf.
//   custom_update_b_*     custom_hmm.py:366-400 — two-pass means / full covariances.
//   custom_global_*       custom_hmm.py:70-92  — flat-start sums.
//
// float64; exp / log from the device math library, exp / reciprocal / log(1 + e) of the two-term logaddexp from lse_unit.h (agreement
// with numpy ~1e-13, tests use 1e-9).
#include <type_traits>

#include <cstdlib>
#include <cstring>

#include "sapr_common.h"

namespace sapr {
namespace {

constexpr int kMaxS = 20, kMaxD = 40;
constexpr int kBlock = 64;

#include "lse_unit.h"

// numpy's logaddexp (npy_logaddexp)
// np.logaddexp(a, b) together with the shares of its two arguments in the sum, exp(a - r) and exp(b - r), from the
// exponential it evaluates anyway (one division instead of two more exponentials)
__device__ __forceinline__ double np_logaddexp_shares(double a, double b, double &share_a, double &share_b) {
  // npy_logaddexp's three cases (a == b; a > b: a + log1p(exp(b - a)); a <= b: b + log1p(exp(a - b)); NaN otherwise) as
  // ONE evaluation with selects: the lanes of a wavefront are different utterances and disagree about which argument
  // is the larger, so as branches both sides ran for every call (16 exponentials and 32 divisions per frame of the
  // forward loop at 10 states instead of 8 and 16).  exp(-|a - b|) is the argument either branch would pass; NaN
  // runs through the arithmetic to the result and to both shares.
  const double tmp = a - b;
  double e, inv, l1p;  // exp(-|a - b|), 1 / (1 + e), log(1 + e) from one short chain (lse_unit.h, round 4)
  lse2_terms(fabs(tmp), &e, &inv, &l1p);
  const double small = e * inv;
  const bool a_larger = tmp > 0;
  const bool same = a == b;  // handles inf == inf
  const double r = (a_larger ? a : b) + l1p;
  share_a = same ? 0.5 : (a_larger ? inv : small);
  share_b = same ? 0.5 : (a_larger ? small : inv);
  return same ? a + 0.693147180559945309417232121458176568 : r;
}
__device__ __forceinline__ double np_logaddexp(double a, double b) {
  const double tmp = a - b;
  double e, inv, l1p;
  lse2_terms(fabs(tmp), &e, &inv, &l1p);
  const double r = (tmp > 0 ? a : b) + l1p;
  return a == b ? a + 0.693147180559945309417232121458176568 : r;  // handles inf == inf
}

// numpy pair-wise sum of n strided values: blocks of <= 128 with 8 accumulators, recursive halving
// above.  The recursion is unrolled at compile time (depth 8: n <= 32768) — device code with a real
// recursive call needs a dynamic stack, which this library avoids.
template <typename T>
__device__ __forceinline__ T np_pairwise_block(const T *p, int n, int stride) {
  if (n < 8) {
    T r = 0;
    for (int i = 0; i < n; ++i) r += p[static_cast<int64_t>(i) * stride];
    return r;
  }
  T r[8];
  for (int j = 0; j < 8; ++j) r[j] = p[static_cast<int64_t>(j) * stride];
  int i;
  for (i = 8; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; ++j) r[j] += p[static_cast<int64_t>(i + j) * stride];
  T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += p[static_cast<int64_t>(i) * stride];
  return res;
}
template <typename T, int DEPTH>
__device__ T np_pairwise(const T *p, int n, int stride) {
  if constexpr (DEPTH == 0) {
    return np_pairwise_block<T>(p, n, stride);
  } else {
    if (n <= 128) return np_pairwise_block<T>(p, n, stride);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise<T, DEPTH - 1>(p, n2, stride) +
           np_pairwise<T, DEPTH - 1>(p + static_cast<int64_t>(n2) * stride, n - n2, stride);
  }
}
__device__ double np_pairwise_rt(const double *p, int n) { return np_pairwise<double, 3>(p, n, 1); }

template <int B, int E, class F>
__device__ __forceinline__ void static_for_c(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for_c<B + 1, E>(f);
  }
}

struct CustomPack {
  const double *means;   // [W][S][D]
  const double *inv;     // [W][S][D][D]  inverse of (cov + 1e-6 I), numpy/LAPACK on the host
  const double *cterm;   // [W][S]        D*log(2*pi) + logdet
  const double *A;       // [W][S][S]
  const double *logA;    // [W][S][S]     np.log(A) (−inf for zeros)
};

// A (T, S) lattice of one utterance: element (t, j) lives at p[(t * S + j) * es].  es = 1 is the
// reference's per-utterance row layout (what the per-method API hands out); es = n_slots with p offset
// by the utterance's slot is the lane-contiguous layout the batched E-step uses internally, where the
// 64 lanes of a wavefront touch 64 consecutive doubles (one 512-byte row) instead of 64 cache lines.
struct Lat {
  double *p;
  int64_t es;
  int S;
  __device__ __forceinline__ double &at(int t, int j) const { return p[(static_cast<int64_t>(t) * S + j) * es]; }
};

// E[t][j] for all frames of one utterance under model w
__device__ void emission_rows(const float *__restrict__ x, int T, int D, int S, const CustomPack &P, int w,
                              const Lat E) {
  double xs[kMaxD], v[kMaxD];
  for (int d = 0; d < D; ++d) xs[d] = 0.0;
  for (int t = 0; t < T; ++t)
    for (int d = 0; d < D; ++d) xs[d] += static_cast<double>(x[static_cast<int64_t>(t) * D + d]);
  for (int t = 0; t < T; ++t) {
    E.at(t, 0) = neg_inf();
    E.at(t, S - 1) = neg_inf();
  }
  for (int j = 1; j < S - 1; ++j) {
    const double *mu = P.means + (static_cast<int64_t>(w) * S + j) * D;
    const double *iv = P.inv + (static_cast<int64_t>(w) * S + j) * D * D;
    for (int a = 0; a < D; ++a) {
      double acc = 0.0;
      for (int b = 0; b < D; ++b) acc += iv[a * D + b] * (xs[b] - T * mu[b]);
      v[a] = acc;
    }
    const double c = P.cterm[static_cast<int64_t>(w) * S + j];
    for (int t = 0; t < T; ++t) {
      double qd = 0.0;
      for (int d = 0; d < D; ++d) qd += (static_cast<double>(x[static_cast<int64_t>(t) * D + d]) - mu[d]) * v[d];
      E.at(t, j) = -0.5 * (c + qd);
    }
  }
}

// ---- evaluation-order-faithful emission ---------------------------------------------------------------
// custom_hmm.py:168-172 evaluates  np.sum(diff.T @ inv_cov @ diff, axis=1):  two BLAS products and a numpy
// row sum.  On the build the golden vectors were taken from (numpy 2.2 / OpenBLAS 0.3.29 SkylakeX) every
// element of both products is ONE fused-multiply-add chain over the contraction index in increasing order,
// starting from 0 — checked bit for bit against the reference's emission matrices in
// tests/test_oracle_custom.py — and np.sum over the contiguous rows of the (T,T) Gram matrix is numpy's
// pair-wise sum.  This kernel performs exactly those operations:
//     M1[t][c] = fma-chain_k diff[k][t] * inv[k][c]          (diff.T @ inv_cov)
//     G[t][s]  = fma-chain_k M1[t][k]  * diff[k][s]          ((...) @ diff)
//     E[t][j]  = -0.5 * ((D log 2pi + logdet) + pairwise_s G[t][s])
// so that decisions that hang on the last bit (flat-start ties in decode) come out as in the reference.
// One thread owns one row (model w, state j, frame t); a workgroup owns one utterance, so the feature
// addresses inside the s / k loops are wavefront-uniform.  Cost O(rows * T * D) per state: decode needs the
// first Tq = D rows only (custom_hmm.py:466), the per-method API all T.
// A leaf of the pair-wise sum holds at most 128 frames; the workgroup stages them in LDS as float64
// (numpy promotes the float32 features before the subtraction) and every thread walks them in lockstep.
constexpr int kLeaf = 128;

template <int DC>
struct ExactRow {
  static constexpr int kCap = DC ? DC : kMaxD;
  double mu[kCap], m1[kCap];
  int D;

  // G[t][s0 .. s0+N) of the staged frames (xd[k][kLeaf], feature k of frame s at xd[k * kLeaf + s]): N
  // independent chains, each a fused-multiply-add chain over k in increasing order
  template <int N>
  __device__ __forceinline__ void gram(const double *xd, int s0, double (&g)[N]) const {
#pragma unroll
    for (int j = 0; j < N; ++j) g[j] = 0.0;
#pragma unroll
    for (int k = 0; k < kCap; ++k)
      if (DC || k < D) {
#pragma unroll
        for (int j = 0; j < N; ++j) g[j] = fma(m1[k], xd[k * kLeaf + s0 + j] - mu[k], g[j]);
      }
  }
  // numpy DOUBLE_pairwise_sum over the n <= 128 staged frames
  __device__ __forceinline__ double leaf(const double *xd, int n) const {
    double g[8];
    if (n < 8) {
      double res = 0.0;
      for (int i = 0; i < n; ++i) {
        double g1[1];
        gram<1>(xd, i, g1);
        res += g1[0];
      }
      return res;
    }
    double r[8];
    gram<8>(xd, 0, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = g[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
      gram<8>(xd, i, g);
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] += g[j];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    if (i < n) {
      // the n % 8 trailing frames as one more block of eight: the columns behind frame n - 1 hold whatever the
      // previous leaf left there (the array has kLeaf columns and i + 8 <= kLeaf), their chains are discarded
      gram<8>(xd, i, g);
#pragma unroll
      for (int j = 0; j < 7; ++j)
        if (i + j < n) res += g[j];
    }
    return res;
  }
};

// grid.x = utterance, grid.y = chunk of 256 rows.
// n_rows > 0: E[((u * W + w) * n_rows + t) * S + j] for t < n_rows (decode: n_rows = Tq <= T);
// n_rows == 0: every frame, E[(offsets[u] + t) * S + j] (W must be 1: the per-method API).
template <int DC>
__global__ __launch_bounds__(256, DC == 13 ? 4 : 2) void custom_emission_exact_kernel(
    const float *__restrict__ feats, const int64_t *__restrict__ offsets, int W, int D, int S, int n_rows,
    CustomPack P, double *__restrict__ E) {
  extern __shared__ double xd[];  // [D][kLeaf]: feature k of the leaf's frame s at xd[k * kLeaf + s]
  const int64_t u = blockIdx.x;
  const int64_t beg = offsets[u];
  const int T = static_cast<int>(offsets[u + 1] - beg);
  const int rows = n_rows ? n_rows : T;
  const int Dn = DC ? DC : D;
  const int n_emit = S - 2;
  const int tasks = W * n_emit * rows;
  if (static_cast<int64_t>(blockIdx.y) * 256 >= tasks) return;  // whole workgroup: no barrier is skipped
  const float *xu = feats + beg * Dn;
  const int task = blockIdx.y * 256 + threadIdx.x;
  const bool live = task < tasks;
  const int t = live ? task % rows : 0;
  const int wj = live ? task / rows : 0;
  const int j = 1 + wj % n_emit, w = wj / n_emit;
  // decode asks for the first n_rows = D rows of every utterance (custom_hmm.py:466); an utterance SHORTER than that
  // (the Python mirror raises the reference's IndexError first, a C-ABI caller may not) must not read its neighbour's
  // frames: rows t >= T re-read a valid frame and are written as -inf
  const bool inside = t < T;
  const int tr = inside ? t : (T > 0 ? T - 1 : 0);
  ExactRow<DC> R;
  R.D = Dn;
  {
    const double *mu = P.means + (static_cast<int64_t>(w) * S + j) * Dn;
    const double *iv = P.inv + (static_cast<int64_t>(w) * S + j) * Dn * Dn;
    double drow[ExactRow<DC>::kCap];
#pragma unroll
    for (int k = 0; k < ExactRow<DC>::kCap; ++k)
      if (DC || k < Dn) {
        R.mu[k] = mu[k];
        drow[k] = (T > 0 ? static_cast<double>(xu[static_cast<int64_t>(tr) * Dn + k]) : 0.0) - R.mu[k];
      }
#pragma unroll
    for (int c = 0; c < ExactRow<DC>::kCap; ++c)
      if (DC || c < Dn) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < ExactRow<DC>::kCap; ++k)
          if (DC || k < Dn) acc = fma(drow[k], iv[k * Dn + c], acc);
        R.m1[c] = acc;
      }
  }
  // numpy's recursive halving above 128 terms as an explicit post-order walk; T is the same for every
  // thread of the workgroup, so the walk (and its barriers) is uniform
  constexpr int kDepth = 24;
  int st_s[kDepth], st_n[kDepth], st_ph[kDepth];
  double val[kDepth];
  int top = 1, vtop = 0;
  st_s[0] = 0;
  st_n[0] = T;
  st_ph[0] = 0;
  while (top > 0) {
    const int i = top - 1;
    if (st_n[i] <= kLeaf) {
      const int n = st_n[i];
      const float *src = xu + static_cast<int64_t>(st_s[i]) * Dn;
      __syncthreads();  // the previous leaf has been consumed
      for (int e = threadIdx.x; e < n * Dn; e += 256) {
        const int k = e / n, fs = e - k * n;
        xd[k * kLeaf + fs] = static_cast<double>(src[fs * Dn + k]);
      }
      __syncthreads();
      val[vtop++] = R.leaf(xd, n);
      --top;
      continue;
    }
    int n2 = st_n[i] / 2;
    n2 -= n2 % 8;
    if (st_ph[i] == 0) {
      st_ph[i] = 1;
      st_s[top] = st_s[i];
      st_n[top] = n2;
      st_ph[top] = 0;
      ++top;
    } else if (st_ph[i] == 1) {
      st_ph[i] = 2;
      st_s[top] = st_s[i] + n2;
      st_n[top] = st_n[i] - n2;
      st_ph[top] = 0;
      ++top;
    } else {
      val[vtop - 2] = val[vtop - 2] + val[vtop - 1];
      --vtop;
      --top;
    }
  }
  if (!live) return;
  const double e = inside ? -0.5 * (P.cterm[static_cast<int64_t>(w) * S + j] + val[0]) : neg_inf();
  double *row = n_rows ? E + ((u * W + w) * static_cast<int64_t>(n_rows) + t) * S : E + (beg + t) * S;
  row[j] = e;
  if (j == 1) {  // the non-emitting columns of this row
    row[0] = neg_inf();
    row[S - 1] = neg_inf();
  }
}

// ---- the same rows, one per lane, differences shared across the 16 lanes of a DPP row (13 dimensions) ------------
// The difference x[k][s] - mu[k] belongs to the (word, state), not to the row: with the 13 (decode) or 16 (all-frames
// mode) rows of one (word, state) on the 16 lanes of a DPP row, lane k computes the eight differences of dimension k
// for a numpy block ONCE and every lane takes them as the broadcast operand of its multiply-add
//     v_fmac_f64_dpp g[f], d[f] row_newbcast:k, m1[k]        (full rate on this chip: scripts/ubench/mix_rate)
// — per row and block 104 multiply-adds + 8 subtractions + 8 running-sum additions (two rows per thread: 164) and 8
// LDS reads (130).  a * b + c is the same fused operation whichever factor comes first, the chains run over k in
// increasing order from 0: the bits are those of ExactRow.  M1 shares the inverse covariance the same way (lane c loads
// inv[k][c], 13 loads per lane instead of 169).
constexpr int kRowStride = kLeaf + 1;  // staged frames xd[k][kRowStride]: dimension k + 1 starts two banks further

template <int L>
__device__ __forceinline__ void fmac8_bcast(double (&g)[8], const double (&d)[8], double m) {
  // s_nop: a DPP operand written by the preceding VALU instruction needs two wait states
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %8, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %9, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %2, %10, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %3, %11, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %4, %12, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %5, %13, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %6, %14, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %7, %15, %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf"
      : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7])
      : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(m), "n"(L));
}
// acc += (src of lane L of the row) * m
template <int L>
__device__ __forceinline__ void fmac_bcast(double &acc, double src, double m) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(m), "n"(L));
}
template <int L>
__device__ __forceinline__ double mov_bcast(double src) {
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(L));
  return r;
}

// D = 13: the 13 dimensions on lanes 0..12 of the row.  D = 39: lane l computes dimensions l, l + 16 and l + 32 (three
// difference arrays; dimension k is read from lane k % 16, array k / 16), 16 of the 39 rows per DPP row.
template <int DD>
struct ExactRowBcast {
  static constexpr int kD = DD;
  static constexpr int kNL = (DD + 15) / 16;  // dimensions per lane
  double m1[kD];
  double mu_own[kNL];  // mean of dimension min((lane & 15) + 16 q, D - 1) of this row's (word, state)
  int x_own[kNL];      // index of that dimension's staged frames: dimension * kRowStride

  template <int K>
  __device__ __forceinline__ void chain(double (&g)[8], const double (&d)[kNL][8]) const {
    fmac8_bcast<K % 16>(g, d[K / 16], m1[K]);
    if constexpr (K + 1 < kD) chain<K + 1>(g, d);
  }
  // G[t][s0 .. s0 + 8) of the staged frames
  __device__ __forceinline__ void block(const double *xd, int s0, double (&g)[8]) const {
    double d[kNL][8];
#pragma unroll
    for (int q = 0; q < kNL; ++q) {
#pragma unroll
      for (int f = 0; f < 8; ++f) d[q][f] = xd[x_own[q] + s0 + f] - mu_own[q];
    }
#pragma unroll
    for (int f = 0; f < 8; ++f) g[f] = 0.0;
    chain<0>(g, d);
  }
  template <int K>
  __device__ __forceinline__ void chain1(double &g, const double *xd, int i) const {
    const double mu_k = mov_bcast<K % 16>(mu_own[K / 16]);
    g = fma(m1[K], xd[K * kRowStride + i] - mu_k, g);
    if constexpr (K + 1 < kD) chain1<K + 1>(g, xd, i);
  }
  // numpy DOUBLE_pairwise_sum over the n <= 128 staged frames
  __device__ __forceinline__ double leaf(const double *xd, int n) const {
    if (n < 8) {
      double res = 0.0;
      for (int i = 0; i < n; ++i) {
        double g = 0.0;
        chain1<0>(g, xd, i);
        res += g;
      }
      return res;
    }
    double r[8], g[8];
    block(xd, 0, r);
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
      block(xd, i, g);
#pragma unroll
      for (int f = 0; f < 8; ++f) r[f] += g[f];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    if (i < n) {
      // the n % 8 trailing frames as one more block of eight: the columns behind frame n - 1 hold whatever was
      // staged there before (a row has kRowStride columns and i + 8 <= kLeaf), their chains are discarded
      block(xd, i, g);
#pragma unroll
      for (int f = 0; f < 7; ++f)
        if (i + f < n) res += g[f];
    }
    return res;
  }
  // M1[c] += da * inv[k][c] for every column c: lane c % 16 of the row holds inv[k][c] in iv_k[c / 16]
  template <int C>
  __device__ __forceinline__ void m1_row(const double (&iv_k)[kNL], double da) {
    fmac_bcast<C % 16>(m1[C], iv_k[C / 16], da);
    if constexpr (C + 1 < kD) m1_row<C + 1>(iv_k, da);
  }
};

template <int DD>
__device__ __forceinline__ void bcast_stage_leaf(double *xd, const float *src, int n) {
  for (int fs = threadIdx.x; fs < n; fs += blockDim.x) {
#pragma unroll
    for (int k = 0; k < DD; ++k) xd[k * kRowStride + fs] = static_cast<double>(src[fs * DD + k]);
  }
}

// numpy's recursive halving above 128 terms as an explicit post-order walk, every leaf staged in turn; T is the same
// for every thread of the workgroup, so the walk (and its barriers) is uniform.  (Its own function: inlined, its
// scratch arrays and the second copy of the leaf cost the one-leaf path its registers.)
template <int DD>
__device__ __noinline__ double bcast_rows_walk(const ExactRowBcast<DD> &R, double *xd, const float *xu, int T) {
  constexpr int kDepth = 24;
  int st_s[kDepth], st_n[kDepth], st_ph[kDepth];
  double val[kDepth];
  int top = 1, vtop = 0;
  st_s[0] = 0;
  st_n[0] = T;
  st_ph[0] = 0;
  while (top > 0) {
    const int i = top - 1;
    if (st_n[i] <= kLeaf) {
      __syncthreads();  // the previous leaf has been consumed
      bcast_stage_leaf<DD>(xd, xu + static_cast<int64_t>(st_s[i]) * DD, st_n[i]);
      __syncthreads();
      val[vtop++] = R.leaf(xd, st_n[i]);
      --top;
      continue;
    }
    int n2 = st_n[i] / 2;
    n2 -= n2 % 8;
    if (st_ph[i] == 0) {
      st_ph[i] = 1;
      st_s[top] = st_s[i];
      st_n[top] = n2;
      st_ph[top] = 0;
      ++top;
    } else if (st_ph[i] == 1) {
      st_ph[i] = 2;
      st_s[top] = st_s[i] + n2;
      st_n[top] = st_n[i] - n2;
      st_ph[top] = 0;
      ++top;
    } else {
      val[vtop - 2] = val[vtop - 2] + val[vtop - 1];
      --vtop;
      --top;
    }
  }
  return val[0];
}

// One workgroup per utterance.  A group = the 16 lanes of a DPP row = rows 16 q .. 16 q + 15 of one (word w, state j):
// group (w * n_emit + j - 1) * ceil(rows / 16) + q; the workgroup walks the groups blockDim / 16 at a time.  An
// utterance of at most kLeaf frames (one leaf) is staged once for all of them.  Outputs as custom_emission_exact_kernel.
// Every lane stays active through the arithmetic (a broadcast reads its source lane whatever that lane's own row is
// worth): lanes without a row work on a clamped copy and do not store.
// 13 dimensions: 128 threads, four wavefronts per SIMD; 39: 256 threads (40 KB of staged frames), two per SIMD.
template <int DD>
__global__ __launch_bounds__(DD > 16 ? 256 : 128, DD > 16 ? 2 : 4) void custom_emission_bcast_kernel(
    const float *__restrict__ feats, const int64_t *__restrict__ offsets, int W, int S, int n_rows, CustomPack P,
    double *__restrict__ E) {
  using Row = ExactRowBcast<DD>;
  constexpr int Dn = DD, NL = Row::kNL;
  __shared__ double xd[Dn * kRowStride];
  const int64_t u = blockIdx.x;
  const int64_t beg = offsets[u];
  const int T = static_cast<int>(offsets[u + 1] - beg);
  const int rows = n_rows ? n_rows : T;
  const int rgs = (rows + 15) / 16;
  const int n_emit = S - 2;
  const int n_groups = W * n_emit * rgs;
  const float *xu = feats + beg * Dn;
  const bool single = T <= kLeaf;
  if (single) {
    bcast_stage_leaf<DD>(xd, xu, T);
    __syncthreads();
  }
  const int l16 = threadIdx.x & 15;
  const int groups_per_pass = static_cast<int>(blockDim.x) / 16;
  for (int base = 0; base < n_groups; base += groups_per_pass) {
    const int gi = base + (threadIdx.x >> 4);
    const bool live_g = gi < n_groups;
    const int gc = live_g ? gi : n_groups - 1;
    const int pr = gc / rgs, t = (gc - pr * rgs) * 16 + l16;
    const int j = 1 + pr % n_emit, w = pr / n_emit;
    const bool live = live_g && t < rows;
    // rows t >= T (an utterance shorter than the n_rows decode asks for) re-read a valid frame and are written as -inf
    const bool inside = t < T;
    const int tr = inside ? t : (T > 0 ? T - 1 : 0);
    Row R;
    const double *mu_g = P.means + (static_cast<int64_t>(w) * S + j) * Dn;
    const double *iv = P.inv + (static_cast<int64_t>(w) * S + j) * Dn * Dn;
    int lk[NL];
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      lk[q] = min(l16 + 16 * q, Dn - 1);
      R.mu_own[q] = mu_g[lk[q]];
      R.x_own[q] = lk[q] * kRowStride;
    }
#pragma unroll
    for (int c = 0; c < Dn; ++c) R.m1[c] = 0.0;
    // M1[c] = chain_k (x[tr][k] - mu[k]) inv[k][c]: row k of the inverse covariance sits on the lanes of the DPP row
    if constexpr (Dn <= 16) {
      double ivk[Dn], x[Dn];
#pragma unroll
      for (int k = 0; k < Dn; ++k) {
        ivk[k] = iv[k * Dn + lk[0]];
        x[k] = T <= 0 ? 0.0 : single ? xd[k * kRowStride + tr] : static_cast<double>(xu[static_cast<int64_t>(tr) * Dn + k]);
      }
      static_for_c<0, Dn>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const double da = x[k] - mov_bcast<k>(R.mu_own[0]);
        const double row[1] = {ivk[k]};
        R.template m1_row<0>(row, da);
      });
    } else {
      // (rolled: 39 x 3 loads would not stay in registers; the next row is in flight while this one is consumed)
      double row[NL];
#pragma unroll
      for (int q = 0; q < NL; ++q) row[q] = iv[lk[q]];
#pragma unroll 1
      for (int k = 0; k < Dn; ++k) {
        double nrow[NL];
        const double *nx = iv + min(k + 1, Dn - 1) * Dn;
#pragma unroll
        for (int q = 0; q < NL; ++q) nrow[q] = nx[lk[q]];
        const double xk = T <= 0 ? 0.0 : single ? xd[k * kRowStride + tr] : static_cast<double>(xu[static_cast<int64_t>(tr) * Dn + k]);
        const double da = xk - mu_g[k];
        R.template m1_row<0>(row, da);
#pragma unroll
        for (int q = 0; q < NL; ++q) row[q] = nrow[q];
      }
    }
    double res;
    if (single) {
      res = R.leaf(xd, T);
    } else {
      const Row Rc = R;
      res = bcast_rows_walk<DD>(Rc, xd, xu, T);
    }
    if (live) {
      const double e = inside ? -0.5 * (P.cterm[static_cast<int64_t>(w) * S + j] + res) : neg_inf();
      double *row = n_rows ? E + ((u * W + w) * static_cast<int64_t>(n_rows) + t) * S : E + (beg + t) * S;
      row[j] = e;
      if (j == 1) {  // the non-emitting columns of this row
        row[0] = neg_inf();
        row[S - 1] = neg_inf();
      }
    }
  }
}

// ---- the four recurrences as device functions ----------------------------------------------------
// forward (custom_hmm.py:176-211): returns the global scale = max(alpha), alpha is stored shifted
__device__ double forward_rows(const Lat E, const double *__restrict__ lA, int T, int S, const Lat al) {
  for (int s = 0; s < S; ++s) al.at(0, s) = neg_inf();
  al.at(0, 0) = 0.0;
  al.at(0, 1) = lA[0 * S + 1] + E.at(0, 1);
  for (int t = 1; t < T; ++t) {
    al.at(t, 0) = neg_inf();
    for (int j = 1; j < S - 1; ++j)
      al.at(t, j) = np_logaddexp(al.at(t - 1, j - 1) + lA[(j - 1) * S + j], al.at(t - 1, j) + lA[j * S + j]) + E.at(t, j);
    al.at(t, S - 1) = al.at(t - 1, S - 2) + lA[(S - 2) * S + S - 1];
  }
  double scale = neg_inf();
  for (int t = 0; t < T; ++t)
    for (int s = 0; s < S; ++s) {
      const double v = al.at(t, s);
      if (v > scale || v != v) scale = v;  // np.max propagates NaN
    }
  for (int t = 0; t < T; ++t)
    for (int s = 0; s < S; ++s) al.at(t, s) -= scale;
  return scale;
}

// backward (custom_hmm.py:213-246)
__device__ void backward_rows(const Lat E, const double *__restrict__ lA, int T, int S, double scale, const Lat be) {
  for (int t = 0; t < T; ++t)
    for (int s = 0; s < S; ++s) be.at(t, s) = neg_inf();
  be.at(T - 1, S - 1) = 0.0;
  for (int t = T - 2; t >= 0; --t) {
    be.at(t, 0) = lA[0 * S + 1] + E.at(t + 1, 1) + be.at(t + 1, 1);
    for (int i = 1; i < S - 2; ++i)
      be.at(t, i) = np_logaddexp(lA[i * S + i] + E.at(t + 1, i) + be.at(t + 1, i),
                                 lA[i * S + i + 1] + E.at(t + 1, i + 1) + be.at(t + 1, i + 1));
    {
      const int i = S - 2;
      be.at(t, i) = np_logaddexp(lA[i * S + i] + E.at(t + 1, i) + be.at(t + 1, i), lA[i * S + i + 1] + be.at(t + 1, i + 1));
    }
  }
  for (int t = 0; t < T - 1; ++t)
    for (int s = 0; s < S; ++s) be.at(t, s) -= scale;
}

// gamma (custom_hmm.py:248-257): row soft-max of alpha + beta via logaddexp.reduce
__device__ void gamma_rows(const Lat al, const Lat be, int T, int S, const Lat ga) {
  for (int t = 0; t < T; ++t) {
    double norm = al.at(t, 0) + be.at(t, 0);
    for (int s = 1; s < S; ++s) norm = np_logaddexp(norm, al.at(t, s) + be.at(t, s));
    for (int s = 0; s < S; ++s) ga.at(t, s) = exp((al.at(t, s) + be.at(t, s)) - norm);
  }
}

__device__ double seq_loglik(const Lat al, int T, int S) {
  double ll = al.at(T - 1, 0);
  for (int s = 1; s < S; ++s) ll = np_logaddexp(ll, al.at(T - 1, s));
  return ll;
}

// xi (custom_hmm.py:259-322), renormalised per frame; exit column uses emission = -inf.
// xi_dense (optional) receives rows t < T-1 of [S][S]; agg (optional) accumulates sum_t xi[t].
//
// Only 2S-1 entries of the (S,S) matrix can be non-zero: (0,1), (i,i) and (i,i+1) for the emitting
// states, (S-1,S-1).  np.sum over the flattened matrix is numpy's pair-wise sum; for S*S <= 128 that is
// eight strided accumulators (element k goes to accumulator k % 8, in increasing k), a fixed tree over
// them, then the S*S % 8 trailing elements one by one.  Adding an exact +0.0 never changes an
// accumulator of non-negative terms, so walking the non-zero entries in flattened order through the
// same accumulators gives the same bits as the dense sum without the S*S-element scratch array.
struct XiEntry {
  int pos;
  double val;
};

__device__ void xi_rows_dense(const Lat al, const Lat be, const Lat E, const double *__restrict__ A,
                              const double *__restrict__ lA, int T, int S, double *__restrict__ xi_dense,
                              double *__restrict__ agg) {
  const double ll = seq_loglik(al, T, S);
  double xr[kMaxS * kMaxS];
  double a[kMaxS], e[kMaxS], b[kMaxS];
  for (int t = 0; t < T - 1; ++t) {
    for (int i = 0; i < S; ++i) {
      a[i] = al.at(t, i);
      e[i] = E.at(t + 1, i);
      b[i] = be.at(t + 1, i);
    }
    for (int k = 0; k < S * S; ++k) xr[k] = 0.0;
    xr[0 * S + 1] = exp(a[0] + lA[0 * S + 1] + e[1] + b[1] - ll);
    for (int i = 1; i < S - 1; ++i) {
      if (A[i * S + i] > 0) xr[i * S + i] = exp(a[i] + lA[i * S + i] + e[i] + b[i] - ll);
      if (i < S - 2) xr[i * S + i + 1] = exp(a[i] + lA[i * S + i + 1] + e[i + 1] + b[i + 1] - ll);
    }
    xr[(S - 2) * S + S - 1] = exp(a[S - 2] + lA[(S - 2) * S + S - 1] + e[S - 1] + b[S - 1] - ll);
    xr[(S - 1) * S + S - 1] = exp(a[S - 1] + lA[(S - 1) * S + S - 1] + e[S - 1] + b[S - 1] - ll);
    // np.sum over the (S,S) matrix: pair-wise over the flattened contiguous array
    const double tot = np_pairwise_rt(xr, S * S);
    if (tot > 0)
      for (int k = 0; k < S * S; ++k) xr[k] /= tot;
    if (agg)
      for (int k = 0; k < S * S; ++k) agg[k] += xr[k];
    if (xi_dense) {
      double *xd = xi_dense + static_cast<int64_t>(t) * S * S;
      for (int k = 0; k < S * S; ++k) xd[k] = xr[k];
    }
  }
}

__device__ void xi_rows(const Lat al, const Lat be, const Lat E, const double *__restrict__ A,
                        const double *__restrict__ lA, int T, int S, double *__restrict__ xi_dense,
                        double *__restrict__ agg) {
  const int n = S * S;
  if (n > 128) return xi_rows_dense(al, be, E, A, lA, T, S, xi_dense, agg);
  const double ll = seq_loglik(al, T, S);
  const int n8 = n - n % 8;
  XiEntry nz[2 * kMaxS];
  double acc[2 * kMaxS];  // running sums of the non-zero entries (same order of additions as agg[k] += ...)
  for (int i = 0; i < 2 * S; ++i) acc[i] = 0.0;
  int cnt = 0;
  double a[kMaxS], e[kMaxS], b[kMaxS];
  for (int t = 0; t < T - 1; ++t) {
    for (int i = 0; i < S; ++i) {
      a[i] = al.at(t, i);
      e[i] = E.at(t + 1, i);
      b[i] = be.at(t + 1, i);
    }
    cnt = 0;
    nz[cnt++] = {0 * S + 1, exp(a[0] + lA[0 * S + 1] + e[1] + b[1] - ll)};
    for (int i = 1; i < S - 1; ++i) {
      // slots keep a fixed meaning across frames (2i-1: self loop, 2i: step to i+1) so that acc[] lines up
      nz[cnt++] = {i * S + i, A[i * S + i] > 0 ? exp(a[i] + lA[i * S + i] + e[i] + b[i] - ll) : 0.0};
      nz[cnt++] = {i * S + i + 1, exp(a[i] + lA[i * S + i + 1] + e[i + 1] + b[i + 1] - ll)};
    }
    nz[cnt++] = {(S - 1) * S + S - 1, exp(a[S - 1] + lA[(S - 1) * S + S - 1] + e[S - 1] + b[S - 1] - ll)};
    double tot;
    if (n < 8) {
      tot = 0.0;
      for (int i = 0; i < cnt; ++i) tot += nz[i].val;
    } else {
      double r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < cnt; ++i) {
        const int pos = nz[i].pos;
        if (pos < n8) {
#pragma unroll
          for (int j = 0; j < 8; ++j) r[j] = (pos % 8 == j) ? r[j] + nz[i].val : r[j];
        }
      }
      tot = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
      for (int i = 0; i < cnt; ++i)
        if (nz[i].pos >= n8) tot += nz[i].val;
    }
    if (tot > 0)
      for (int i = 0; i < cnt; ++i) nz[i].val /= tot;
    for (int i = 0; i < cnt; ++i) acc[i] += nz[i].val;
    if (xi_dense) {
      double *xd = xi_dense + static_cast<int64_t>(t) * n;
      for (int k = 0; k < n; ++k) xd[k] = 0.0;
      for (int i = 0; i < cnt; ++i) xd[nz[i].pos] = nz[i].val;
    }
  }
  if (agg && T > 1)
    for (int i = 0; i < cnt; ++i) agg[nz[i].pos] += acc[i];
}

// one lane = one utterance against model utt_model[u].  lane_slots == 0: lattices E/alpha/beta/gamma are
// [total_frames][S] rows at the utterance's frame offset (the reference's layout); lane_slots > 0: they
// are [max_T][S][lane_slots] with the utterance index as the fastest axis (coalesced; what baum_welch
// uses).  xi_dense (optional) is [total_frames][S][S] (rows t < T-1 used).
// utt_out[u] = {LL (scaled alpha, logaddexp.reduce(alpha[-1])), scale, agg_gamma[S], agg_xi[S][S]}
__global__ __launch_bounds__(kBlock) void custom_estep_kernel(
    const float *__restrict__ feats, const int64_t *__restrict__ offsets,
    const int32_t *__restrict__ utt_model, int64_t n_utts, int D, int S, CustomPack P, int64_t lane_slots,
    double *__restrict__ Eo, double *__restrict__ alpha, double *__restrict__ beta,
    double *__restrict__ gamma, double *__restrict__ xi_dense, double *__restrict__ utt_out) {
  const int64_t u = blockIdx.x * static_cast<int64_t>(kBlock) + threadIdx.x;
  if (u >= n_utts) return;
  const int w = utt_model ? utt_model[u] : 0;
  const int64_t beg = offsets[u];
  const int T = static_cast<int>(offsets[u + 1] - beg);
  const int K = 2 + S + S * S;
  double *out = utt_out + u * K;
  for (int k = 0; k < K; ++k) out[k] = 0.0;
  if (T <= 0) return;
  const int64_t base = lane_slots ? u : beg * S, es = lane_slots ? lane_slots : 1;
  const Lat E{Eo + base, es, S}, al{alpha + base, es, S}, be{beta + base, es, S}, ga{gamma + base, es, S};
  const double *lA = P.logA + static_cast<int64_t>(w) * S * S;
  const double *A = P.A + static_cast<int64_t>(w) * S * S;

  emission_rows(feats + beg * D, T, D, S, P, w, E);
  const double scale = forward_rows(E, lA, T, S, al);
  backward_rows(E, lA, T, S, scale, be);
  gamma_rows(al, be, T, S, ga);
  {  // aggregated_gamma += sum(gamma[:-1])   (custom_hmm.py:434)
    double gs[kMaxS];
    for (int s = 0; s < S; ++s) gs[s] = 0.0;
    for (int t = 0; t < T - 1; ++t)
      for (int s = 0; s < S; ++s) gs[s] += ga.at(t, s);
    for (int s = 0; s < S; ++s) out[2 + s] = gs[s];
  }
  out[0] = seq_loglik(al, T, S);  // of the SCALED alpha (custom_hmm.py:438)
  out[1] = scale;
  xi_rows(al, be, E, A, lA, T, S, xi_dense ? xi_dense + beg * S * S : nullptr, out + 2 + S);
}

// ---- batched E-step, compile-time shapes --------------------------------------------------------------
// Same arithmetic as custom_estep_kernel (operation for operation: the golden-vector tests cover both), laid
// out for registers: S and D are template parameters, so the per-row state (alpha / beta rows, the 2S-2
// structurally non-zero xi entries and their running sums, the row-sum vector v) lives in VGPRs instead of
// the 4.9 KB of scratch memory per lane the run-time-shaped kernel needs.  Lattices are the lane-contiguous
// [max_T][S][lane_slots] layout only.  alpha and beta are stored UNSHIFTED; the reference's `alpha -= scale`
// / `beta[:-1] -= scale` (custom_hmm.py:208-209,244) happen on the fly where the values are read, which is
// the same single rounding and saves a read-modify-write pass over each lattice; gamma is produced inside
// the backward loop (it needs only row t of both lattices).
template <int S>
constexpr int xi_pos(int i) {  // flattened (S,S) position of the i-th structurally non-zero xi entry
  return i == 0 ? 1 : (i == 2 * S - 3 ? (S - 1) * S + S - 1 : ((i + 1) / 2) * S + (i + 1) / 2 + (i % 2 == 0 ? 1 : 0));
}
// np.sum over the flattened (S,S) matrix = numpy's pair-wise sum of S*S values of which only the entries at
// xi_pos are non-zero: walk numpy's recursion at compile time and feed each entry to the accumulator its
// position selects (adding the exact zeros in between changes nothing)
template <int S, int START, int N>
__device__ __forceinline__ double xi_pairwise(const double (&val)[2 * S - 2]) {
  if constexpr (N > 128) {
    constexpr int n2 = N / 2 - (N / 2) % 8;
    const double left = xi_pairwise<S, START, n2>(val);
    return left + xi_pairwise<S, START + n2, N - n2>(val);
  } else if constexpr (N < 8) {
    double r = 0.0;
    static_for_c<0, 2 * S - 2>([&](auto ic) {
      constexpr int i = decltype(ic)::value, p = xi_pos<S>(i) - START;
      if constexpr (p >= 0 && p < N) r += val[i];
    });
    return r;
  } else {
    constexpr int n8 = N - N % 8;
    double r[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    static_for_c<0, 2 * S - 2>([&](auto ic) {
      constexpr int i = decltype(ic)::value, p = xi_pos<S>(i) - START;
      if constexpr (p >= 0 && p < n8) r[p % 8] += val[i];
    });
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    static_for_c<0, 2 * S - 2>([&](auto ic) {
      constexpr int i = decltype(ic)::value, p = xi_pos<S>(i) - START;
      if constexpr (p >= n8 && p < N) res += val[i];
    });
    return res;
  }
}

// ---- emission of the batched E-step: E[t][j] = -0.5 (c_j + d_t . v_j),  v_j = C_j^-1 (sum_s x_s - T mu_j)
// (custom_hmm.py:146-174).  Inside the lane-per-utterance E-step kernel this was one pass over the utterance's frames per
// emitting state: 9 x 0.5 GB of features streamed from HBM per 100 000 utterances (a wavefront's 336 KB do not stay in
// cache between passes) — half of that kernel's time.  Here a WORKGROUP holds 64 utterances and a wavefront one of eight
// states: the eight wavefronts walk the same frames at the same time on one CU, so a feature row crosses HBM once and
// reaches the other seven from L1 / L2.  Every lane performs the operations it performed inside the E-step kernel, in
// the same order (the frame sum is evaluated by each wavefront, as before by each lane).
// STAGED: features from the slot-major copy feat_t[t][d][slot] (sapr_custom_stage_features) — a wavefront's load of one
// value is one 256-byte row instead of 64 private 4-byte reads.
constexpr int kEmitStates = 8;  // wavefronts (states) per workgroup; S - 2 is a multiple at the batched shapes
template <int S, int D, bool STAGED>
__global__ __launch_bounds__(kBlock * kEmitStates) __attribute__((amdgpu_waves_per_eu(D <= 13 ? 4 : 2))) void custom_emit_kernel(const float *__restrict__ feats,
                                                             const int64_t *__restrict__ offsets,
                                                             const int32_t *__restrict__ utt_model, int64_t n_utts,
                                                             CustomPack P, int64_t es, double *__restrict__ Eo,
                                                             const float *__restrict__ feat_t,
                                                             const double *__restrict__ frame_sums) {
  static_assert((S - 2) % kEmitStates == 0, "whole workgroups of states");
  const int64_t u = blockIdx.x * static_cast<int64_t>(kBlock) + (threadIdx.x & (kBlock - 1));
  if (u >= n_utts) return;
  const int j = static_cast<int>(blockIdx.y) * kEmitStates + static_cast<int>(threadIdx.x / kBlock) + 1;
  const int w = utt_model ? utt_model[u] : 0;
  const int64_t beg = offsets[u];
  const int T = static_cast<int>(offsets[u + 1] - beg);
  if (T <= 0) return;
  const float *__restrict__ x = feats + beg * D;
  const float *__restrict__ xt = STAGED ? feat_t + u : nullptr;
  auto feat = [&](int t, int d) -> double {
    if constexpr (STAGED)
      return static_cast<double>(xt[(static_cast<int64_t>(t) * D + d) * es]);
    else
      return static_cast<double>(x[static_cast<int64_t>(t) * D + d]);
  };
  double *__restrict__ E = Eo + u;
  auto at = [es](int t, int jj) { return (static_cast<int64_t>(t) * S + jj) * es; };
  // frame sums: the dimensions are dealt to the workgroup's wavefronts (each sum still runs over the frames in order, in
  // one lane) and exchanged through LDS — two feature rows per frame and wavefront instead of all D; four frames of
  // loads in flight per step
  constexpr int kNd = (D + kEmitStates - 1) / kEmitStates;
  __shared__ double s_xs[D][kBlock];
  const int wv = static_cast<int>(threadIdx.x / kBlock), ln = static_cast<int>(threadIdx.x & (kBlock - 1));
  // (round 4) the frame sums do not change between EM iterations: sapr_custom_stage_features evaluates them once per
  // batch — the same additions in the same order, custom_frame_sums_kernel — and every iteration's pass over the
  // features for them (0.5 GB per 100 000 utterances, a third of this kernel's HBM traffic) falls away
  if (frame_sums == nullptr) {
    double part[kNd];
#pragma unroll
    for (int k = 0; k < kNd; ++k) part[k] = 0.0;
    int t = 0;
    for (; t + 4 <= T; t += 4) {
      double f[4][kNd];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < kNd; ++k) f[q][k] = (wv + k * kEmitStates < D) ? feat(t + q, wv + k * kEmitStates) : 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < kNd; ++k) part[k] += f[q][k];
    }
    for (; t < T; ++t)
#pragma unroll
      for (int k = 0; k < kNd; ++k)
        if (wv + k * kEmitStates < D) part[k] += feat(t, wv + k * kEmitStates);
#pragma unroll
    for (int k = 0; k < kNd; ++k)
      if (wv + k * kEmitStates < D) s_xs[wv + k * kEmitStates][ln] = part[k];
  }
  __syncthreads();  // (frame_sums is a kernel argument: the branch above is uniform)
  double xs[D];
#pragma unroll
  for (int d = 0; d < D; ++d) xs[d] = frame_sums ? frame_sums[static_cast<int64_t>(d) * es + u] : s_xs[d][ln];
  if (j == 1) {  // entry and exit state never emit
    for (int t = 0; t < T; ++t) {
      E[at(t, 0)] = neg_inf();
      E[at(t, S - 1)] = neg_inf();
    }
  }
  const double *mu = P.means + (static_cast<int64_t>(w) * S + j) * D;
  const double *iv = P.inv + (static_cast<int64_t>(w) * S + j) * D * D;
  double v[D], m[D];
#pragma unroll
  for (int a = 0; a < D; ++a) m[a] = mu[a];
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double acc = 0.0;
#pragma unroll
    for (int b = 0; b < D; ++b) acc += iv[a * D + b] * (xs[b] - T * m[b]);
    v[a] = acc;
    // one row of the inverse covariance in flight at a time: left free, the compiler issues all D x D loads at once and
    // spills ~600 bytes per lane around them (0.5 GB of scratch traffic per 100 000 utterances x 8 states)
    asm volatile("" ::: "memory");
  }
  const double c = P.cterm[static_cast<int64_t>(w) * S + j];
  constexpr int kNf = D <= 13 ? 4 : 2;  // frames of loads in flight
  int t = 0;
  for (; t + kNf <= T; t += kNf) {
    float xf[kNf][D];
#pragma unroll
    for (int q = 0; q < kNf; ++q)
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if constexpr (STAGED)
          xf[q][d] = xt[(static_cast<int64_t>(t + q) * D + d) * es];
        else
          xf[q][d] = x[static_cast<int64_t>(t + q) * D + d];
      }
#pragma unroll
    for (int q = 0; q < kNf; ++q) {
      double qd = 0.0;
#pragma unroll
      for (int d = 0; d < D; ++d) qd += (static_cast<double>(xf[q][d]) - m[d]) * v[d];
      E[at(t + q, j)] = -0.5 * (c + qd);
    }
  }
  for (; t < T; ++t) {
    double qd = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) qd += (feat(t, d) - m[d]) * v[d];
    E[at(t, j)] = -0.5 * (c + qd);
  }
}

// PASS 0: every utterance — forward pass over the stored log-densities, then the backward half as a smoothing recursion wherever that
// returns what the reference returns (see the classification below); the utterances where it would not are appended
// to `redo`.  PASS 1: the utterances of `redo` again, start to end in the reference's own operation order.
// (10 states x 13 dimensions: capped at 256 registers so that two workgroups share a CU — 100 000 utterances are 1.5
// rounds of one wavefront per SIMD otherwise, and the second wavefront covers the first's exp / log1p latencies;
// the larger shapes keep every register they can get in pass 1)
template <int S, int D, int PASS>
__global__ __launch_bounds__(kBlock, (S <= 10 && D <= 13) ? 2 : 1) void custom_estep_fast_kernel(
    const float *__restrict__ feats, const int64_t *__restrict__ offsets, const int32_t *__restrict__ utt_model,
    int64_t n_utts, CustomPack P, int64_t es, double *__restrict__ Eo, double *__restrict__ alpha,
    double *__restrict__ beta, double *__restrict__ gamma, double *__restrict__ utt_out, int32_t *__restrict__ redo,
    int32_t *__restrict__ redo_count, const float *__restrict__ feat_t) {
  constexpr bool kFirst = PASS != 1;
  int64_t u = blockIdx.x * static_cast<int64_t>(kBlock) + threadIdx.x;
  if constexpr (kFirst) {
    if (u >= n_utts) return;
  } else {
    if (u >= *redo_count) return;  // (the grid covers n_utts: almost every workgroup leaves here)
    u = redo[u];
  }
  const int w = utt_model ? utt_model[u] : 0;
  const int64_t beg = offsets[u];
  const int T = static_cast<int>(offsets[u + 1] - beg);
  constexpr int K = 2 + S + S * S;
  double *out = utt_out + u * K;
  if constexpr (kFirst)
    for (int k = 0; k < K; ++k) out[k] = 0.0;
  if (T <= 0) return;
  double *__restrict__ E = Eo + u, *__restrict__ al = alpha + u, *__restrict__ be = beta + u,
                      *__restrict__ ga = gamma + u;
  auto at = [es](int t, int j) { return (static_cast<int64_t>(t) * S + j) * es; };
  const double *__restrict__ lA = P.logA + static_cast<int64_t>(w) * S * S;
  const double *__restrict__ A = P.A + static_cast<int64_t>(w) * S * S;

  // (the log-densities E are in their lattice: custom_emit_kernel ran before pass 0)

  // ---- forward (custom_hmm.py:176-211); scale = np.max(alpha) (NaN propagates).  Pass 0 keeps the rows in registers
  // only (its smoothing recursion needs the last one): the alpha lattice is written by pass 1, which runs the forward
  // pass again for the utterances it redoes (the same operations: the same values) and reads the rows back, unshifted,
  // in its backward half — 0.8 GB of lattice writes per 100 000 utterances less in pass 0
  double scale = neg_inf();
  double prev[S];
  {
    double cur[S];
#pragma unroll
    for (int s2 = 0; s2 < S; ++s2) prev[s2] = neg_inf();
    prev[0] = 0.0;
    prev[1] = lA[0 * S + 1] + E[at(0, 1)];
#pragma unroll
    for (int s2 = 0; s2 < S; ++s2) {
      if constexpr (!kFirst) al[at(0, s2)] = prev[s2];
      if (prev[s2] > scale || prev[s2] != prev[s2]) scale = prev[s2];
    }
    // (round 4) the log-densities of frame t + 1 are requested at the top of frame t: left to the compiler the loads
    // sat behind the frame's exp / log chains, a few instructions in front of their use — 46 % of the kernel's
    // wave-cycles were s_waitcnt vmcnt at the 1.5 wavefronts per SIMD a 100 000-utterance grid gives
    double e_cur[S], e_nxt[S];
    if (T > 1) {
#pragma unroll
      for (int j = 1; j < S - 1; ++j) e_nxt[j] = E[at(1, j)];
    }
    for (int t = 1; t < T; ++t) {
#pragma unroll
      for (int j = 1; j < S - 1; ++j) e_cur[j] = e_nxt[j];
      if (t + 1 < T) {
#pragma unroll
        for (int j = 1; j < S - 1; ++j) e_nxt[j] = E[at(t + 1, j)];
      }
      cur[0] = neg_inf();
#pragma unroll
      for (int j = 1; j < S - 1; ++j) {
        // the share of state j's forward mass that STAYED in j (the rest came from j - 1): the smoothing form of the
        // backward pass (below) runs on it; it waits in the beta lattice, which holds nothing else for an utterance
        // that takes that form
        if constexpr (kFirst) {
          double move, stay;
          cur[j] = np_logaddexp_shares(prev[j - 1] + lA[(j - 1) * S + j], prev[j] + lA[j * S + j], move, stay) + e_cur[j];
          be[at(t, j)] = stay;
        } else {
          cur[j] = np_logaddexp(prev[j - 1] + lA[(j - 1) * S + j], prev[j] + lA[j * S + j]) + e_cur[j];
        }
      }
      cur[S - 1] = prev[S - 2] + lA[(S - 2) * S + S - 1];
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) {
        if constexpr (!kFirst) al[at(t, s2)] = cur[s2];
        if (cur[s2] > scale || cur[s2] != cur[s2]) scale = cur[s2];
        prev[s2] = cur[s2];
      }
    }
    // LL of the SCALED alpha: logaddexp.reduce(alpha[-1] - scale)   (custom_hmm.py:268,438)
    double l = prev[0] - scale;
#pragma unroll
    for (int s2 = 1; s2 < S; ++s2) l = np_logaddexp(l, prev[s2] - scale);
    out[0] = l;
    out[1] = scale;
  }
  const double ll = out[0];

  // ---- which form of the backward pass this utterance takes (round 3b).  custom_hmm.py:213-322 computes beta in the
  // log domain, gamma = softmax(alpha + beta) and the xi terms exp(alpha - s + log a + e + beta - s - LL), renormalised
  // per frame: 46 of the 62 float64 transcendentals of a frame.  In exact arithmetic all of it follows from the
  // forward pass: the shares move / stay of state j's forward mass at frame t + 1 are P(q_t | q_(t+1) = j, O), so
  //     xi_t(j-1 -> j) = gamma_(t+1)(j) move,   xi_t(j -> j) = gamma_(t+1)(j) stay,   gamma_t(i) = xi_t(i -> i) + xi_t(i -> i+1)
  // — multiplications and additions.  What the reference returns differs where its exponentials underflow: every xi
  // term carries the factor rho exp(-s) (s = max alpha of the utterance, rho = the exit state's share of the last
  // forward row) until the per-frame renormalisation divides it out, so with c0 = log rho - s
  //     c0 >= -678   no term above 1e-13 leaves the normal range (exp(-708)): the smoothing recursion gives the
  //                  reference's values to rounding
  //     c0 <  -750   EVERY term underflows to exactly 0 (exp(x) = 0 below -745.2), the frame totals are 0, nothing is
  //                  renormalised: the utterance contributes posteriors but no transition counts (trained models
  //                  with densities >> 1, e.g. digital silence)
  //     otherwise    (or NaN) some terms are denormal or gone: the reference's own operation order (pass 1)
  // The exit state's xi term is exp(-inf) = 0 in every mode, so the last step (all mass in the exit state) adds nothing.
  if constexpr (kFirst) {
    const double c0 = (prev[S - 1] - (ll + scale)) - scale;
    const int mode = c0 >= -678.0 ? 0 : (c0 < -750.0 ? 1 : 2);
    if (mode == 2) {  // pass 1 takes this utterance again
      redo[atomicAdd(redo_count, 1)] = static_cast<int32_t>(u);
      return;
    }
    double gs[S];
    constexpr int CNT = 2 * S - 2;
    double acc[CNT];
#pragma unroll
    for (int s2 = 0; s2 < S; ++s2) gs[s2] = 0.0;
#pragma unroll
    for (int i = 0; i < CNT; ++i) acc[i] = 0.0;
    double g1[S];  // gamma of row t + 1
    {
      double lg[S];
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) lg[s2] = (prev[s2] - scale) + (s2 == S - 1 ? 0.0 : neg_inf());
      double mx = lg[0];
#pragma unroll
      for (int s2 = 1; s2 < S; ++s2) mx = lg[s2] > mx ? lg[s2] : mx;
      double den = 0.0;
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) {
        lg[s2] = exp_unit(lg[s2] - mx);
        den += lg[s2];
      }
      const double inv = 1.0 / den;
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) g1[s2] = lg[s2] * inv;
    }
    // shares of the step into row t + 1, read one row ahead
    double sy[S];
    if (T > 1) {
#pragma unroll
      for (int j = 1; j < S - 1; ++j) sy[j] = be[at(T - 1, j)];
    }
#pragma unroll
    for (int s2 = 0; s2 < S; ++s2) ga[at(T - 1, s2)] = g1[s2];
    for (int t = T - 2; t >= 0; --t) {
      double nsy[S];
      if (t >= 1) {
#pragma unroll
        for (int j = 1; j < S - 1; ++j) nsy[j] = be[at(t, j)];
      }
      double g0[S], x_move[S], x_stay[S];  // xi_t(j-1 -> j), xi_t(j -> j) for the emitting j
#pragma unroll
      for (int j = 1; j < S - 1; ++j) {
        x_stay[j] = g1[j] * sy[j];
        x_move[j] = g1[j] - x_stay[j];  // the two shares add up to 1 (to rounding): one lattice instead of two
      }
      g0[0] = x_move[1];
#pragma unroll
      for (int i = 1; i < S - 2; ++i) g0[i] = x_stay[i] + x_move[i + 1];
      g0[S - 2] = x_stay[S - 2] + g1[S - 1];  // the exit state is entered from S - 2 only
      g0[S - 1] = 0.0;
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) {
        ga[at(t, s2)] = g0[s2];
        gs[s2] += g0[s2];
      }
      if (mode == 0 && t + 1 < T - 1) {
        acc[0] += x_move[1];
#pragma unroll
        for (int i = 1; i < S - 1; ++i) {
          acc[2 * i - 1] += x_stay[i];
          if (i + 1 < S - 1) acc[2 * i] += x_move[i + 1];
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) g1[s2] = g0[s2];
#pragma unroll
      for (int j = 1; j < S - 1; ++j) sy[j] = nsy[j];
    }
#pragma unroll
    for (int s2 = 0; s2 < S; ++s2) out[2 + s2] = gs[s2];
    if (T > 1) {
      double *agg = out + 2 + S;
      static_for_c<0, CNT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        agg[xi_pos<S>(i)] += acc[i];
      });
    }
    return;
  } else {
  // ---- backward (custom_hmm.py:213-246) with gamma (:248-257) of each finished row, and — while row t+1 of beta,
  // the emissions of frame t+1 and row t of alpha are in registers — the xi terms of step t (:259-322) and the
  // aggregated gamma over t < T-1 (:434).  The reference adds those over ascending t; here they are added in the
  // order the backward recursion meets them (descending t): the same sums to rounding, compared at 1e-9, and
  // three lattice re-reads per frame less than a separate pass.
  {
    double nxt[S], cur[S];  // unshifted beta rows t+1 and t
    double gs[S];
    constexpr int CNT = 2 * S - 2;
    double acc[CNT];
#pragma unroll
    for (int s2 = 0; s2 < S; ++s2) {
      nxt[s2] = neg_inf();
      gs[s2] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < CNT; ++i) acc[i] = 0.0;
    nxt[S - 1] = 0.0;
    // gamma of row t; a[] receives alpha[t] - scale, g[] the posteriors
    auto gamma_row = [&](int t, const double (&b)[S], bool shifted_row, double (&a)[S], double (&g)[S]) {
      double lg[S];
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) {
        a[s2] = al[at(t, s2)] - scale;
        lg[s2] = a[s2] + (shifted_row ? b[s2] - scale : b[s2]);
      }
      // exp(lg - logaddexp.reduce(lg)) as exp(lg - max) / sum: S exponentials and one division instead of S - 1
      // logaddexp (exp + log1p each) and S more exponentials; same value to rounding (gamma is compared at 1e-9)
      double mx = lg[0];
#pragma unroll
      for (int s2 = 1; s2 < S; ++s2) mx = lg[s2] > mx ? lg[s2] : mx;
      double den = 0.0;
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) {
        lg[s2] = exp_unit(lg[s2] - mx);
        den += lg[s2];
      }
      const double inv = 1.0 / den;
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) {
        g[s2] = lg[s2] * inv;
        ga[at(t, s2)] = g[s2];
      }
    };
    double arow[S], grow[S];
#pragma unroll
    for (int s2 = 0; s2 < S; ++s2) be[at(T - 1, s2)] = nxt[s2];
    gamma_row(T - 1, nxt, false, arow, grow);  // the last beta row is not shifted (be[:-1] -= scale)
    for (int t = T - 2; t >= 0; --t) {
      double e1[S];
      e1[0] = e1[S - 1] = neg_inf();
#pragma unroll
      for (int s2 = 1; s2 < S - 1; ++s2) e1[s2] = E[at(t + 1, s2)];
      cur[0] = lA[0 * S + 1] + e1[1] + nxt[1];
#pragma unroll
      for (int i = 1; i < S - 2; ++i)
        cur[i] = np_logaddexp(lA[i * S + i] + e1[i] + nxt[i], lA[i * S + i + 1] + e1[i + 1] + nxt[i + 1]);
      cur[S - 2] = np_logaddexp(lA[(S - 2) * S + S - 2] + e1[S - 2] + nxt[S - 2],
                                lA[(S - 2) * S + S - 1] + nxt[S - 1]);
      cur[S - 1] = neg_inf();
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) be[at(t, s2)] = cur[s2];
      gamma_row(t, cur, true, arow, grow);
      // xi of step t: alpha[t] - scale, emissions and (shifted) beta of frame t+1
      {
        const bool shifted = (t + 1) < (T - 1);
        double b[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
          gs[i] += grow[i];
          b[i] = shifted ? nxt[i] - scale : nxt[i];
        }
        double val[CNT];
        val[0] = exp_unit(arow[0] + lA[0 * S + 1] + e1[1] + b[1] - ll);
#pragma unroll
        for (int i = 1; i < S - 1; ++i) {
          val[2 * i - 1] = A[i * S + i] > 0 ? exp_unit(arow[i] + lA[i * S + i] + e1[i] + b[i] - ll) : 0.0;
          val[2 * i] = exp_unit(arow[i] + lA[i * S + i + 1] + e1[i + 1] + b[i + 1] - ll);
        }
        val[CNT - 1] = exp_unit(arow[S - 1] + lA[(S - 1) * S + S - 1] + e1[S - 1] + b[S - 1] - ll);
        const double tot = xi_pairwise<S, 0, S * S>(val);
        if (tot > 0) {
#pragma unroll
          for (int i = 0; i < CNT; ++i) val[i] /= tot;
        }
#pragma unroll
        for (int i = 0; i < CNT; ++i) acc[i] += val[i];
      }
#pragma unroll
      for (int s2 = 0; s2 < S; ++s2) nxt[s2] = cur[s2];
    }
#pragma unroll
    for (int s2 = 0; s2 < S; ++s2) out[2 + s2] = gs[s2];
    if (T > 1) {
      double *agg = out + 2 + S;
      static_for_c<0, CNT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        agg[xi_pos<S>(i)] += acc[i];
      });
    }
  }
  }  // PASS == 1
}

// single-utterance pieces with caller-supplied inputs (the reference's per-method API, used by its
// tests): op 0 emission(features) 1 forward(E) 2 backward(E, scale) 3 gamma(alpha, beta) 4 xi(alpha, beta, E)
__global__ void custom_piece_kernel(int op, const float *__restrict__ x, int T, int D, int S, CustomPack P,
                                    double *__restrict__ E, double *__restrict__ al, double *__restrict__ be,
                                    double *__restrict__ ga, double *__restrict__ xi, double *__restrict__ scalar) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const Lat LE{E, 1, S}, La{al, 1, S}, Lb{be, 1, S}, Lg{ga, 1, S};
  if (op == 0) emission_rows(x, T, D, S, P, 0, LE);
  if (op == 1) scalar[0] = forward_rows(LE, P.logA, T, S, La);
  if (op == 2) backward_rows(LE, P.logA, T, S, scalar[0], Lb);
  if (op == 3) gamma_rows(La, Lb, T, S, Lg);
  if (op == 4) xi_rows(La, Lb, LE, P.A, P.logA, T, S, xi, nullptr);
}

// Viterbi of custom_hmm.py:462-514 for every (utterance, model) over the first Tq frames (Tq =
// features.shape[0] = D, the reference's quirk), strict '>' from -inf.  Erows[(u*W+w)*Tq + t][S] comes from
// custom_emission_exact_kernel, so V holds the reference's bits: np.log(A) is evaluated on the host and the
// trellis only adds and compares.  scores[u][w]; paths[u][w][Tq].
__global__ __launch_bounds__(kBlock) void custom_decode_kernel(
    const double *__restrict__ Erows, int64_t n_utts, int W, int S, int num_states, int Tq, CustomPack P,
    double *__restrict__ scores, int32_t *__restrict__ paths) {
  const int64_t idx = blockIdx.x * static_cast<int64_t>(kBlock) + threadIdx.x;
  if (idx >= n_utts * W) return;
  const int w = static_cast<int>(idx % W);
  const double *E = Erows + idx * static_cast<int64_t>(Tq) * S;
  const double *lA = P.logA + static_cast<int64_t>(w) * S * S;

  double V[kMaxS], Vn[kMaxS];
  int32_t *bp = paths + idx * static_cast<int64_t>(Tq);
  // back-pointers: Tq x S small ints in a local array (Tq <= kMaxD, S <= kMaxS)
  unsigned char back[kMaxD * kMaxS];
  for (int s = 0; s < S; ++s) V[s] = neg_inf();
  for (int i = 0; i < Tq * S; ++i) back[i] = 0;
  V[0] = 0.0;
  V[1] = lA[0 * S + 1] + E[1];
  for (int t = 1; t < Tq; ++t) {
    for (int s = 0; s < S; ++s) Vn[s] = neg_inf();
    for (int j = 1; j < S; ++j) {
      int cand[2], nc = 0;
      if (j == 1) {
        cand[nc++] = 1;
        if (t == 1) cand[nc++] = 0;
      } else if (j == S - 1) {
        if (t < num_states) continue;
        cand[nc++] = j - 1;
        cand[nc++] = j;
      } else {
        cand[nc++] = j - 1;
        cand[nc++] = j;
      }
      double best = neg_inf();
      int arg = -1;
      for (int c = 0; c < nc; ++c) {
        const double sc = V[cand[c]] + lA[cand[c] * S + j];
        if (sc > best) {
          best = sc;
          arg = cand[c];
        }
      }
      if (arg >= 0) {
        Vn[j] = (j != S - 1) ? best + E[static_cast<int64_t>(t) * S + j] : best;
        back[t * S + j] = static_cast<unsigned char>(arg);
      }
    }
    for (int s = 0; s < S; ++s) V[s] = Vn[s];
  }
  scores[idx] = V[S - 1];
  int cur = S - 1;
  for (int t = Tq - 1; t >= 0; --t) {
    bp[t] = cur;
    cur = back[t * S + cur];
  }
}

// The same trellis for the left-to-right models of this vocabulary with S as a template parameter (10, 18): scores,
// transition terms and the frame's emission row live in registers (the kernel above indexes V[] / back[] with run-time
// values: scratch, and its loads sit inside the dependent chain), the next frame's row is loaded under the current
// frame's arithmetic, the back-pointers of a frame are one 64-bit word (2 bits per state: 0 = unset -> state 0 as in
// the byte array above, 1 = from j - 1, 2 = stay).  Same comparisons in the same order.
template <int S>
__global__ __launch_bounds__(kBlock) void custom_decode_lr_kernel(
    const double *__restrict__ Erows, int64_t n_utts, int W, int num_states, int Tq, CustomPack P,
    double *__restrict__ scores, int32_t *__restrict__ paths) {
  const int64_t idx = blockIdx.x * static_cast<int64_t>(kBlock) + threadIdx.x;
  if (idx >= n_utts * W) return;
  const int w = static_cast<int>(idx % W);
  const double *E = Erows + idx * static_cast<int64_t>(Tq) * S;
  const double *lA = P.logA + static_cast<int64_t>(w) * S * S;
  double a_in[S], a_st[S], V[S];
#pragma unroll
  for (int j = 1; j < S; ++j) {
    a_in[j] = lA[(j - 1) * S + j];
    a_st[j] = lA[j * S + j];
  }
#pragma unroll
  for (int s = 0; s < S; ++s) V[s] = neg_inf();
  V[0] = 0.0;
  V[1] = a_in[1] + E[1];
  unsigned long long back[kMaxD];
  back[0] = 0;
  double e_next[S - 2];
#pragma unroll
  for (int i = 0; i < S - 2; ++i) e_next[i] = Tq > 1 ? E[S + 1 + i] : 0.0;
  for (int t = 1; t < Tq; ++t) {
    double e[S - 2], Vn[S];
#pragma unroll
    for (int i = 0; i < S - 2; ++i) e[i] = e_next[i];
    if (t + 1 < Tq) {
#pragma unroll
      for (int i = 0; i < S - 2; ++i) e_next[i] = E[static_cast<int64_t>(t + 1) * S + 1 + i];
    }
    unsigned long long bits = 0;
    Vn[0] = neg_inf();
    {  // j == 1: stay, then (first step only) the entry state
      double best = neg_inf();
      unsigned code = 0;
      const double s_stay = V[1] + a_st[1];
      if (s_stay > best) {
        best = s_stay;
        code = 2;
      }
      if (t == 1) {
        const double s_in = V[0] + a_in[1];
        if (s_in > best) {
          best = s_in;
          code = 1;
        }
      }
      Vn[1] = code ? best + e[0] : neg_inf();
      bits |= static_cast<unsigned long long>(code) << 2;
    }
#pragma unroll
    for (int j = 2; j < S - 1; ++j) {
      double best = neg_inf();
      unsigned code = 0;
      const double s_in = V[j - 1] + a_in[j], s_stay = V[j] + a_st[j];
      if (s_in > best) {
        best = s_in;
        code = 1;
      }
      if (s_stay > best) {
        best = s_stay;
        code = 2;
      }
      Vn[j] = code ? best + e[j - 1] : neg_inf();
      bits |= static_cast<unsigned long long>(code) << (2 * j);
    }
    Vn[S - 1] = neg_inf();
    if (t >= num_states) {  // the exit state: no emission term
      double best = neg_inf();
      unsigned code = 0;
      const double s_in = V[S - 2] + a_in[S - 1], s_stay = V[S - 1] + a_st[S - 1];
      if (s_in > best) {
        best = s_in;
        code = 1;
      }
      if (s_stay > best) {
        best = s_stay;
        code = 2;
      }
      if (code) Vn[S - 1] = best;
      bits |= static_cast<unsigned long long>(code) << (2 * (S - 1));
    }
    back[t] = bits;
#pragma unroll
    for (int s = 0; s < S; ++s) V[s] = Vn[s];
  }
  scores[idx] = V[S - 1];
  int32_t *bp = paths + idx * static_cast<int64_t>(Tq);
  int cur = S - 1;
  for (int t = Tq - 1; t >= 0; --t) {
    bp[t] = cur;
    const unsigned code = static_cast<unsigned>(back[t] >> (2 * cur)) & 3u;
    cur = code == 2 ? cur : code == 1 ? cur - 1 : 0;
  }
}

// decoder.py:35-49 over the custom models: first strict maximum in model order starting from -inf
// (a NaN score never wins; no finite or +inf score -> word -1, score -inf, path untouched).
__global__ void custom_best_word_kernel(const double *__restrict__ scores, const int32_t *__restrict__ paths,
                                        int64_t n_utts, int W, int Tq, int32_t *__restrict__ best_word,
                                        double *__restrict__ best_score, int32_t *__restrict__ best_path) {
  const int64_t u = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (u >= n_utts) return;
  double best = neg_inf();
  int bw = -1;
  for (int w = 0; w < W; ++w) {
    const double sc = scores[u * W + w];
    if (sc > best) {
      best = sc;
      bw = w;
    }
  }
  best_word[u] = bw;
  best_score[u] = best;
  if (bw >= 0)
    for (int t = 0; t < Tq; ++t) best_path[u * Tq + t] = paths[(u * W + bw) * static_cast<int64_t>(Tq) + t];
}

// update_B (custom_hmm.py:366-400) is two-pass: means first, then covariances about the NEW means.
// Both passes are split into a part that is parallel over utterances and an ordered combination, so
// that they scale to 10^5 utterances and can be summed across ranks between the passes:
//
//   pass 1   update_b_utt_sums_kernel   one lane per (utterance, state, dim): the reference's
//                                       np.sum(gamma[:, j:j+1] * features.T, axis=0) of ONE sequence
//                                       (frames in order) -> part[u][j][d], occ_part[u][j]
//            update_b_fold_kernel       one lane per (model, state, dim): adds the per-sequence sums in
//                                       LIST ORDER, exactly like the reference's loop over sequences
//                                       -> unnormalised sum_x[w][j][d], occ[w][j]
//   pass 2   update_b_scatter_kernel    one lane per (chunk of >= 32 utterances, model, state, a, b):
//                                       sum gamma * (x_a - mu_a)(x_b - mu_b) over the chunk
//            update_b_fold_kernel       chunks in order -> unnormalised scatter[w][j][a][b]
//   custom_normalise_kernel             x / occ where occ > 0 (host: symmetrise, floor the diagonal)
// utterances per pass-2 chunk: 32, or more when that would exceed the 65535 (chunk, model) rows of one grid
__host__ __device__ inline int64_t chunk_utts(int64_t n_utts, int W) {
  const int64_t need = (n_utts * W + 65534) / 65535;
  return need > 32 ? need : 32;
}

// gamma (t, j) of utterance u: reference row layout [total_frames][S] (lane_slots == 0) or the batched
// E-step's lane-contiguous [max_T][S][lane_slots]
// one frame-major feature row as 16-byte pieces (rows are 4-byte aligned: D floats back to back)
struct __attribute__((packed, aligned(4))) RowQuad {
  float a, b, c, d;
};
template <int D>
__device__ __forceinline__ void load_row_f32(const float *__restrict__ p, float (&f)[D]) {
  constexpr int Q = D / 4;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const RowQuad v = *reinterpret_cast<const RowQuad *>(p + 4 * q);
    f[4 * q] = v.a, f[4 * q + 1] = v.b, f[4 * q + 2] = v.c, f[4 * q + 3] = v.d;
  }
#pragma unroll
  for (int d = 4 * Q; d < D; ++d) f[d] = p[d];
}

__device__ __forceinline__ double gamma_at(const double *__restrict__ gamma, int64_t lane_slots, int64_t u, int64_t beg,
                                           int t, int j, int S) {
  return lane_slots ? gamma[(static_cast<int64_t>(t) * S + j) * lane_slots + u] : gamma[(beg + t) * S + j];
}

__global__ void update_b_utt_sums_kernel(const float *__restrict__ feats, const int64_t *__restrict__ offsets,
                                         int64_t n_utts, int D, int S, const double *__restrict__ gamma,
                                         int64_t lane_slots, double *__restrict__ part,
                                         double *__restrict__ occ_part) {
  const int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (idx >= n_utts * S * D) return;
  const int64_t u = idx / (S * D);
  const int j = static_cast<int>((idx / D) % S), d = static_cast<int>(idx % D);
  const int64_t beg = offsets[u];
  const int T = static_cast<int>(offsets[u + 1] - beg);
  double mu = 0.0, ou = 0.0;
  if (j != 0 && j != S - 1) {
    for (int t = 0; t < T; ++t) {
      const double g = gamma_at(gamma, lane_slots, u, beg, t, j, S);
      mu += g * static_cast<double>(feats[(beg + t) * D + d]);
      ou += g;
    }
  }
  part[idx] = mu;
  if (d == 0) occ_part[u * S + j] = ou;
}

// out[w][k] = sum over rows r (in order) with row_model(r) == w of part[r][k].  Rows are utterances
// (row_model = utt_model, or model 0 when NULL) or, `interleaved`, (chunk, model) pairs: row r belongs
// to model r % W.
template <bool ALL>
__device__ __forceinline__ double fold_rows(const double *__restrict__ part, const int32_t *__restrict__ row_model,
                                            int64_t r, int64_t step, int64_t n_rows, int w, int64_t K, int64_t k) {
  // the adds stay in row order; the loads of kAhead rows are issued together so the loop runs at the add
  // latency instead of the memory latency (a row is its own cache line: 64 of them in flight per lane)
  constexpr int kAhead = ALL ? 64 : 16;
  double acc = 0.0;
  for (; r + (kAhead - 1) * step < n_rows; r += kAhead * step) {
    double v[kAhead];
    int mw[ALL ? 1 : kAhead];
#pragma unroll
    for (int i = 0; i < kAhead; ++i) {
      const int64_t ri = r + i * step;
      v[i] = part[ri * K + k];
      if constexpr (!ALL) mw[i] = row_model[ri];
    }
    __builtin_amdgcn_sched_barrier(0);  // every load in flight before the first add waits on one
#pragma unroll
    for (int i = 0; i < kAhead; ++i) acc = (ALL || mw[ALL ? 0 : i] == w) ? acc + v[i] : acc;
  }
  for (; r < n_rows; r += step)
    if (ALL || row_model[r] == w) acc += part[r * K + k];
  return acc;
}

__global__ __launch_bounds__(64) void update_b_fold_kernel(const double *__restrict__ part,
                                                           const int32_t *__restrict__ row_model, int64_t n_rows,
                                                           int W, int64_t K, int interleaved,
                                                           double *__restrict__ out) {
  const int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (idx >= W * K) return;
  const int w = static_cast<int>(idx / K);
  const int64_t k = idx - static_cast<int64_t>(w) * K;
  if (interleaved)
    out[idx] = fold_rows<true>(part, row_model, w, W, n_rows, w, K, k);
  else if (row_model == nullptr)  // model 0 owns every row when there is no map
    out[idx] = w == 0 ? fold_rows<true>(part, row_model, 0, 1, n_rows, w, K, k) : 0.0;
  else
    out[idx] = fold_rows<false>(part, row_model, 0, 1, n_rows, w, K, k);
}

// The list-order fold of ONE model's rows with the rows staged through LDS (round 4): out[k] = sum_r part[r][k], rows
// added one after another in row order — the chain of update_b_fold_kernel, bit for bit — but the chain's wavefront
// reads its rows from LDS while the other fifteen wavefronts of the workgroup copy the next tile of rows in with
// 16-byte loads.  update_b_fold_kernel's one wavefront fetched its own rows (13 of 64 lanes, 16 rows in flight) and
// ran at the memory latency: 1.30 ms per 100 000 rows of 13 sums; here the chain of dependent float64 additions is
// the critical path.  K <= 64 columns, one workgroup.
constexpr int kFoldLdsDoubles = 6144;  // per buffer (48 KB); two buffers
__global__ __launch_bounds__(1024) void ordered_fold_lds_kernel(const double *__restrict__ part, int64_t n_rows, int K,
                                                                double *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) double buf[2][kFoldLdsDoubles];
  const int tid = threadIdx.x;
  const int tile_rows = (kFoldLdsDoubles / K) & ~1;  // even: a tile starts on a 16-byte boundary of `part`
  const int64_t n_tiles = (n_rows + tile_rows - 1) / tile_rows;
  auto stage = [&](int64_t tile, int b, int first_thread, int n_threads) {
    const int64_t r0 = tile * tile_rows;
    const int64_t rows = (r0 + tile_rows <= n_rows) ? tile_rows : (n_rows - r0);
    const int64_t n = rows * K, n2 = n >> 1;
    const double2 *src = reinterpret_cast<const double2 *>(part + r0 * K);
    double2 *dst = reinterpret_cast<double2 *>(buf[b]);
    int64_t i = tid - first_thread;
    for (; i + 3 * n_threads < n2; i += 4 * n_threads) {  // four 16-byte loads in flight per thread
      const double2 v0 = src[i], v1 = src[i + n_threads], v2 = src[i + 2 * n_threads], v3 = src[i + 3 * n_threads];
      dst[i] = v0;
      dst[i + n_threads] = v1;
      dst[i + 2 * n_threads] = v2;
      dst[i + 3 * n_threads] = v3;
    }
    for (; i < n2; i += n_threads) dst[i] = src[i];
    if ((n & 1) && tid == first_thread) buf[b][n - 1] = part[r0 * K + n - 1];
  };
  double acc = 0.0;
  if (n_tiles > 0) stage(0, 0, 0, 1024);
  __syncthreads();
  for (int64_t tile = 0; tile < n_tiles; ++tile) {
    const int b = static_cast<int>(tile & 1);
    if (tid >= 64) {  // wavefronts 1-15: next tile -> the other buffer
      if (tile + 1 < n_tiles) stage(tile + 1, b ^ 1, 64, 960);
    } else if (tid < K) {  // wavefront 0, one lane per column: the sequential chain
      const int64_t r0 = tile * tile_rows;
      const int rows = static_cast<int>((r0 + tile_rows <= n_rows) ? tile_rows : (n_rows - r0));
      const double *p = buf[b] + tid;
      int r = 0;
      for (; r + 16 <= rows; r += 16) {
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = p[(r + i) * K];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += v[i];
      }
      for (; r < rows; ++r) acc += p[r * K];
    }
    __syncthreads();
  }
  if (tid < K) out[tid] = acc;
}
// list-order fold of all rows (one model): the LDS-staged chain where it applies
inline void launch_ordered_fold(hipStream_t st, const double *part, int64_t n_rows, int64_t K, double *out) {
  if (K <= 64 && (reinterpret_cast<uintptr_t>(part) & 15) == 0)
    SAPR_LAUNCH(ordered_fold_lds_kernel, dim3(1), dim3(1024), 0, st, part, n_rows, static_cast<int>(K), out);
  else
    SAPR_LAUNCH(update_b_fold_kernel, dim3(static_cast<unsigned>((K + 63) / 64)), dim3(64), 0, st, part,
                static_cast<const int32_t *>(nullptr), n_rows, 1, K, 0, out);
}

// The same sums as a FIXED-SHAPE TREE (round 3, the default of the per-iteration folds): a workgroup owns eight
// neighbouring columns of one model (one 64-byte line per row); its 32 row lanes each add every 32nd row of the model
// (four accumulators, rows r, r + 32, r + 64, r + 96 of a lane's sequence), then the 4 x 32 partial sums meet in a
// fixed LDS tree.  The shape depends on (n_rows, W, K) only, so the result is deterministic and the same on every
// rank — but its roundings are not the list-order chain's: differences of a few ulp, far inside the 1e-7 the golden
// Baum-Welch histories are compared at (north star: 1e-5 on log-likelihoods).  The ordered kernel above stays for
// the flat-start sums (bit-identical global mean) and behind SAPR_CUSTOM_FOLD=ordered.
__global__ __launch_bounds__(256) void update_b_fold_tree_kernel(const double *__restrict__ part,
                                                                 const int32_t *__restrict__ row_model, int64_t n_rows,
                                                                 int W, int64_t K, int interleaved,
                                                                 double *__restrict__ out) {
  __shared__ double red[256];
  const int kk = threadIdx.x & 7, rg = threadIdx.x >> 3;
  const int64_t kblocks = (K + 7) / 8;
  const int w = static_cast<int>(blockIdx.x / kblocks);
  const int64_t k = (blockIdx.x - static_cast<int64_t>(w) * kblocks) * 8 + kk;
  if (gridDim.y > 1) {  // two-level fold (launch_fold): this block takes the blockIdx.y-th run of rows -> out[run][W][K]
    const int64_t per = (n_rows + gridDim.y - 1) / gridDim.y, lo = per * blockIdx.y;
    const int64_t hi = lo + per < n_rows ? lo + per : n_rows;
    part += lo * K;
    n_rows = hi > lo ? hi - lo : 0;
    out += static_cast<int64_t>(blockIdx.y) * W * K;
  }
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (k < K && (interleaved || row_model != nullptr || w == 0)) {
    // rows of this model: interleaved -> r % W == w; mapped -> row_model[r] == w; neither -> all rows (model 0)
    const int64_t first = interleaved ? w + static_cast<int64_t>(rg) * W : rg;
    const int64_t step = interleaved ? 32 * static_cast<int64_t>(W) : 32;
    const bool mapped = !interleaved && row_model != nullptr;
    int64_t r = first;
    for (; r + 3 * step < n_rows; r += 4 * step) {
      const double v0 = part[r * K + k], v1 = part[(r + step) * K + k];
      const double v2 = part[(r + 2 * step) * K + k], v3 = part[(r + 3 * step) * K + k];
      if (mapped) {
        a0 += row_model[r] == w ? v0 : 0.0;
        a1 += row_model[r + step] == w ? v1 : 0.0;
        a2 += row_model[r + 2 * step] == w ? v2 : 0.0;
        a3 += row_model[r + 3 * step] == w ? v3 : 0.0;
      } else {
        a0 += v0;
        a1 += v1;
        a2 += v2;
        a3 += v3;
      }
    }
    for (; r < n_rows; r += step)
      if (!mapped || row_model[r] == w) a0 += part[r * K + k];
  }
  red[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
#pragma unroll
  for (int sft = 128; sft >= 8; sft >>= 1) {
    if (threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
    __syncthreads();
  }
  if (threadIdx.x < 8 && k < K) out[static_cast<int64_t>(w) * K + k] = red[threadIdx.x];
}

// per-iteration folds: the tree unless SAPR_CUSTOM_FOLD=ordered asks for the reference's list order
inline bool fold_ordered() {
  const char *e = std::getenv("SAPR_CUSTOM_FOLD");
  return e && std::strcmp(e, "ordered") == 0;
}
inline void launch_fold(hipStream_t st, const double *part, const int32_t *row_model, int64_t n_rows, int W, int64_t K,
                        int interleaved, double *out) {
  if (fold_ordered()) {
    SAPR_LAUNCH(update_b_fold_kernel, dim3(static_cast<unsigned>((static_cast<int64_t>(W) * K + 63) / 64)), dim3(64), 0, st,
                part, row_model, n_rows, W, K, interleaved, out);
  } else {
    const unsigned gx = static_cast<unsigned>(static_cast<int64_t>(W) * ((K + 7) / 8));
    // many rows, few columns (the E-step's 100 000 per-utterance rows of 112 sums: 14 workgroups read 90 MB, 0.39 ms):
    // 128 runs of rows first, on a stream-ordered scratch buffer, then the 128 partial rows — same fixed shape for
    // every rank, 0.39 -> see DESIGN 4.4
    void *scratch = nullptr;
    constexpr unsigned kRuns = 128;
    if (n_rows >= 8192 && gx < 64 && W == 1 && row_model == nullptr && !interleaved &&
        hipMallocAsync(&scratch, static_cast<size_t>(kRuns) * K * sizeof(double), st) == hipSuccess) {
      SAPR_LAUNCH(update_b_fold_tree_kernel, dim3(gx, kRuns), dim3(256), 0, st, part, row_model, n_rows, W, K, 0,
                  static_cast<double *>(scratch));
      SAPR_LAUNCH(update_b_fold_tree_kernel, dim3(gx), dim3(256), 0, st, static_cast<const double *>(scratch), row_model,
                  static_cast<int64_t>(kRuns), W, K, 0, out);
      (void)hipFreeAsync(scratch, st);
      return;
    }
    SAPR_LAUNCH(update_b_fold_tree_kernel, dim3(gx), dim3(256), 0, st, part, row_model, n_rows, W, K, interleaved, out);
  }
}

__global__ void update_b_scatter_kernel(const float *__restrict__ feats, const int64_t *__restrict__ offsets,
                                        const int32_t *__restrict__ utt_model, int64_t n_utts, int64_t per_chunk,
                                        int W, int D, int S, const double *__restrict__ gamma, int64_t lane_slots,
                                        const double *__restrict__ means, double *__restrict__ part) {
  // blockIdx.y = chunk * W + w ; blockIdx.x / threadIdx.x walk (j, a, b)
  const int K = S * D * D;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const int64_t chunk = blockIdx.y / W;
  const int w = static_cast<int>(blockIdx.y % W);
  const int j = k / (D * D), a = (k / D) % D, b = k % D;
  double c = 0.0;
  if (j != 0 && j != S - 1) {
    const double ma = means[(w * S + j) * D + a], mb = means[(w * S + j) * D + b];
    const int64_t u0 = chunk * per_chunk, u1 = u0 + per_chunk < n_utts ? u0 + per_chunk : n_utts;
    for (int64_t u = u0; u < u1; ++u) {
      if ((utt_model ? utt_model[u] : 0) != w) continue;
      const int64_t beg = offsets[u];
      const int T = static_cast<int>(offsets[u + 1] - beg);
      for (int t = 0; t < T; ++t) {
        const double da = static_cast<double>(feats[(beg + t) * D + a]) - ma;
        const double db = static_cast<double>(feats[(beg + t) * D + b]) - mb;
        c += gamma_at(gamma, lane_slots, u, beg, t, j, S) * (da * db);
      }
    }
  }
  part[static_cast<int64_t>(blockIdx.y) * K + k] = c;
}

// pass 2 for ONE model (utt_model == NULL) and compile-time D: one lane per utterance, one state per
// workgroup row, the D (D + 1) / 2 distinct entries of the symmetric scatter matrix in registers; the
// 256 lanes of a tile are then combined in a fixed order (xor-butterfly inside the wavefront, wavefronts
// in sequence).  Each term is gamma * (da * db) exactly as in update_b_scatter_kernel; only the order of
// the additions differs (the reference's own order is that of a BLAS matmul).
__device__ __forceinline__ double wave_sum_f64c(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int D>
__global__ __launch_bounds__(256) void update_b_scatter_lane_kernel(
    const float *__restrict__ feats, const int64_t *__restrict__ offsets, int64_t n_utts, int S,
    const double *__restrict__ gamma, int64_t lane_slots, const double *__restrict__ means,
    double *__restrict__ part) {
  constexpr int kTri = D * (D + 1) / 2;
  __shared__ double red[4][kTri];
  const int64_t tile = blockIdx.x;
  const int j = blockIdx.y + 1;  // emitting states 1 .. S-2
  const int64_t u = tile * 256 + threadIdx.x;
  const bool live = u < n_utts;
  const int64_t beg = live ? offsets[u] : 0;
  const int T = live ? static_cast<int>(offsets[u + 1] - beg) : 0;
  double mu[D];
#pragma unroll
  for (int d = 0; d < D; ++d) mu[d] = means[j * D + d];
  double acc[kTri];
#pragma unroll
  for (int i = 0; i < kTri; ++i) acc[i] = 0.0;
  // The next frame's row and posterior are in flight under the 104 float64 operations of the current one (at two
  // wavefronts per SIMD — 182 accumulator registers — nothing else covers the loads).  acc += (g dx_a) dx_b: one
  // multiplication per dimension and one fused multiply-add per entry, where g (dx_a dx_b) took two operations per
  // entry; same value to a rounding (the reference rounds the product, then the scaling, then the sum).
  float xf[D];
  double g = 0.0;
  if (T > 0) {
    load_row_f32<D>(feats + beg * D, xf);
    g = gamma_at(gamma, lane_slots, u, beg, 0, j, S);
  }
  for (int t = 0; t < T; ++t) {
    double dx[D], gdx[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      dx[d] = static_cast<double>(xf[d]) - mu[d];
      gdx[d] = g * dx[d];
    }
    if (t + 1 < T) {
      load_row_f32<D>(feats + (beg + t + 1) * D, xf);
      g = gamma_at(gamma, lane_slots, u, beg, t + 1, j, S);
    }
    int i = 0;
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = a; b < D; ++b) {
        acc[i] = __builtin_fma(gdx[a], dx[b], acc[i]);
        ++i;
      }
  }
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
#pragma unroll
  for (int i = 0; i < kTri; ++i) {
    const double v = wave_sum_f64c(acc[i]);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  // part[tile][j][a][b] (rows of the fold kernel are tiles; states 0 and S-1 stay zero)
  double *dst = part + static_cast<int64_t>(tile) * S * D * D + static_cast<int64_t>(j) * D * D;
  for (int k = threadIdx.x; k < D * D; k += 256) {
    const int a = k / D, b = k % D;
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    const int i = lo * D - lo * (lo - 1) / 2 + (hi - lo);
    dst[k] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
  }
  if (blockIdx.y == 0)  // the non-emitting states' rows of this tile
    for (int k = threadIdx.x; k < D * D; k += 256) {
      part[static_cast<int64_t>(tile) * S * D * D + k] = 0.0;
      part[static_cast<int64_t>(tile) * S * D * D + static_cast<int64_t>(S - 1) * D * D + k] = 0.0;
    }
}

// x[w][j][...] /= occ[w][j] where occ > 0 (`per` trailing values per state)
__global__ void custom_normalise_kernel(double *__restrict__ x, const double *__restrict__ occ, int64_t n_states,
                                        int per) {
  const int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (idx >= n_states * per) return;
  const double o = occ[idx / per];
  if (o > 0) x[idx] /= o;
}

// flat start (custom_hmm.py:70-92): per-utterance float32 row sums in numpy's pair-wise order, then a
// float64 accumulation in utterance order; and sum of centred outer products.
__device__ float pairwise_f32_strided(const float *p, int n, int stride) { return np_pairwise<float, 8>(p, n, stride); }

// part[u][d] = np.sum(feature, axis=1)[d] of ONE (D,T) float32 array: float32 pair-wise over the T values
// (parallel over utterances); update_b_fold_kernel then adds the utterances in list order in float64
__global__ void custom_global_sum_kernel(const float *__restrict__ feats, const int64_t *__restrict__ offsets,
                                         int64_t n_utts, int D, double *__restrict__ part) {
  const int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (idx >= n_utts * D) return;
  const int64_t u = idx / D;
  const int d = static_cast<int>(idx - u * D);
  const int64_t beg = offsets[u];
  const int T = static_cast<int>(offsets[u + 1] - beg);
  part[idx] = static_cast<double>(pairwise_f32_strided(feats + beg * D + d, T, D));
}

// part[chunk][a][b] = sum over the chunk's frames of (x_a - mean_a)(x_b - mean_b); chunks are folded in order
constexpr int kChunkFrames = 4096;
__global__ void custom_global_cov_kernel(const float *__restrict__ feats, int64_t total_frames, int D,
                                         const double *__restrict__ mean, double *__restrict__ part) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D * D) return;
  const int a = idx / D, b = idx % D;
  const double ma = mean[a], mb = mean[b];
  const int64_t f0 = static_cast<int64_t>(blockIdx.y) * kChunkFrames;
  const int64_t f1 = f0 + kChunkFrames < total_frames ? f0 + kChunkFrames : total_frames;
  double c = 0.0;
  for (int64_t f = f0; f < f1; ++f)
    c += (static_cast<double>(feats[f * D + a]) - ma) * (static_cast<double>(feats[f * D + b]) - mb);
  part[static_cast<int64_t>(blockIdx.y) * D * D + idx] = c;
}

// pass 1 for one model and D-dimensional features, lane per utterance (the layout the gamma lattice wants: one
// coalesced 512-byte row per wavefront and frame; update_b_utt_sums_kernel's lane per (utterance, state, dimension)
// touches five rows of it per wavefront load): block (tile of 256 utterances, emitting state) -> part[tile][S][D],
// occ_part[tile][S]; the fold then runs over tiles.  Fixed-shape sums (butterfly inside a wavefront, the four
// wavefronts in order): deterministic, a few ulp from the per-utterance chain, which SAPR_CUSTOM_FOLD=ordered keeps.
template <int D>
__global__ __launch_bounds__(256) void update_b_sums_lane_kernel(
    const float *__restrict__ feats, const int64_t *__restrict__ offsets, int64_t n_utts, int S,
    const double *__restrict__ gamma, int64_t lane_slots, double *__restrict__ part, double *__restrict__ occ_part) {
  __shared__ double red[4][D + 1];
  const int64_t tile = blockIdx.x;
  const int j = blockIdx.y + 1;  // emitting states 1 .. S-2
  const int64_t u = tile * 256 + threadIdx.x;
  const bool live = u < n_utts;
  const int64_t beg = live ? offsets[u] : 0;
  const int T = live ? static_cast<int>(offsets[u + 1] - beg) : 0;
  double acc[D], occ = 0.0;
#pragma unroll
  for (int d = 0; d < D; ++d) acc[d] = 0.0;
  for (int t = 0; t < T; ++t) {
    const double g = gamma_at(gamma, lane_slots, u, beg, t, j, S);
    float xf[D];
    load_row_f32<D>(feats + (beg + t) * D, xf);
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] += g * static_cast<double>(xf[d]);
    occ += g;
  }
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const double v = wave_sum_f64c(acc[d]);
    if (lane == 0) red[wave][d] = v;
  }
  {
    const double v = wave_sum_f64c(occ);
    if (lane == 0) red[wave][D] = v;
  }
  __syncthreads();
  double *dst = part + (static_cast<int64_t>(tile) * S + j) * D;
  if (threadIdx.x < D) dst[threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
  if (threadIdx.x == D) occ_part[tile * S + j] = ((red[0][D] + red[1][D]) + red[2][D]) + red[3][D];
  if (blockIdx.y == 0 && threadIdx.x < 2 * D + 2) {  // the non-emitting states' rows of this tile
    const int k = threadIdx.x;
    if (k < D) part[(static_cast<int64_t>(tile) * S) * D + k] = 0.0;
    else if (k < 2 * D) part[(static_cast<int64_t>(tile) * S + (S - 1)) * D + (k - D)] = 0.0;
    else if (k == 2 * D) occ_part[tile * S] = 0.0;
    else occ_part[tile * S + S - 1] = 0.0;
  }
}

// ---------------------------------------------------------------------------------------
// update_B as ONE pass of weighted moments on the float64 matrix cores (round 3b; single model, D = 13).
// custom_hmm.py:366-400 is two passes over the data — means, then covariances about the NEW means; both are
// functions of the posterior-weighted moments of the frames about any fixed centre c (x' = x - c):
//     occ_s = sum g_s,   s1_s = sum g_s x',   S2_s = sum g_s x' x'^T
//     mean_s = c + s1_s / occ_s,   cov_s = S2_s / occ_s - d d^T  with  d = s1_s / occ_s
// (algebraically the reference's values; with c = the global mean the subtraction loses about one digit of sixteen).
// All states at once: Gamma (states x rows) times Y (rows x 105), Y = [the 91 products x'_a x'_b (a <= b) | x' | 1],
// one row per (utterance, frame).  v_mfma_f64_16x16x4_f64: M = state, N = column of Y (7 tiles), K = 4 rows — the
// same frame of four neighbouring utterances, so a wavefront's posterior reads of a (frame, state) fall into one
// 128-byte line of the slot-major lattice and every feature row is read once, 52 bytes at a time in sequence.
// The lane-per-utterance kernels this replaces (update_b_sums_lane / update_b_scatter_lane: 1.4 + 1.7 ms per
// 100 000 utterances) re-read the rows once per state through an L1 that 64 private rows per wavefront thrash.
// A wavefront owns 16 utterances at a time (persistent, strided), stages the four rows of a step in its own 2 KB
// of LDS as float64 and forms Y's entries from two ds_read_b64 each.  Partial tiles -> part[wavefront][16][112].
// ---------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int kMomCols = 112;  // 7 tiles of 16: 91 pairs, 13 linear, the constant, 7 unused
template <int D>
__global__ __launch_bounds__(256) void update_b_moments_mfma_kernel(
    const float *__restrict__ feats, const int64_t *__restrict__ offsets, int64_t n_utts, int S,
    const double *__restrict__ gamma, int64_t lane_slots, const double *__restrict__ center,
    double *__restrict__ part) {
  static_assert(D * (D + 1) / 2 + D + 1 <= kMomCols && D <= 13, "column layout");
  constexpr int kTri = D * (D + 1) / 2, NTile = kMomCols / 16;
  __shared__ double s_rows[4][4][4][16];  // [wavefront][group][k][x' (D), 1, 0, 0]
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64, n = lane & 15, k = lane >> 4;
  // the two factors of this lane's column in every tile: indices into a staged row (13 = the constant 1, 14 = 0)
  int ia[NTile], ib[NTile];
#pragma unroll
  for (int tau = 0; tau < NTile; ++tau) {
    const int c = 16 * tau + n;
    int a = 14, b = 14;
    if (c < kTri) {
      int rest = c;
      a = 0;
      while (rest >= D - a) {  // row a of the upper triangle holds D - a entries
        rest -= D - a;
        ++a;
      }
      b = a + rest;
    } else if (c < kTri + D) {
      a = c - kTri;
      b = 13;
    } else if (c == kTri + D) {
      a = b = 13;
    }
    ia[tau] = a;
    ib[tau] = b;
  }
  const double cen = n < D ? center[n] : 0.0;
  f64x4 acc[NTile];
#pragma unroll
  for (int tau = 0; tau < NTile; ++tau) acc[tau] = f64x4{0.0, 0.0, 0.0, 0.0};
  const int64_t n_blocks = (n_utts + 15) / 16;
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * 4, wid = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  const bool emitting = n >= 1 && n <= S - 2;  // A rows: the state is the lane's n
  for (int64_t blk = wid; blk < n_blocks; blk += n_waves) {
    int64_t beg[4], utt[4];
    int T[4];
    int Tmax = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      utt[g] = blk * 16 + 4 * g + k;
      const bool live = utt[g] < n_utts;
      beg[g] = live ? offsets[utt[g]] : 0;
      T[g] = live ? static_cast<int>(offsets[utt[g] + 1] - beg[g]) : 0;
      Tmax = T[g] > Tmax ? T[g] : Tmax;
    }
    Tmax = wave_max_i32(Tmax);
    // the next step's feature values and posteriors are fetched before this step's products: at three wavefronts
    // per SIMD nothing else covers the round trip between a step's loads and its 28 MFMAs
    float xn[4];
    double an[4];
    auto fetch = [&](int t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        xn[g] = (n < D && t < T[g]) ? feats[(beg[g] + t) * D + n] : 0.0f;
        an[g] = (emitting && t < T[g]) ? gamma_at(gamma, lane_slots, utt[g], beg[g], t, n, S) : 0.0;
      }
    };
    if (Tmax > 0) fetch(0);
    for (int t = 0; t < Tmax; ++t) {
      // stage: row (group g, k) = frame t of utterance 16 blk + 4 g + k, as x' with the constants behind it
      double a[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        s_rows[wave][g][k][n] = n < D ? static_cast<double>(xn[g]) - cen : (n == 13 ? 1.0 : 0.0);
        a[g] = an[g];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (t + 1 < Tmax) fetch(t + 1);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const double *row = s_rows[wave][g][k];
#pragma unroll
        for (int tau = 0; tau < NTile; ++tau)
          acc[tau] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g], row[ia[tau]] * row[ib[tau]], acc[tau], 0, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  // C/D of the float64 MFMA: column = lane & 15, row = (lane >> 4) + 4 * register
  double *dst = part + wid * 16 * kMomCols;
#pragma unroll
  for (int tau = 0; tau < NTile; ++tau)
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(k + 4 * r) * kMomCols + 16 * tau + n] = acc[tau][r];
}

// Flat-start covariance sum (custom_hmm.py:81-92) as a Gram product on the float64 matrix cores: with x' = x - mean,
// C[a][b] = sum over frames of x'_a x'_b is A^T A for A = (frames x D), so ONE v_mfma_f64_16x16x4_f64 takes four
// frames: lane (n, k) loads x[4 g + k][n] once — 52 contiguous bytes per frame — and hands x' to the instruction as
// BOTH operands.  (custom_global_cov_kernel reads two values per entry and frame: 338 loads per frame.)  Persistent
// wavefronts, two accumulators in flight, partial 16 x 16 tiles -> part[wavefront][256].
template <int D>
__global__ __launch_bounds__(256) void global_cov_mfma_kernel(const float *__restrict__ feats, int64_t total_frames,
                                                              const double *__restrict__ mean,
                                                              double *__restrict__ part) {
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64, n = lane & 15, k = lane >> 4;
  const double mu = n < D ? mean[n] : 0.0;
  const int64_t n_groups = (total_frames + 3) / 4;
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * 4, wid = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  auto fetch = [&](int64_t g) {
    const int64_t f = 4 * g + k;
    return (n < D && g < n_groups && f < total_frames) ? static_cast<double>(feats[f * D + n]) - mu : 0.0;
  };
  for (int64_t g = wid; g < n_groups; g += 2 * n_waves) {
    const double v0 = fetch(g), v1 = fetch(g + n_waves);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, v1, acc1, 0, 0, 0);
  }
  double *dst = part + wid * 256;
#pragma unroll
  for (int r = 0; r < 4; ++r) dst[(k + 4 * r) * 16 + n] = acc0[r] + acc1[r];  // row = (lane >> 4) + 4 r, column = lane & 15
}
__global__ void global_cov_unpack_kernel(const double *__restrict__ tile, int D, double *__restrict__ cov_out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < D * D) cov_out[idx] = tile[(idx / D) * 16 + idx % D];
}

// feat_t[t][d][slot] = feats[offsets[slot] + t][d] (0 past the utterance's end): the slot-major copy the batched E-step
// reads (one coalesced row per wavefront and value).  A 64 x 64 transpose through LDS: an utterance's frames are
// contiguous floats, so the reads are 256-byte runs per utterance and the writes 256-byte runs per (frame, dimension).
__global__ __launch_bounds__(256) void custom_stage_kernel(const float *__restrict__ feats,
                                                           const int64_t *__restrict__ offsets, int64_t n_utts, int D,
                                                           int max_T, int64_t es, float *__restrict__ out) {
  __shared__ float tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t u0 = static_cast<int64_t>(blockIdx.x) * 64;
  const int64_t r0 = static_cast<int64_t>(blockIdx.y) * 64;  // index into an utterance's T * D floats
  for (int uu = ty; uu < 64; uu += 4) {
    const int64_t u = u0 + uu;
    float v = 0.0f;
    if (u < n_utts) {
      const int64_t beg = offsets[u];
      const int64_t len = (offsets[u + 1] - beg) * D;
      if (r0 + tx < len) v = feats[beg * D + r0 + tx];
    }
    tile[uu][tx] = v;
  }
  __syncthreads();
  const int64_t rows = static_cast<int64_t>(max_T) * D;
  for (int rr = ty; rr < 64; rr += 4)
    if (r0 + rr < rows && u0 + tx < es) out[(r0 + rr) * es + u0 + tx] = tile[tx][rr];
}

// frame_sums[d][slot] = sum_t x_t[d] of the slot's utterance, the frames added one after another in float64 — what
// custom_emit_kernel evaluates per call when it is not given them (custom_hmm.py:168-172's row sum needs sum_s d_s).
// One lane per slot, coalesced rows of the slot-major copy, four frames of loads in flight.
__global__ __launch_bounds__(256) void custom_frame_sums_kernel(const float *__restrict__ feat_t,
                                                               const int64_t *__restrict__ offsets, int64_t n_utts,
                                                               int D, int64_t es, double *__restrict__ frame_sums) {
  const int64_t u = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  const int d = static_cast<int>(blockIdx.y);
  if (u >= es) return;
  double acc = 0.0;
  if (u < n_utts) {
    const int T = static_cast<int>(offsets[u + 1] - offsets[u]);
    const float *__restrict__ p = feat_t + static_cast<int64_t>(d) * es + u;
    const int64_t step = static_cast<int64_t>(D) * es;
    int t = 0;
    for (; t + 4 <= T; t += 4) {
      const float f0 = p[t * step], f1 = p[(t + 1) * step], f2 = p[(t + 2) * step], f3 = p[(t + 3) * step];
      acc += static_cast<double>(f0);
      acc += static_cast<double>(f1);
      acc += static_cast<double>(f2);
      acc += static_cast<double>(f3);
    }
    for (; t < T; ++t) acc += static_cast<double>(p[t * step]);
  }
  frame_sums[static_cast<int64_t>(d) * es + u] = acc;
}

}  // namespace
}  // namespace sapr

using namespace sapr;

static int check_dims(int S, int D) {
  if (S < 3 || S > kMaxS || D < 1 || D > kMaxD)
    return fail(SAPR_ERR_UNSUPPORTED, "custom-HMM kernels support 3 <= S <= %d, D <= %d (got S=%d D=%d)", kMaxS,
                kMaxD, S, D);
  return 0;
}

static int custom_estep_impl(const float *feats, const int64_t *offsets, const int32_t *utt_model,
                             int64_t n_utts, int32_t D, int32_t S, int32_t W, const double *means,
                             const double *inv, const double *cterm, const double *A, const double *logA,
                             int64_t lane_slots, double *E, double *alpha, double *beta, double *gamma,
                             double *xi_dense, double *utt_out, void *stream, const float *feat_t,
                             const double *frame_sums = nullptr) {
  (void)W;
  if (int rc = check_dims(S, D)) return rc;
  SAPR_REQUIRE(n_utts >= 0, "bad n_utts");
  SAPR_REQUIRE(lane_slots == 0 || lane_slots >= n_utts, "lane_slots must be 0 (row layout) or >= n_utts");
  if (n_utts == 0) return 0;
  SAPR_REQUIRE(feats && offsets && means && inv && cterm && A && logA && E && alpha && beta && gamma && utt_out,
               "NULL pointer argument");
  CustomPack P{means, inv, cterm, A, logA};
  const dim3 grid(static_cast<unsigned>((n_utts + kBlock - 1) / kBlock));
  if (lane_slots > 0 && xi_dense == nullptr && (S == 10 || S == 18) && (D == 13 || D == 39)) {
    // batched training shapes: the register-resident kernels (alpha is left UNSHIFTED in its lattice; the beta lattice
    // holds scratch except for the utterances pass 1 redoes).  Two passes: forward + smoothing for every utterance, then
    // the reference-order kernel for the few whose xi terms the reference computes in the denormal range
    hipStream_t st = as_stream(stream);
    int32_t *redo = nullptr;
    SAPR_HIP_TRY(hipMallocAsync(reinterpret_cast<void **>(&redo), (static_cast<size_t>(n_utts) + 1) * sizeof(int32_t), st));
    int32_t *redo_count = redo + n_utts;
    SAPR_HIP_TRY(hipMemsetAsync(redo_count, 0, sizeof(int32_t), st));
    const dim3 egrid(grid.x, static_cast<unsigned>((S - 2) / kEmitStates)), eblock(kBlock * kEmitStates);
#define SAPR_ESTEP_FAST(SS, DD)                                                                                        \
    if (feat_t)                                                                                                         \
      SAPR_LAUNCH((custom_emit_kernel<SS, DD, true>), egrid, eblock, 0, st, feats, offsets, utt_model, n_utts, P, \
                  lane_slots, E, feat_t, frame_sums);                                                                   \
    else                                                                                                                \
      SAPR_LAUNCH((custom_emit_kernel<SS, DD, false>), egrid, eblock, 0, st, feats, offsets, utt_model, n_utts,   \
                  P, lane_slots, E, feat_t, static_cast<const double *>(nullptr));                                      \
    SAPR_LAUNCH((custom_estep_fast_kernel<SS, DD, 0>), grid, dim3(kBlock), 0, st, feats, offsets, utt_model, n_utts, P, \
                lane_slots, E, alpha, beta, gamma, utt_out, redo, redo_count, feat_t);                                  \
    SAPR_LAUNCH((custom_estep_fast_kernel<SS, DD, 1>), grid, dim3(kBlock), 0, st, feats, offsets, utt_model, n_utts, P, \
                lane_slots, E, alpha, beta, gamma, utt_out, redo, redo_count, feat_t)
    if (S == 10 && D == 13) {
      SAPR_ESTEP_FAST(10, 13);
    } else if (S == 18 && D == 39) {
      SAPR_ESTEP_FAST(18, 39);
    } else if (S == 10 && D == 39) {
      SAPR_ESTEP_FAST(10, 39);
    } else {
      SAPR_ESTEP_FAST(18, 13);
    }
#undef SAPR_ESTEP_FAST
    (void)hipFreeAsync(redo, st);
    SAPR_HIP_TRY(hipGetLastError());
    return 0;
  }
  SAPR_LAUNCH(custom_estep_kernel, grid, dim3(kBlock), 0,
                     as_stream(stream), feats, offsets, utt_model, n_utts, D, S, P, lane_slots, E, alpha, beta, gamma,
                     xi_dense, utt_out);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sapr_custom_estep(const float *feats, const int64_t *offsets, const int32_t *utt_model,
                                 int64_t n_utts, int32_t D, int32_t S, int32_t W, const double *means,
                                 const double *inv, const double *cterm, const double *A, const double *logA,
                                 int64_t lane_slots, double *E, double *alpha, double *beta, double *gamma,
                                 double *xi_dense, double *utt_out, void *stream) {
  return custom_estep_impl(feats, offsets, utt_model, n_utts, D, S, W, means, inv, cterm, A, logA, lane_slots, E, alpha,
                           beta, gamma, xi_dense, utt_out, stream, nullptr);
}

// the same with the features also given in slot-major order (sapr_custom_stage_features, same lane_slots): the
// batched training shapes read them from there
extern "C" int sapr_custom_estep_staged(const float *feats, const int64_t *offsets, const int32_t *utt_model,
                                        int64_t n_utts, int32_t D, int32_t S, int32_t W, const double *means,
                                        const double *inv, const double *cterm, const double *A, const double *logA,
                                        int64_t lane_slots, double *E, double *alpha, double *beta, double *gamma,
                                        double *xi_dense, double *utt_out, const float *feat_t,
                                        const double *frame_sums, void *stream) {
  SAPR_REQUIRE(feat_t == nullptr || lane_slots > 0, "staged features need the slot layout (lane_slots > 0)");
  SAPR_REQUIRE(frame_sums == nullptr || feat_t != nullptr, "frame sums come with the staged features");
  return custom_estep_impl(feats, offsets, utt_model, n_utts, D, S, W, means, inv, cterm, A, logA, lane_slots, E, alpha,
                           beta, gamma, xi_dense, utt_out, stream, feat_t, frame_sums);
}

// feat_t[max_T][D][lane_slots] (float32) = the frame-major features in slot-major order, zero past each utterance;
// frame_sums[D][lane_slots] (float64, may be NULL) = every utterance's sum over its frames, in frame order
extern "C" int sapr_custom_stage_features(const float *feats, const int64_t *offsets, int64_t n_utts, int32_t D,
                                          int32_t max_T, int64_t lane_slots, float *feat_t, double *frame_sums,
                                          void *stream) {
  SAPR_REQUIRE(feats && offsets && feat_t && n_utts >= 0 && D > 0 && D <= kMaxD && max_T >= 0, "bad arguments");
  SAPR_REQUIRE(lane_slots >= n_utts && lane_slots > 0, "lane_slots must be >= n_utts");
  const int64_t rows = static_cast<int64_t>(max_T) * D;
  SAPR_REQUIRE((rows + 63) / 64 <= 65535, "max_T * D exceeds one launch's grid");
  if (rows > 0)
    SAPR_LAUNCH(custom_stage_kernel, dim3(static_cast<unsigned>((lane_slots + 63) / 64), static_cast<unsigned>((rows + 63) / 64)),
                dim3(256), 0, as_stream(stream), feats, offsets, n_utts, D, max_T, lane_slots, feat_t);
  SAPR_HIP_TRY(hipGetLastError());
  if (frame_sums) {
    SAPR_LAUNCH(custom_frame_sums_kernel, dim3(static_cast<unsigned>((lane_slots + 255) / 256), static_cast<unsigned>(D)),
                dim3(256), 0, as_stream(stream), feat_t, offsets, n_utts, D, lane_slots, frame_sums);
    SAPR_HIP_TRY(hipGetLastError());
  }
  return 0;
}

extern "C" int sapr_custom_piece(int32_t op, const float *x, int32_t T, int32_t D, int32_t S, const double *means,
                                 const double *inv, const double *cterm, const double *A, const double *logA,
                                 double *E, double *alpha, double *beta, double *gamma, double *xi, double *scalar,
                                 void *stream) {
  if (int rc = check_dims(S, D > 0 ? D : 1)) return rc;
  SAPR_REQUIRE(op >= 0 && op <= 4 && T > 0, "bad op / T");
  CustomPack P{means, inv, cterm, A, logA};
  SAPR_LAUNCH(custom_piece_kernel, dim3(1), dim3(64), 0, as_stream(stream), op, x, T, D, S, P, E, alpha, beta,
                     gamma, xi, scalar);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

template <int DC>
static void launch_emission_exact(hipStream_t st, const float *feats, const int64_t *offsets, int64_t n_utts, int W,
                                  int D, int S, int n_rows, int max_T, const CustomPack &P, double *E) {
  // grid.y covers the longest possible row list; workgroups past an utterance's own list exit at once
  const int64_t max_rows = n_rows ? n_rows : max_T;
  const int64_t chunks = (static_cast<int64_t>(W) * (S - 2) * max_rows + 255) / 256;
  SAPR_LAUNCH((custom_emission_exact_kernel<DC>), dim3(static_cast<unsigned>(n_utts), static_cast<unsigned>(chunks)),
              dim3(256), kLeaf * D * sizeof(double), st, feats, offsets, W, D, S, n_rows, P, E);
}

extern "C" int sapr_custom_emission_exact(const float *feats, const int64_t *offsets, int64_t n_utts, int32_t W,
                                          int32_t D, int32_t S, int32_t n_rows, int32_t max_T,
                                          const double *means, const double *inv, const double *cterm, double *E,
                                          void *stream) {
  if (int rc = check_dims(S, D)) return rc;
  SAPR_REQUIRE(n_utts >= 0 && W > 0 && n_rows >= 0 && max_T >= 0, "bad sizes");
  SAPR_REQUIRE(static_cast<int64_t>(W) * (S - 2) * (n_rows ? n_rows : max_T) <= 65535 * int64_t{256},
               "too many emission rows per utterance for one launch");
  SAPR_REQUIRE(n_rows > 0 || W == 1, "all-frames mode (n_rows == 0) takes one model");
  SAPR_REQUIRE(n_utts < (int64_t{1} << 31), "too many utterances for one launch");
  if (n_utts == 0) return 0;
  SAPR_REQUIRE(feats && offsets && means && inv && cterm && E, "NULL pointer argument");
  CustomPack P{means, inv, cterm, nullptr, nullptr};
  hipStream_t st = as_stream(stream);
  static const bool rows1 = [] {
    const char *e = std::getenv("SAPR_CUSTOM_EMISSION_ROWS1");
    return e && e[0] == '1';
  }();
  if ((D == 13 || D == 39) && !rows1) {
    // one row per lane, differences broadcast over the DPP row (custom_emission_bcast_kernel);
    // SAPR_CUSTOM_EMISSION_ROWS1=1 keeps the one-thread-per-row kernel
    if (D == 13)
      SAPR_LAUNCH(custom_emission_bcast_kernel<13>, dim3(static_cast<unsigned>(n_utts)), dim3(128), 0, st, feats,
                  offsets, W, S, n_rows, P, E);
    else
      SAPR_LAUNCH(custom_emission_bcast_kernel<39>, dim3(static_cast<unsigned>(n_utts)), dim3(256), 0, st, feats,
                  offsets, W, S, n_rows, P, E);
  } else if (D == 13)
    launch_emission_exact<13>(st, feats, offsets, n_utts, W, D, S, n_rows, max_T, P, E);
  else if (D == 39)
    launch_emission_exact<39>(st, feats, offsets, n_utts, W, D, S, n_rows, max_T, P, E);
  else
    launch_emission_exact<0>(st, feats, offsets, n_utts, W, D, S, n_rows, max_T, P, E);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sapr_custom_decode(const float *feats, const int64_t *offsets, int64_t n_utts, int32_t W,
                                  int32_t D, int32_t S, int32_t num_states, int32_t Tq, const double *means,
                                  const double *inv, const double *cterm, const double *A, const double *logA,
                                  double *e_rows, double *scores, int32_t *paths, int32_t *best_word,
                                  double *best_score, int32_t *best_path, void *stream) {
  if (int rc = check_dims(S, D)) return rc;
  SAPR_REQUIRE(n_utts >= 0 && W > 0 && Tq > 0 && Tq <= kMaxD, "bad sizes (Tq <= %d)", kMaxD);
  if (n_utts == 0) return 0;
  SAPR_REQUIRE(feats && offsets && means && inv && cterm && A && logA && e_rows && scores && paths,
               "NULL pointer argument");
  SAPR_REQUIRE((best_word == nullptr) == (best_score == nullptr) && (best_word == nullptr) == (best_path == nullptr),
               "best_word / best_score / best_path go together");
  // every utterance must hold at least Tq frames (the reference raises IndexError otherwise,
  // custom_hmm.py:500): the host mirror checks before the launch
  if (int rc = sapr_custom_emission_exact(feats, offsets, n_utts, W, D, S, Tq, Tq, means, inv, cterm, e_rows, stream))
    return rc;
  CustomPack P{means, inv, cterm, A, logA};
  const int64_t n = n_utts * W;
  hipStream_t st = as_stream(stream);
  const dim3 dgrid(static_cast<unsigned>((n + kBlock - 1) / kBlock));
  static const bool generic_trellis = [] {
    const char *e = std::getenv("SAPR_CUSTOM_DECODE_GENERIC");
    return e && e[0] == '1';
  }();
  if (S == 10 && !generic_trellis)
    SAPR_LAUNCH(custom_decode_lr_kernel<10>, dgrid, dim3(kBlock), 0, st, e_rows, n_utts, W, num_states, Tq, P, scores, paths);
  else if (S == 18 && !generic_trellis)
    SAPR_LAUNCH(custom_decode_lr_kernel<18>, dgrid, dim3(kBlock), 0, st, e_rows, n_utts, W, num_states, Tq, P, scores, paths);
  else
    SAPR_LAUNCH(custom_decode_kernel, dgrid, dim3(kBlock), 0, st, e_rows, n_utts, W, S, num_states, Tq, P, scores, paths);
  if (best_word)
    SAPR_LAUNCH(custom_best_word_kernel, dim3(static_cast<unsigned>((n_utts + 255) / 256)), dim3(256), 0, st, scores,
                paths, n_utts, W, Tq, best_word, best_score, best_path);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

constexpr int kMomWaves = 4096;  // persistent wavefronts of the moments kernel = rows of its fold
static size_t update_b_ws_doubles(int64_t n_utts, int W, int D, int S) {
  const int64_t n = n_utts > 0 ? n_utts : 1;
  const int64_t per = chunk_utts(n, W);
  const int64_t chunks = (n + per - 1) / per;
  const size_t pass1 = static_cast<size_t>(n) * S * D + static_cast<size_t>(n) * S;
  const size_t pass2 = static_cast<size_t>(chunks) * W * S * D * D;  // also covers the (fewer) 256-utterance tiles
  const size_t mom = static_cast<size_t>(kMomWaves) * 16 * kMomCols;  // sapr_custom_update_b_moments
  const size_t m = pass1 > pass2 ? pass1 : pass2;
  return m > mom ? m : mom;
}

extern "C" int sapr_custom_update_b_workspace_bytes(int64_t n_utts, int32_t W, int32_t D, int32_t S, size_t *bytes) {
  if (int rc = check_dims(S, D)) return rc;
  SAPR_REQUIRE(bytes && n_utts >= 0 && W > 0, "bad arguments");
  *bytes = update_b_ws_doubles(n_utts, W, D, S) * sizeof(double);
  return 0;
}

// pass 1, unnormalised: sum_x[W][S][D] and occ[W][S] of this rank's utterances (reference order)
extern "C" int sapr_custom_update_b_sums(const float *feats, const int64_t *offsets, const int32_t *utt_model,
                                         int64_t n_utts, int32_t W, int32_t D, int32_t S, const double *gamma,
                                         int64_t lane_slots, double *sum_x_out, double *occ_out, void *workspace,
                                         size_t ws_bytes, void *stream) {
  if (int rc = check_dims(S, D)) return rc;
  SAPR_REQUIRE(n_utts >= 0 && W > 0, "bad sizes");
  SAPR_REQUIRE(feats && offsets && gamma && sum_x_out && occ_out && workspace, "NULL pointer argument");
  if (ws_bytes < update_b_ws_doubles(n_utts, W, D, S) * sizeof(double))
    return fail(SAPR_ERR_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes,
                update_b_ws_doubles(n_utts, W, D, S) * sizeof(double));
  hipStream_t st = as_stream(stream);
  double *part = static_cast<double *>(workspace);
  double *occ_part = part + static_cast<size_t>(n_utts) * S * D;
  if (utt_model == nullptr && W == 1 && D == 13 && n_utts > 0 && S > 2 && !fold_ordered()) {
    const int64_t tiles = (n_utts + 255) / 256;
    double *occ_tiles = part + static_cast<size_t>(tiles) * S * D;
    SAPR_LAUNCH((update_b_sums_lane_kernel<13>), dim3(static_cast<unsigned>(tiles), static_cast<unsigned>(S - 2)),
                dim3(256), 0, st, feats, offsets, n_utts, S, gamma, lane_slots, part, occ_tiles);
    launch_fold(st, part, nullptr, tiles, 1, static_cast<int64_t>(S) * D, 0, sum_x_out);
    launch_fold(st, occ_tiles, nullptr, tiles, 1, static_cast<int64_t>(S), 0, occ_out);
    SAPR_HIP_TRY(hipGetLastError());
    return 0;
  }
  const int64_t n1 = n_utts * S * D;
  if (n1 > 0)
    SAPR_LAUNCH(update_b_utt_sums_kernel, dim3(static_cast<unsigned>((n1 + 255) / 256)), dim3(256), 0, st, feats,
                offsets, n_utts, D, S, gamma, lane_slots, part, occ_part);
  const int64_t k1 = static_cast<int64_t>(S) * D, k2 = S;
  launch_fold(st, part, utt_model, n_utts, W, k1, 0, sum_x_out);
  launch_fold(st, occ_part, utt_model, n_utts, W, k2, 0, occ_out);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

// both passes as one: out[16][112] = posterior-weighted moments about `center` of this rank's utterances for ONE
// model with 13-dimensional features — row s (states 1 .. S-2, other rows zero): columns 0..90 the upper triangle of
// sum g x'x'^T (row by row), 91..103 sum g x', 104 sum g.  The caller sums across ranks, then
// mean = center + s1 / occ, cov = S2 / occ - (s1 / occ)(s1 / occ)^T.
extern "C" int sapr_custom_update_b_moments(const float *feats, const int64_t *offsets, int64_t n_utts, int32_t D,
                                            int32_t S, const double *gamma, int64_t lane_slots, const double *center,
                                            double *out, void *workspace, size_t ws_bytes, void *stream) {
  if (int rc = check_dims(S, D)) return rc;
  SAPR_REQUIRE(n_utts >= 0, "bad sizes");
  SAPR_REQUIRE(feats && offsets && gamma && center && out && workspace, "NULL pointer argument");
  if (D != 13 || S > 16) return fail(SAPR_ERR_UNSUPPORTED, "moments kernel: D = 13 and at most 16 states (got %d, %d)", D, S);
  if (ws_bytes < update_b_ws_doubles(n_utts, 1, D, S) * sizeof(double))
    return fail(SAPR_ERR_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes,
                update_b_ws_doubles(n_utts, 1, D, S) * sizeof(double));
  hipStream_t st = as_stream(stream);
  double *part = static_cast<double *>(workspace);
  const int64_t n_blocks = (n_utts + 15) / 16;
  int64_t waves = n_blocks < kMomWaves ? n_blocks : kMomWaves;
  if (waves < 1) waves = 1;
  const unsigned grid = static_cast<unsigned>((waves + 3) / 4);
  SAPR_LAUNCH((update_b_moments_mfma_kernel<13>), dim3(grid), dim3(256), 0, st, feats, offsets, n_utts, S, gamma,
              lane_slots, center, part);
  launch_fold(st, part, nullptr, static_cast<int64_t>(grid) * 4, 1, static_cast<int64_t>(16) * kMomCols, 0, out);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

// pass 2, unnormalised: scatter[W][S][D][D] = sum gamma * outer(x - means, x - means) of this rank's utterances
extern "C" int sapr_custom_update_b_scatter(const float *feats, const int64_t *offsets, const int32_t *utt_model,
                                            int64_t n_utts, int32_t W, int32_t D, int32_t S, const double *gamma,
                                            int64_t lane_slots, const double *means, double *scatter_out,
                                            void *workspace, size_t ws_bytes, void *stream) {
  if (int rc = check_dims(S, D)) return rc;
  SAPR_REQUIRE(n_utts >= 0 && W > 0, "bad sizes");
  SAPR_REQUIRE(feats && offsets && gamma && means && scatter_out && workspace, "NULL pointer argument");
  if (ws_bytes < update_b_ws_doubles(n_utts, W, D, S) * sizeof(double))
    return fail(SAPR_ERR_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes,
                update_b_ws_doubles(n_utts, W, D, S) * sizeof(double));
  hipStream_t st = as_stream(stream);
  double *part = static_cast<double *>(workspace);
  const int64_t per = chunk_utts(n_utts, W);
  const int64_t chunks = (n_utts + per - 1) / per;
  const int K = S * D * D;
  if (utt_model == nullptr && W == 1 && D == 13 && n_utts > 0) {
    // single model, 13-dimensional features: lane-per-utterance kernel, rows of the fold are 256-utterance tiles
    const int64_t tiles = (n_utts + 255) / 256;
    SAPR_LAUNCH((update_b_scatter_lane_kernel<13>), dim3(static_cast<unsigned>(tiles), static_cast<unsigned>(S - 2)),
                dim3(256), 0, st, feats, offsets, n_utts, S, gamma, lane_slots, means, part);
    launch_fold(st, part, nullptr, tiles, 1, static_cast<int64_t>(K), 1, scatter_out);
    SAPR_HIP_TRY(hipGetLastError());
    return 0;
  }
  if (chunks > 0)
    SAPR_LAUNCH(update_b_scatter_kernel, dim3(static_cast<unsigned>((K + 255) / 256), static_cast<unsigned>(chunks * W)),
                dim3(256), 0, st, feats, offsets, utt_model, n_utts, per, W, D, S, gamma, lane_slots, means, part);
  launch_fold(st, part, nullptr, chunks * W, W, static_cast<int64_t>(K), 1, scatter_out);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

// out[k] = part[0][k] + part[1][k] + ... in row order (the reference's accumulation over sequences,
// custom_hmm.py:434-439: aggregated_gamma / aggregated_xi / total log-likelihood), on the device
extern "C" int sapr_custom_fold_rows(const double *part, int64_t n_rows, int64_t K, double *out, void *stream) {
  SAPR_REQUIRE(part && out && n_rows >= 0 && K > 0, "bad arguments");
  launch_fold(as_stream(stream), part, nullptr, n_rows, 1, K, 0, out);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

// in place: x[n_states][per] /= occ[n_states] where occ > 0 (after the cross-rank sum, if any)
extern "C" int sapr_custom_normalise(double *x, const double *occ, int64_t n_states, int32_t per, void *stream) {
  SAPR_REQUIRE(x && occ && n_states >= 0 && per > 0, "bad arguments");
  const int64_t n = n_states * per;
  if (n > 0)
    SAPR_LAUNCH(custom_normalise_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                x, occ, n_states, per);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

// single-process convenience: both passes and both normalisations
extern "C" int sapr_custom_update_b(const float *feats, const int64_t *offsets, const int32_t *utt_model,
                                    int64_t n_utts, int32_t W, int32_t D, int32_t S, const double *gamma,
                                    int64_t lane_slots, double *means_out, double *occ_out, double *covs_out,
                                    void *workspace, size_t ws_bytes, void *stream) {
  SAPR_REQUIRE(means_out && occ_out && covs_out, "NULL pointer argument");
  if (int rc = sapr_custom_update_b_sums(feats, offsets, utt_model, n_utts, W, D, S, gamma, lane_slots, means_out,
                                         occ_out, workspace, ws_bytes, stream))
    return rc;
  if (int rc = sapr_custom_normalise(means_out, occ_out, static_cast<int64_t>(W) * S, D, stream)) return rc;
  if (int rc = sapr_custom_update_b_scatter(feats, offsets, utt_model, n_utts, W, D, S, gamma, lane_slots, means_out,
                                            covs_out, workspace, ws_bytes, stream))
    return rc;
  return sapr_custom_normalise(covs_out, occ_out, static_cast<int64_t>(W) * S, D * D, stream);
}

constexpr int kCovWaves = 8192;  // persistent wavefronts of global_cov_mfma_kernel = rows of its fold
extern "C" int sapr_custom_global_workspace_bytes(int64_t n_utts, int64_t total_frames, int32_t D, size_t *bytes) {
  SAPR_REQUIRE(bytes && n_utts >= 0 && total_frames >= 0 && D > 0 && D <= kMaxD, "bad arguments");
  const size_t s1 = static_cast<size_t>(n_utts > 0 ? n_utts : 1) * D;
  const size_t s2 = static_cast<size_t>((total_frames + kChunkFrames - 1) / kChunkFrames + 1) * D * D;
  const size_t s3 = static_cast<size_t>(kCovWaves + 1) * 256;  // partial tiles of the matrix-core covariance + their sum
  const size_t m = s1 > s2 ? s1 : s2;
  *bytes = (m > s3 ? m : s3) * sizeof(double);
  return 0;
}

extern "C" int sapr_custom_global_sum(const float *feats, const int64_t *offsets, int64_t n_utts, int32_t D,
                                      double *sum_out, void *workspace, size_t ws_bytes, void *stream) {
  SAPR_REQUIRE(feats && offsets && sum_out && workspace && D > 0 && D <= kMaxD && n_utts >= 0, "bad arguments");
  SAPR_REQUIRE(ws_bytes >= static_cast<size_t>(n_utts) * D * sizeof(double), "workspace too small");
  double *part = static_cast<double *>(workspace);
  const int64_t n = n_utts * D;
  if (n > 0)
    SAPR_LAUNCH(custom_global_sum_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                feats, offsets, n_utts, D, part);
  launch_ordered_fold(as_stream(stream), part, n_utts, D, sum_out);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sapr_custom_global_cov(const float *feats, int64_t total_frames, int32_t D, const double *mean,
                                      double *cov_out, void *workspace, size_t ws_bytes, void *stream) {
  SAPR_REQUIRE(feats && mean && cov_out && workspace && D > 0 && D <= kMaxD && total_frames >= 0, "bad arguments");
  if (D == 13 && total_frames > 0 && !fold_ordered() && ws_bytes >= static_cast<size_t>(kCovWaves + 1) * 256 * sizeof(double)) {
    hipStream_t st = as_stream(stream);
    double *part = static_cast<double *>(workspace);
    const int64_t n_groups = (total_frames + 3) / 4;
    int64_t waves = n_groups < kCovWaves ? n_groups : kCovWaves;
    const unsigned grid = static_cast<unsigned>((waves + 3) / 4);
    double *tile = part + static_cast<size_t>(grid) * 4 * 256;
    SAPR_LAUNCH((global_cov_mfma_kernel<13>), dim3(grid), dim3(256), 0, st, feats, total_frames, mean, part);
    launch_fold(st, part, nullptr, static_cast<int64_t>(grid) * 4, 1, 256, 0, tile);
    SAPR_LAUNCH(global_cov_unpack_kernel, dim3(1), dim3(256), 0, st, tile, D, cov_out);
    SAPR_HIP_TRY(hipGetLastError());
    return 0;
  }
  const int64_t chunks = (total_frames + kChunkFrames - 1) / kChunkFrames;
  SAPR_REQUIRE(ws_bytes >= static_cast<size_t>(chunks) * D * D * sizeof(double), "workspace too small");
  SAPR_REQUIRE(chunks <= 65535, "too many frames for one launch: shard the feature list");
  double *part = static_cast<double *>(workspace);
  if (chunks > 0)
    SAPR_LAUNCH(custom_global_cov_kernel, dim3((D * D + 63) / 64, static_cast<unsigned>(chunks)), dim3(64), 0,
                as_stream(stream), feats, total_frames, D, mean, part);
  SAPR_LAUNCH(update_b_fold_kernel, dim3((D * D + 63) / 64), dim3(64), 0, as_stream(stream), part,
              static_cast<const int32_t *>(nullptr), chunks, 1, static_cast<int64_t>(D) * D, 1, cov_out);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}
