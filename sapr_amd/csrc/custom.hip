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
// float64; exp/log1p/log from the device math library (agreement ~1e-13, tests use 1e-9).
#include "sapr_common.h"

namespace sapr {
namespace {

constexpr int kMaxS = 20, kMaxD = 40;
constexpr int kBlock = 64;

// numpy's logaddexp (npy_logaddexp)
__device__ __forceinline__ double np_logaddexp(double a, double b) {
  if (a == b) return a + 0.693147180559945309417232121458176568;  // handles inf == inf
  const double tmp = a - b;
  if (tmp > 0) return a + log1p(exp(-tmp));
  if (tmp <= 0) return b + log1p(exp(tmp));
  return tmp;  // NaN
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

struct CustomPack {
  const double *means;   // [W][S][D]
  const double *inv;     // [W][S][D][D]  inverse of (cov + 1e-6 I), numpy/LAPACK on the host
  const double *cterm;   // [W][S]        D*log(2*pi) + logdet
  const double *A;       // [W][S][S]
  const double *logA;    // [W][S][S]     np.log(A) (−inf for zeros)
};

// E[t][j] for all frames of one utterance under model w; E has row stride S
__device__ void emission_rows(const float *__restrict__ x, int T, int D, int S, const CustomPack &P, int w,
                              double *__restrict__ E) {
  double xs[kMaxD], v[kMaxD];
  for (int d = 0; d < D; ++d) xs[d] = 0.0;
  for (int t = 0; t < T; ++t)
    for (int d = 0; d < D; ++d) xs[d] += static_cast<double>(x[static_cast<int64_t>(t) * D + d]);
  for (int t = 0; t < T; ++t) {
    E[static_cast<int64_t>(t) * S] = neg_inf();
    E[static_cast<int64_t>(t) * S + S - 1] = neg_inf();
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
      E[static_cast<int64_t>(t) * S + j] = -0.5 * (c + qd);
    }
  }
}

// ---- the four recurrences as device functions (rows have stride S) -------------------------------
// forward (custom_hmm.py:176-211): returns the global scale = max(alpha), alpha is stored shifted
__device__ double forward_rows(const double *__restrict__ E, const double *__restrict__ lA, int T, int S,
                               double *__restrict__ al) {
  for (int s = 0; s < S; ++s) al[s] = neg_inf();
  al[0] = 0.0;
  al[1] = lA[0 * S + 1] + E[1];
  for (int t = 1; t < T; ++t) {
    const double *p = al + static_cast<int64_t>(t - 1) * S;
    double *c = al + static_cast<int64_t>(t) * S;
    const double *e = E + static_cast<int64_t>(t) * S;
    c[0] = neg_inf();
    for (int j = 1; j < S - 1; ++j)
      c[j] = np_logaddexp(p[j - 1] + lA[(j - 1) * S + j], p[j] + lA[j * S + j]) + e[j];
    c[S - 1] = p[S - 2] + lA[(S - 2) * S + S - 1];
  }
  double scale = neg_inf();
  for (int64_t i = 0; i < static_cast<int64_t>(T) * S; ++i) {
    const double v = al[i];
    if (v > scale || v != v) scale = v;  // np.max propagates NaN
  }
  for (int64_t i = 0; i < static_cast<int64_t>(T) * S; ++i) al[i] -= scale;
  return scale;
}

// backward (custom_hmm.py:213-246)
__device__ void backward_rows(const double *__restrict__ E, const double *__restrict__ lA, int T, int S,
                              double scale, double *__restrict__ be) {
  for (int64_t i = 0; i < static_cast<int64_t>(T) * S; ++i) be[i] = neg_inf();
  be[static_cast<int64_t>(T - 1) * S + S - 1] = 0.0;
  for (int t = T - 2; t >= 0; --t) {
    const double *n = be + static_cast<int64_t>(t + 1) * S;
    const double *e = E + static_cast<int64_t>(t + 1) * S;
    double *c = be + static_cast<int64_t>(t) * S;
    c[0] = lA[0 * S + 1] + e[1] + n[1];
    for (int i = 1; i < S - 2; ++i)
      c[i] = np_logaddexp(lA[i * S + i] + e[i] + n[i], lA[i * S + i + 1] + e[i + 1] + n[i + 1]);
    {
      const int i = S - 2;
      c[i] = np_logaddexp(lA[i * S + i] + e[i] + n[i], lA[i * S + i + 1] + n[i + 1]);
    }
  }
  for (int t = 0; t < T - 1; ++t)
    for (int s = 0; s < S; ++s) be[static_cast<int64_t>(t) * S + s] -= scale;
}

// gamma (custom_hmm.py:248-257): row soft-max of alpha + beta via logaddexp.reduce
__device__ void gamma_rows(const double *__restrict__ al, const double *__restrict__ be, int T, int S,
                           double *__restrict__ ga) {
  for (int t = 0; t < T; ++t) {
    const double *a = al + static_cast<int64_t>(t) * S, *b = be + static_cast<int64_t>(t) * S;
    double *g = ga + static_cast<int64_t>(t) * S;
    double norm = a[0] + b[0];
    for (int s = 1; s < S; ++s) norm = np_logaddexp(norm, a[s] + b[s]);
    for (int s = 0; s < S; ++s) g[s] = exp((a[s] + b[s]) - norm);
  }
}

__device__ double seq_loglik(const double *__restrict__ al, int T, int S) {
  const double *a = al + static_cast<int64_t>(T - 1) * S;
  double ll = a[0];
  for (int s = 1; s < S; ++s) ll = np_logaddexp(ll, a[s]);
  return ll;
}

// xi (custom_hmm.py:259-322), renormalised per frame; exit column uses emission = -inf.
// xi_dense (optional) receives rows t < T-1 of [S][S]; agg (optional) accumulates sum_t xi[t].
__device__ void xi_rows(const double *__restrict__ al, const double *__restrict__ be, const double *__restrict__ E,
                        const double *__restrict__ A, const double *__restrict__ lA, int T, int S,
                        double *__restrict__ xi_dense, double *__restrict__ agg) {
  const double ll = seq_loglik(al, T, S);
  double xr[kMaxS * kMaxS];
  for (int t = 0; t < T - 1; ++t) {
    const double *a = al + static_cast<int64_t>(t) * S;
    const double *e = E + static_cast<int64_t>(t + 1) * S;
    const double *b = be + static_cast<int64_t>(t + 1) * S;
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

// one lane = one utterance against model utt_model[u]; lattices E/alpha/beta/gamma are [total_frames][S]
// rows at the utterance's frame offset; xi_dense (optional) is [total_frames][S][S] (rows t < T-1 used).
// utt_out[u] = {LL (scaled alpha, logaddexp.reduce(alpha[-1])), scale, agg_gamma[S], agg_xi[S][S]}
__global__ __launch_bounds__(kBlock) void custom_estep_kernel(
    const float *__restrict__ feats, const int64_t *__restrict__ offsets,
    const int32_t *__restrict__ utt_model, int64_t n_utts, int D, int S, CustomPack P,
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
  double *E = Eo + beg * S, *al = alpha + beg * S, *be = beta + beg * S, *ga = gamma + beg * S;
  const double *lA = P.logA + static_cast<int64_t>(w) * S * S;
  const double *A = P.A + static_cast<int64_t>(w) * S * S;

  emission_rows(feats + beg * D, T, D, S, P, w, E);
  const double scale = forward_rows(E, lA, T, S, al);
  backward_rows(E, lA, T, S, scale, be);
  gamma_rows(al, be, T, S, ga);
  for (int t = 0; t < T - 1; ++t)  // aggregated_gamma += sum(gamma[:-1])   (custom_hmm.py:434)
    for (int s = 0; s < S; ++s) out[2 + s] += ga[static_cast<int64_t>(t) * S + s];
  out[0] = seq_loglik(al, T, S);  // of the SCALED alpha (custom_hmm.py:438)
  out[1] = scale;
  xi_rows(al, be, E, A, lA, T, S, xi_dense ? xi_dense + beg * S * S : nullptr, out + 2 + S);
}

// single-utterance pieces with caller-supplied inputs (the reference's per-method API, used by its
// tests): op 0 emission(features) 1 forward(E) 2 backward(E, scale) 3 gamma(alpha, beta) 4 xi(alpha, beta, E)
__global__ void custom_piece_kernel(int op, const float *__restrict__ x, int T, int D, int S, CustomPack P,
                                    double *__restrict__ E, double *__restrict__ al, double *__restrict__ be,
                                    double *__restrict__ ga, double *__restrict__ xi, double *__restrict__ scalar) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (op == 0) emission_rows(x, T, D, S, P, 0, E);
  if (op == 1) scalar[0] = forward_rows(E, P.logA, T, S, al);
  if (op == 2) backward_rows(E, P.logA, T, S, scalar[0], be);
  if (op == 3) gamma_rows(al, be, T, S, ga);
  if (op == 4) xi_rows(al, be, E, P.A, P.logA, T, S, xi, nullptr);
}

// Viterbi of custom_hmm.py:462-514 for every (utterance, model): the emission matrix covers ALL T
// frames (the row-sum term needs them), the trellis only the first Tq.  scores[u][w]; paths[u][w][Tq].
__global__ __launch_bounds__(kBlock) void custom_decode_kernel(
    const float *__restrict__ feats, const int64_t *__restrict__ offsets, int64_t n_utts, int W, int D,
    int S, int num_states, int Tq, CustomPack P, double *__restrict__ Escratch /* [n_utts*W][Tmax][S] */,
    int Tmax, double *__restrict__ scores, int32_t *__restrict__ paths) {
  const int64_t idx = blockIdx.x * static_cast<int64_t>(kBlock) + threadIdx.x;
  if (idx >= n_utts * W) return;
  const int64_t u = idx / W;
  const int w = static_cast<int>(idx - u * W);
  const int64_t beg = offsets[u];
  const int T = static_cast<int>(offsets[u + 1] - beg);
  double *E = Escratch + idx * static_cast<int64_t>(Tmax) * S;
  emission_rows(feats + beg * D, T, D, S, P, w, E);
  const double *lA = P.logA + static_cast<int64_t>(w) * S * S;

  double V[kMaxS], Vn[kMaxS];
  int32_t *bp = paths + idx * static_cast<int64_t>(Tq);  // reused below: first as scratch row store
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

// update_B (custom_hmm.py:366-400), pass 1: means[j][d] = sum_u sum_t gamma[t][j] x[t][d] / occ[j]
// One lane per (model, state, dim); utterances of the model are visited in list order, frames in order.
__global__ void custom_update_means_kernel(const float *__restrict__ feats, const int64_t *__restrict__ offsets,
                                           const int32_t *__restrict__ utt_model, int64_t n_utts, int W, int D,
                                           int S, const double *__restrict__ gamma, double *__restrict__ means,
                                           double *__restrict__ occ) {
  const int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (idx >= static_cast<int64_t>(W) * S * D) return;
  const int w = static_cast<int>(idx / (S * D));
  const int j = static_cast<int>((idx / D) % S), d = static_cast<int>(idx % D);
  if (j == 0 || j == S - 1) {
    means[idx] = 0.0;
    if (d == 0) occ[w * S + j] = 0.0;
    return;
  }
  double m = 0.0, o = 0.0;
  for (int64_t u = 0; u < n_utts; ++u) {
    if ((utt_model ? utt_model[u] : 0) != w) continue;
    const int64_t beg = offsets[u];
    const int T = static_cast<int>(offsets[u + 1] - beg);
    // np.sum(gamma[:, j:j+1] * features.T, axis=0): per-utterance column sums, then added
    double mu = 0.0, ou = 0.0;
    for (int t = 0; t < T; ++t) {
      const double g = gamma[(beg + t) * S + j];
      mu += g * static_cast<double>(feats[(beg + t) * D + d]);
      ou += g;
    }
    m += mu;
    o += ou;
  }
  if (o > 0) m /= o;
  means[idx] = m;
  if (d == 0) occ[w * S + j] = o;
}

// pass 2: covs[j] = sum gamma[t][j] * outer(x_t - mean_j, x_t - mean_j) / occ[j], symmetrised,
// diagonal floored
__global__ void custom_update_covs_kernel(const float *__restrict__ feats, const int64_t *__restrict__ offsets,
                                          const int32_t *__restrict__ utt_model, int64_t n_utts, int W, int D,
                                          int S, const double *__restrict__ gamma,
                                          const double *__restrict__ means, const double *__restrict__ occ,
                                          double *__restrict__ covs_raw) {
  const int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (idx >= static_cast<int64_t>(W) * S * D * D) return;
  const int w = static_cast<int>(idx / (static_cast<int64_t>(S) * D * D));
  const int j = static_cast<int>((idx / (D * D)) % S);
  const int a = static_cast<int>((idx / D) % D), b = static_cast<int>(idx % D);
  if (j == 0 || j == S - 1) {
    covs_raw[idx] = 0.0;
    return;
  }
  const double ma = means[(w * S + j) * D + a], mb = means[(w * S + j) * D + b];
  double c = 0.0;
  for (int64_t u = 0; u < n_utts; ++u) {
    if ((utt_model ? utt_model[u] : 0) != w) continue;
    const int64_t beg = offsets[u];
    const int T = static_cast<int>(offsets[u + 1] - beg);
    for (int t = 0; t < T; ++t) {
      const double da = static_cast<double>(feats[(beg + t) * D + a]) - ma;
      const double db = static_cast<double>(feats[(beg + t) * D + b]) - mb;
      c += gamma[(beg + t) * S + j] * (da * db);
    }
  }
  const double o = occ[w * S + j];
  covs_raw[idx] = o > 0 ? c / o : c;
}

// flat start (custom_hmm.py:70-92): per-utterance float32 row sums in numpy's pair-wise order, then a
// float64 accumulation in utterance order; and sum of centred outer products.
__device__ float pairwise_f32_strided(const float *p, int n, int stride) { return np_pairwise<float, 8>(p, n, stride); }

__global__ void custom_global_sum_kernel(const float *__restrict__ feats, const int64_t *__restrict__ offsets,
                                         int64_t n_utts, int D, double *__restrict__ sum_out) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  double acc = 0.0;
  for (int64_t u = 0; u < n_utts; ++u) {
    const int64_t beg = offsets[u];
    const int T = static_cast<int>(offsets[u + 1] - beg);
    // np.sum(feature, axis=1) on the float32 (D,T) array: float32 pair-wise over the T contiguous values
    acc += static_cast<double>(pairwise_f32_strided(feats + beg * D + d, T, D));
  }
  sum_out[d] = acc;
}

__global__ void custom_global_cov_kernel(const float *__restrict__ feats, int64_t total_frames, int D,
                                         const double *__restrict__ mean, double *__restrict__ cov_out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D * D) return;
  const int a = idx / D, b = idx % D;
  const double ma = mean[a], mb = mean[b];
  double c = 0.0;
  for (int64_t f = 0; f < total_frames; ++f)
    c += (static_cast<double>(feats[f * D + a]) - ma) * (static_cast<double>(feats[f * D + b]) - mb);
  cov_out[idx] = c;
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

extern "C" int sapr_custom_estep(const float *feats, const int64_t *offsets, const int32_t *utt_model,
                                 int64_t n_utts, int32_t D, int32_t S, int32_t W, const double *means,
                                 const double *inv, const double *cterm, const double *A, const double *logA,
                                 double *E, double *alpha, double *beta, double *gamma, double *xi_dense,
                                 double *utt_out, void *stream) {
  (void)W;
  if (int rc = check_dims(S, D)) return rc;
  SAPR_REQUIRE(n_utts >= 0, "bad n_utts");
  if (n_utts == 0) return 0;
  SAPR_REQUIRE(feats && offsets && means && inv && cterm && A && logA && E && alpha && beta && gamma && utt_out,
               "NULL pointer argument");
  CustomPack P{means, inv, cterm, A, logA};
  SAPR_LAUNCH(custom_estep_kernel, dim3(static_cast<unsigned>((n_utts + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     as_stream(stream), feats, offsets, utt_model, n_utts, D, S, P, E, alpha, beta, gamma, xi_dense,
                     utt_out);
  SAPR_HIP_TRY(hipGetLastError());
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

extern "C" int sapr_custom_decode(const float *feats, const int64_t *offsets, int64_t n_utts, int32_t W,
                                  int32_t D, int32_t S, int32_t num_states, int32_t Tq, int32_t max_T,
                                  const double *means, const double *inv, const double *cterm, const double *A,
                                  const double *logA, double *e_scratch, double *scores, int32_t *paths,
                                  void *stream) {
  if (int rc = check_dims(S, D)) return rc;
  SAPR_REQUIRE(n_utts >= 0 && W > 0 && Tq > 0 && Tq <= kMaxD && max_T >= Tq, "bad sizes (Tq <= %d, max_T >= Tq)", kMaxD);
  if (n_utts == 0) return 0;
  SAPR_REQUIRE(feats && offsets && means && inv && cterm && A && logA && e_scratch && scores && paths,
               "NULL pointer argument");
  CustomPack P{means, inv, cterm, A, logA};
  const int64_t n = n_utts * W;
  SAPR_LAUNCH(custom_decode_kernel, dim3(static_cast<unsigned>((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     as_stream(stream), feats, offsets, n_utts, W, D, S, num_states, Tq, P, e_scratch, max_T, scores,
                     paths);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sapr_custom_update_b(const float *feats, const int64_t *offsets, const int32_t *utt_model,
                                    int64_t n_utts, int32_t W, int32_t D, int32_t S, const double *gamma,
                                    double *means_out, double *occ_out, double *covs_out, void *stream) {
  if (int rc = check_dims(S, D)) return rc;
  SAPR_REQUIRE(feats && offsets && gamma && means_out && occ_out && covs_out, "NULL pointer argument");
  const int64_t n1 = static_cast<int64_t>(W) * S * D, n2 = n1 * D;
  SAPR_LAUNCH(custom_update_means_kernel, dim3(static_cast<unsigned>((n1 + 63) / 64)), dim3(64), 0,
                     as_stream(stream), feats, offsets, utt_model, n_utts, W, D, S, gamma, means_out, occ_out);
  SAPR_LAUNCH(custom_update_covs_kernel, dim3(static_cast<unsigned>((n2 + 63) / 64)), dim3(64), 0,
                     as_stream(stream), feats, offsets, utt_model, n_utts, W, D, S, gamma, means_out, occ_out,
                     covs_out);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sapr_custom_global_sum(const float *feats, const int64_t *offsets, int64_t n_utts, int32_t D,
                                      double *sum_out, void *stream) {
  SAPR_REQUIRE(feats && offsets && sum_out && D > 0 && D <= kMaxD, "bad arguments");
  SAPR_LAUNCH(custom_global_sum_kernel, dim3(1), dim3(64), 0, as_stream(stream), feats, offsets, n_utts, D,
                     sum_out);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sapr_custom_global_cov(const float *feats, int64_t total_frames, int32_t D, const double *mean,
                                      double *cov_out, void *stream) {
  SAPR_REQUIRE(feats && mean && cov_out && D > 0 && D <= kMaxD, "bad arguments");
  SAPR_LAUNCH(custom_global_cov_kernel, dim3((D * D + 63) / 64), dim3(64), 0, as_stream(stream), feats,
                     total_frames, D, mean, cov_out);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}
