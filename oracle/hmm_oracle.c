/* CPU oracle in C: hmmlearn 0.3.3 GaussianHMM (diag) log-density, Viterbi, forward, backward.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: hmmlearn is an
 * un-vendored dependency (assignment2/poetry.lock:430-431) absent from the container; this
 * restates its published algorithm (hmmlearn/stats.py _log_multivariate_normal_density_diag;
 * hmmlearn/_hmmc.cpp viterbi / forward_log / backward_log) for the reference's call sites
 * decoder.py:43 and hmmlearn_hmm.py:103-104.  It exists so that parity can be checked at the
 * benchmark's FULL sizes (the numpy restatement oracle/hmmlearn_oracle.py is the readable
 * one and is checked against this file in tests/test_oracle_hmmlearn.py) and so that the CPU
 * baseline of bench.py times compiled code, as the reference's hmmlearn path is compiled C++.
 *
 * Build: oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp).  Logs of startprob/transmat and
 * the per-state constant come from numpy on the host (np.log is not bit-identical to libm).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NEG_INF (-INFINITY)

/* numpy pairwise sum over a contiguous run of n <= 128 doubles
   (numpy/core/src/umath/loops_utils.h.src DOUBLE_pairwise_sum) */
static double np_pairwise_sum(const double *a, int n) {
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += a[i];
    return r;
  }
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

/* logB[t][s] = -0.5 * (gconst[s] + sum_d ((x[t][d] - mu[s][d])^2 / var[s][d]))   (stats.py)
   sum_order 0: numpy pair-wise (X C-contiguous (T,D): fit/score, hmmlearn_hmm.py:80-81);
   sum_order 1: X is the transposed view of a (D,T) array (decoder.py:59): numpy lays the
   (T,S,D) temporary out t-fastest and adds the D slices left to right — unless T == 1, where
   the view is C-contiguous and the reduction is pair-wise again.  Rule pinned against numpy
   in tests/test_oracle_hmmlearn.py::test_sum_order_rule_matches_numpy. */
void oracle_log_density_diag(const float *X, int T, int D, const double *means, const double *vars,
                             const double *gconst, int S, int sum_order, double *logB) {
  double q[1024];
  const int seq = sum_order && T > 1;
  for (int t = 0; t < T; ++t)
    for (int s = 0; s < S; ++s) {
      for (int d = 0; d < D; ++d) {
        double df = (double)X[(size_t)t * D + d] - means[s * D + d];
        q[d] = (df * df) / vars[s * D + d];
      }
      double quad;
      if (seq) {
        quad = q[0];
        for (int d = 1; d < D; ++d) quad += q[d];
      } else {
        quad = np_pairwise_sum(q, D);
      }
      logB[(size_t)t * S + s] = -0.5 * (gconst[s] + quad);
    }
}

/* _hmmc.cpp viterbi; tie: 0 = lower index on equal back-trace scores, 1 = higher (0.3.x pair max) */
double oracle_viterbi(const double *log_start, const double *log_trans, const double *logB, int T,
                      int S, int tie, int32_t *states) {
  double *lat = (double *)malloc(sizeof(double) * (size_t)T * S);
  for (int i = 0; i < S; ++i) lat[i] = log_start[i] + logB[i];
  for (int t = 1; t < T; ++t)
    for (int i = 0; i < S; ++i) {
      double m = NEG_INF;
      for (int j = 0; j < S; ++j) {
        double v = lat[(size_t)(t - 1) * S + j] + log_trans[j * S + i];
        if (v > m) m = v; /* std::max */
      }
      lat[(size_t)t * S + i] = m + logB[(size_t)t * S + i];
    }
  const double *row = lat + (size_t)(T - 1) * S;
  int prev = 0;
  for (int i = 1; i < S; ++i)
    if (row[i] > row[prev]) prev = i; /* std::max_element: first maximum */
  double lp = row[prev];
  states[T - 1] = prev;
  for (int t = T - 2; t >= 0; --t) {
    double best = NEG_INF;
    int arg = 0;
    for (int i = 0; i < S; ++i) {
      double v = lat[(size_t)t * S + i] + log_trans[i * S + prev];
      int take = tie ? (best < v || (best == v && arg < i)) : (v > best);
      if (take) {
        best = v;
        arg = i;
      }
    }
    prev = arg;
    states[t] = prev;
  }
  free(lat);
  return lp;
}

static double logsumexp(const double *v, int n) {
  double m = v[0];
  for (int i = 1; i < n; ++i)
    if (v[i] > m) m = v[i];
  if (isinf(m)) return m;
  double acc = 0.0;
  for (int i = 0; i < n; ++i) acc += exp(v[i] - m);
  return log(acc) + m;
}

/* _hmmc.cpp forward_log → returns log_prob, fills fwd[T][S] */
double oracle_forward_log(const double *log_start, const double *log_trans, const double *logB,
                          int T, int S, double *fwd) {
  double work[256];
  for (int i = 0; i < S; ++i) fwd[i] = log_start[i] + logB[i];
  for (int t = 1; t < T; ++t)
    for (int j = 0; j < S; ++j) {
      for (int i = 0; i < S; ++i) work[i] = fwd[(size_t)(t - 1) * S + i] + log_trans[i * S + j];
      fwd[(size_t)t * S + j] = logsumexp(work, S) + logB[(size_t)t * S + j];
    }
  return logsumexp(fwd + (size_t)(T - 1) * S, S);
}

/* _hmmc.cpp backward_log */
void oracle_backward_log(const double *log_trans, const double *logB, int T, int S, double *bwd) {
  double work[256];
  for (int i = 0; i < S; ++i) bwd[(size_t)(T - 1) * S + i] = 0.0;
  for (int t = T - 2; t >= 0; --t)
    for (int i = 0; i < S; ++i) {
      for (int j = 0; j < S; ++j)
        work[j] = log_trans[i * S + j] + logB[(size_t)(t + 1) * S + j] + bwd[(size_t)(t + 1) * S + j];
      bwd[(size_t)t * S + i] = logsumexp(work, S);
    }
}

/* decoder.py:35-49 over a ragged frame-major batch: every utterance against W models
   (GaussianHMM.decode), strict '>' arg-max in model order; path of the winner (model 0 if none).
   `which`: 0 = Viterbi scores (+ paths), 1 = forward log-likelihoods (no paths). */
void oracle_decode_batch(const float *feats, const int64_t *offsets, int64_t n_utts, int D,
                         const double *means, const double *vars, const double *gconst,
                         const double *log_start, const double *log_trans, int W, int S, int tie,
                         int sum_order, int which, double *scores, int32_t *best_word,
                         int32_t *path) {
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t u = 0; u < n_utts; ++u) {
    const int64_t beg = offsets[u];
    const int T = (int)(offsets[u + 1] - beg);
    if (T <= 0) {
      for (int w = 0; w < W; ++w) scores[u * W + w] = NEG_INF;
      if (best_word) best_word[u] = -1;
      continue;
    }
    double *logB = (double *)malloc(sizeof(double) * (size_t)T * S);
    double *fwd = which ? (double *)malloc(sizeof(double) * (size_t)T * S) : NULL;
    int32_t *st = (int32_t *)malloc(sizeof(int32_t) * (size_t)T * (size_t)W);
    double bs = NEG_INF;
    int bw = -1;
    for (int w = 0; w < W; ++w) {
      oracle_log_density_diag(feats + beg * D, T, D, means + (size_t)w * S * D,
                              vars + (size_t)w * S * D, gconst + (size_t)w * S, S, sum_order, logB);
      double lp;
      if (which)
        lp = oracle_forward_log(log_start + (size_t)w * S, log_trans + (size_t)w * S * S, logB, T, S, fwd);
      else
        lp = oracle_viterbi(log_start + (size_t)w * S, log_trans + (size_t)w * S * S, logB, T, S, tie,
                            st + (size_t)w * T);
      scores[u * W + w] = lp;
      if (lp > bs) {
        bs = lp;
        bw = w;
      }
    }
    if (best_word) best_word[u] = bw;
    if (path && !which) memcpy(path + beg, st + (size_t)(bw < 0 ? 0 : bw) * T, sizeof(int32_t) * (size_t)T);
    free(logB);
    free(st);
    if (fwd) free(fwd);
  }
}
