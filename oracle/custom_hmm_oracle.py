"""CPU oracle: numpy restatement of the reference's from-scratch HMM.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PINNED against the
imported reference (``tests/golden/make_golden.py``) and its known answers.

Every function cites the span of ``/root/reference/assignment2/custom_hmm.py``
whose arithmetic it restates.  State layout: index 0 = non-emitting entry,
1..N_s emitting, S-1 = non-emitting exit (``custom_hmm.py:24``).  Features are
``(D, T)`` channel-first (``mfcc_extract.py:15-24``).

The restatement keeps every load-bearing quirk of the reference:

* the "Mahalanobis" term is the ROW SUM of a (T,T) Gram matrix
  (``custom_hmm.py:168-172``), i.e. ``d_t^T C^-1 (sum_s d_s)``;
* ``decode`` runs the trellis over ``features.shape[0]`` (= D) frames only
  (``custom_hmm.py:466``);
* ``xi`` is renormalised per frame and its exit column is always 0
  (``custom_hmm.py:301-320``);
* the forward pass subtracts one global ``max(alpha)`` (``custom_hmm.py:208-209``).
"""
from __future__ import annotations

import numpy as np

NEG_INF = -np.inf
_EPS_REG = 1e-6  # custom_hmm.py:160


# --------------------------------------------------------------------------- init
def global_mean(feature_set):
    """custom_hmm.py:70-80 — per-utterance row sums accumulated in list order."""
    acc = np.zeros(feature_set[0].shape[0])
    n = 0
    for f in feature_set:
        acc += np.sum(f, axis=1)
        n += f.shape[1]
    return acc / n


def global_covariance(feature_set, mean, var_floor_factor=0.001):
    """custom_hmm.py:82-92 then :42-49 — biased full covariance, off-diagonals
    zeroed, diagonal floored at ``var_floor_factor * mean(diag)``."""
    d = mean.shape[0]
    cov = np.zeros((d, d))
    n = 0
    for f in feature_set:
        c = f - mean[:, None]
        cov += c @ c.T
        n += f.shape[1]
    cov = cov / n
    cov *= np.eye(d)
    floor = var_floor_factor * np.mean(np.diag(cov))
    np.fill_diagonal(cov, np.maximum(np.diag(cov), floor))
    return cov


def flat_start_transitions(feature_set, num_states):
    """custom_hmm.py:94-116 — a_ii = exp(-1/(avg_frames_per_state-1))."""
    total = sum(f.shape[1] for f in feature_set)
    avg = total / (len(feature_set) * num_states)
    aii = np.exp(-1 / (avg - 1))
    S = num_states + 2
    A = np.zeros((S, S))
    A[0, 1] = 1.0
    for i in range(1, num_states + 1):
        A[i, i] = aii
        A[i, i + 1] = 1 - aii
    A[-1, -1] = 1.0
    return A


def flat_start(feature_set, num_states, var_floor_factor=0.001):
    """custom_hmm.py:35-68 — returns dict(global_mean, global_covariance, A, mean, covariance)."""
    gm = global_mean(feature_set)
    gc = global_covariance(feature_set, gm, var_floor_factor)
    S = num_states + 2
    return {
        "global_mean": gm,
        "global_covariance": gc,
        "A": flat_start_transitions(feature_set, num_states),
        "mean": np.tile(gm, (S, 1)),
        "covariance": np.repeat(gc[None], S, axis=0).copy(),
    }


# ----------------------------------------------------------------------- emission
def emission(features, means, covs, gram="blas"):
    """custom_hmm.py:146-174 — (T,S) log "densities"; columns 0 and S-1 stay -inf.

    Follows the reference literally (T x T Gram, row sum) so that the oracle is a restatement, not a
    simplification.  ``gram="blas"`` evaluates the two products with numpy's matmul like the
    reference, so its last bits are those of the BLAS of the machine it runs on; ``gram="chain"``
    states the order the golden build's BLAS uses (one k-ascending fused-multiply-add chain per
    element, oracle/gram_oracle.c) and is machine-independent.  The HIP decode and
    ``compute_emission_matrix`` implement the chain order and are compared bit for bit; the batched
    E-step uses the algebraically equal ``d_t . (C^-1 sum_s d_s)`` form and is compared with a
    tolerance.
    """
    D, T = features.shape
    S = means.shape[0]
    out = np.full((T, S), NEG_INF)
    for j in range(1, S - 1):
        diff = features - means[j, :, None]
        cov = covs[j] + _EPS_REG * np.eye(D)
        inv = np.linalg.inv(cov)
        _, logdet = np.linalg.slogdet(cov)
        if gram == "chain":
            from . import c_oracle
            G = c_oracle.matmul_fma_chain(c_oracle.matmul_fma_chain(diff.T, inv), diff)
        else:
            G = diff.T @ inv @ diff
        out[:, j] = -0.5 * (D * np.log(2 * np.pi) + logdet + np.sum(G, axis=1))
    return out


def emission_rowsum_form(features, means, covs):
    """Algebraically equal O(T*D) form of :func:`emission` (what the HIP kernel computes)."""
    D, T = features.shape
    S = means.shape[0]
    out = np.full((T, S), NEG_INF)
    for j in range(1, S - 1):
        diff = features.astype(np.float64) - means[j, :, None]
        cov = covs[j] + _EPS_REG * np.eye(D)
        inv = np.linalg.inv(cov)
        _, logdet = np.linalg.slogdet(cov)
        v = inv @ diff.sum(axis=1)
        out[:, j] = -0.5 * (D * np.log(2 * np.pi) + logdet + diff.T @ v)
    return out


# ------------------------------------------------------------------ forward/backward
def _log(x):
    with np.errstate(divide="ignore"):
        return np.log(x)


def forward(E, A):
    """custom_hmm.py:176-211 — returns (alpha - scale, scale)."""
    T, S = E.shape
    lgA = _log(A)
    al = np.full((T, S), NEG_INF)
    al[0, 0] = 0
    al[0, 1] = lgA[0, 1] + E[0, 1]
    diag = np.diagonal(lgA)          # lg A[j, j]
    sup = np.diagonal(lgA, offset=1)  # lg A[j, j+1]
    for t in range(1, T):
        p = al[t - 1]
        with np.errstate(invalid="ignore"):
            # j = 1 .. S-2 : from j-1 (entry for j = 1) and self
            al[t, 1:S - 1] = np.logaddexp(p[0:S - 2] + sup[0:S - 2],
                                          p[1:S - 1] + diag[1:S - 1]) + E[t, 1:S - 1]
        al[t, S - 1] = p[S - 2] + sup[S - 2]
    scale = np.max(al)
    return al - scale, scale


def backward(E, A, scale):
    """custom_hmm.py:213-246."""
    T, S = E.shape
    lgA = _log(A)
    be = np.full((T, S), NEG_INF)
    be[-1, -1] = 0
    diag = np.diagonal(lgA)
    sup = np.diagonal(lgA, offset=1)
    for t in range(T - 2, -1, -1):
        n, e = be[t + 1], E[t + 1]
        be[t, 0] = sup[0] + e[1] + n[1]
        with np.errstate(invalid="ignore"):
            if S > 3:
                i = np.arange(1, S - 2)
                be[t, i] = np.logaddexp(diag[i] + e[i] + n[i], sup[i] + e[i + 1] + n[i + 1])
            i = S - 2
            be[t, i] = np.logaddexp(diag[i] + e[i] + n[i], sup[i] + n[i + 1])
    be[:-1] -= scale
    return be


def gamma(alpha, beta):
    """custom_hmm.py:248-257 — row soft-max of alpha+beta."""
    with np.errstate(invalid="ignore"):
        lg = alpha + beta
        return np.exp(lg - np.logaddexp.reduce(lg, axis=1, keepdims=True))


def xi(alpha, beta, E, A):
    """custom_hmm.py:259-322 — (T-1,S,S), structurally allowed cells only, per-t renormalised."""
    T, S = alpha.shape
    lgA = _log(A)
    out = np.zeros((T - 1, S, S))
    ll = np.logaddexp.reduce(alpha[-1])
    with np.errstate(invalid="ignore", over="ignore"):
        for t in range(T - 1):
            a, e, b = alpha[t], E[t + 1], beta[t + 1]
            out[t, 0, 1] = np.exp(a[0] + lgA[0, 1] + e[1] + b[1] - ll)
            for i in range(1, S - 1):
                if A[i, i] > 0:
                    out[t, i, i] = np.exp(a[i] + lgA[i, i] + e[i] + b[i] - ll)
                if i < S - 2:
                    out[t, i, i + 1] = np.exp(a[i] + lgA[i, i + 1] + e[i + 1] + b[i + 1] - ll)
            out[t, -2, -1] = np.exp(a[-2] + lgA[-2, -1] + e[-1] + b[-1] - ll)
            out[t, -1, -1] = np.exp(a[-1] + lgA[-1, -1] + e[-1] + b[-1] - ll)
            s = np.sum(out[t])
            if s > 0:
                out[t] /= s
    return out


# ------------------------------------------------------------------------- M-step
def update_A(A, agg_xi, agg_gamma):
    """custom_hmm.py:351-364 — in place."""
    S = A.shape[0]
    A[0, 1] = 1.0
    for i in range(1, S - 1):
        if agg_gamma[i] > 0:
            A[i, i] = agg_xi[i, i] / agg_gamma[i]
            A[i, i + 1] = 1.0 - A[i, i]
    A[-1, -1] = 1.0
    return A


def update_B(features_list, gammas, global_cov, var_floor_factor):
    """custom_hmm.py:366-400 — two-pass (means over all t, then full covariance
    about the NEW means), symmetrised, diagonal floored; entry/exit rows zero."""
    D = features_list[0].shape[0]
    S = gammas[0].shape[1]
    means = np.zeros((S, D))
    covs = np.zeros((S, D, D))
    occ = np.zeros(S)
    for f, g in zip(features_list, gammas):
        for j in range(1, S - 1):
            means[j] += np.sum(g[:, j:j + 1] * f.T, axis=0)
            occ[j] += np.sum(g[:, j])
    for j in range(1, S - 1):
        if occ[j] > 0:
            means[j] /= occ[j]
    for f, g in zip(features_list, gammas):
        for j in range(1, S - 1):
            d = f.T - means[j]
            # sum_t g[t,j] * outer(d_t, d_t), accumulated frame by frame like :385-386
            for t in range(f.shape[1]):
                covs[j] += g[t, j] * np.outer(d[t], d[t])
    floor = var_floor_factor * np.mean(np.diagonal(global_cov))
    for j in range(1, S - 1):
        if occ[j] > 0:
            covs[j] /= occ[j]
            covs[j] = (covs[j] + covs[j].T) / 2
            idx = np.diag_indices(D)
            covs[j][idx] = np.maximum(covs[j][idx], floor)
    return means, covs


def e_step(features, A, means, covs):
    """One utterance of custom_hmm.py:424-439 → (gamma, xi, seq_ll)."""
    E = emission(features, means, covs)
    al, sc = forward(E, A)
    be = backward(E, A, sc)
    g = gamma(al, be)
    x = xi(al, be, E, A)
    with np.errstate(invalid="ignore"):
        ll = np.logaddexp.reduce(al[-1])
    return g, x, ll


def baum_welch(features_list, A, means, covs, global_cov, var_floor_factor=0.001,
               max_iter=15, tol=1e-4):
    """custom_hmm.py:402-460 — returns (history, A, means, covs); convergence is
    tested BEFORE the M-step (:449-453)."""
    A = A.copy()
    hist = []
    prev = float("-inf")
    for _ in range(max_iter):
        S = A.shape[0]
        agg_g = np.zeros(S)
        agg_x = np.zeros((S, S))
        gammas = []
        total = 0
        for f in features_list:
            g, x, ll = e_step(f, A, means, covs)
            gammas.append(g)
            agg_g += np.sum(g[:-1], axis=0)
            agg_x += np.sum(x, axis=0)
            total += ll
        hist.append(total)
        with np.errstate(invalid="ignore"):
            if abs(total - prev) < tol:
                break
        prev = total
        update_A(A, agg_x, agg_g)
        means, covs = update_B(features_list, gammas, global_cov, var_floor_factor)
    return hist, A, means, covs


# ------------------------------------------------------------------------ Viterbi
def decode(features, A, means, covs, num_states, gram="blas"):
    """custom_hmm.py:462-514 — returns (log_prob, path) with the T:=features.shape[0] quirk.

    Strict ``>`` from -inf: ties keep the first listed predecessor, an
    all -inf cell stays untouched with back-pointer 0 (:489-503).
    """
    Tq = features.shape[0]
    E = emission(features, means, covs, gram=gram)
    S = means.shape[0]
    lgA = _log(A)
    V = np.full((Tq, S), NEG_INF)
    bp = np.zeros((Tq, S), dtype=int)
    V[0, 0] = 0
    V[0, 1] = lgA[0, 1] + E[0, 1]
    for t in range(1, Tq):
        for j in range(1, S):
            if j == 1:
                cand = [1, 0] if t == 1 else [1]
            elif j == S - 1:
                if t < num_states:
                    continue
                cand = [j - 1, j]
            else:
                cand = [j - 1, j]
            best, arg = NEG_INF, None
            for i in cand:
                sc = V[t - 1, i] + lgA[i, j]
                if sc > best:
                    best, arg = sc, i
            if arg is not None:
                V[t, j] = best + E[t, j] if j != S - 1 else best
                bp[t, j] = arg
    path = []
    cur = S - 1
    for t in range(Tq - 1, -1, -1):
        path.append(int(cur))
        cur = bp[t, cur]
    path.reverse()
    return float(V[Tq - 1, -1]), path
