"""CPU oracle: restatement of hmmlearn 0.3.3 ``GaussianHMM`` (diag, implementation="log").

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

**PARITY UNPINNED.**  The arithmetic of this path lives in hmmlearn 0.3.3
(``/root/reference/assignment2/poetry.lock:430-431``), an un-vendored dependency
that is not installed in the build container and cannot be fetched (no network).
This file restates hmmlearn's published algorithm (``hmmlearn/stats.py``
``_log_multivariate_normal_density_diag``; ``hmmlearn/_hmmc.cpp`` ``forward_log``,
``backward_log``, ``compute_log_xi_sum``, ``viterbi``; ``hmmlearn/hmm.py``
``GaussianHMM._do_mstep``; ``hmmlearn/base.py`` ``fit``/``ConvergenceMonitor``) and is
anchored on the reference's call sites:

* construction  – ``hmmlearn_hmm.py:27-43`` (n_components = N_s+2, diag,
  params="stmc", init_params="", implementation="log", flat start);
* ``fit``/``score`` – ``hmmlearn_hmm.py:103-104``;
* ``decode``     – ``decoder.py:43`` on a ``(T, D)`` float32 view (``decoder.py:59``).

No reference test or golden vector touches this path, so the
only pins are internal: a brute-force path enumeration for Viterbi, a
textbook scaled forward for the log-likelihood, and EM monotonicity.

Unverified detail carried as a parameter: the back-trace tie-break.  The C++
``_hmmc.viterbi`` of 0.3.x is recalled to keep ``std::max`` over ``(value, index)``
pairs (equal values -> HIGHER state index; the final state uses
``std::max_element`` -> lowest index); the older Cython ``_argmax`` kept the LOWER
index.  ``tie="high"`` (default, 0.3.3 as recalled) / ``tie="low"`` select them and
the HIP kernel implements both.
"""
from __future__ import annotations

import numpy as np

NEG_INF = -np.inf
TINY = np.finfo(float).tiny


def _log(x):
    with np.errstate(divide="ignore"):
        return np.log(np.asarray(x, dtype=np.float64))


# ------------------------------------------------------------------ log-density
def log_density_diag(X, means, covars):
    """hmmlearn/stats.py ``_log_multivariate_normal_density_diag`` (0.3.x form).

    X (T,D) any float dtype; means/covars (S,D) float64 → (T,S) float64.
    ``X - means`` promotes float32 frames to float64 before anything is rounded;
    both ``sum(axis=-1)`` run numpy's pair-wise reduction over a contiguous axis
    (8 accumulators, then a sequential tail) — the HIP kernel reproduces that
    order term for term so its scores are bit-identical.
    """
    nc, nf = means.shape
    covars = np.maximum(covars, TINY)
    with np.errstate(over="ignore"):
        return -0.5 * (nf * np.log(2 * np.pi)
                       + np.log(covars).sum(axis=-1)
                       + ((X[:, None, :] - means) ** 2 / covars).sum(axis=-1))


def density_constants(means, covars):
    """Host-side per-state constant ``nf*log(2*pi) + sum(log(covars))`` exactly as
    numpy evaluates it inside :func:`log_density_diag` (used by the product path's
    parameter packer; restated here so tests can compare)."""
    nc, nf = means.shape
    covars = np.maximum(covars, TINY)
    return nf * np.log(2 * np.pi) + np.log(covars).sum(axis=-1)


# -------------------------------------------------------------------- logsumexp
def _logsumexp(v):
    """_hmmc.cpp ``logsumexp``: max-shifted, returns the max itself when it is infinite."""
    m = np.max(v)
    if np.isinf(m):
        return m
    return np.log(np.sum(np.exp(v - m))) + m  # sequential in C++; n<=32 here so np.sum order == loop order for n<8 only


def _logsumexp_seq(v):
    m = max(v)
    if np.isinf(m):
        return m
    acc = 0.0
    for x in v:
        acc += np.exp(x - m)
    return np.log(acc) + m


def _logaddexp(a, b):
    """_hmmc.cpp ``logaddexp`` (log1p form)."""
    if a == NEG_INF:
        return b
    if b == NEG_INF:
        return a
    return max(a, b) + np.log1p(np.exp(-abs(b - a)))


# ---------------------------------------------------------------- forward/backward
def forward_log(startprob, transmat, logB):
    """_hmmc.cpp ``forward_log`` → (log_prob, fwd (T,S))."""
    ls, lA = _log(startprob), _log(transmat)
    T, S = logB.shape
    fwd = np.empty((T, S))
    fwd[0] = ls + logB[0]
    for t in range(1, T):
        for j in range(S):
            fwd[t, j] = _logsumexp_seq(fwd[t - 1] + lA[:, j]) + logB[t, j]
    return _logsumexp_seq(fwd[T - 1]), fwd


def backward_log(startprob, transmat, logB):
    """_hmmc.cpp ``backward_log`` → bwd (T,S)."""
    lA = _log(transmat)
    T, S = logB.shape
    bwd = np.empty((T, S))
    bwd[T - 1] = 0.0
    for t in range(T - 2, -1, -1):
        for i in range(S):
            bwd[t, i] = _logsumexp_seq(lA[i] + logB[t + 1] + bwd[t + 1])
    return bwd


def log_xi_sum(fwd, transmat, bwd, logB):
    """_hmmc.cpp ``compute_log_xi_sum`` → (S,S) log of summed xi."""
    lA = _log(transmat)
    T, S = logB.shape
    lp = _logsumexp_seq(fwd[T - 1])
    out = np.full((S, S), NEG_INF)
    for t in range(T - 1):
        for i in range(S):
            for j in range(S):
                lx = fwd[t, i] + lA[i, j] + logB[t + 1, j] + bwd[t + 1, j] - lp
                out[i, j] = _logaddexp(out[i, j], lx)
    return out


def posteriors(fwd, bwd):
    """base.py ``_compute_posteriors_log`` — row soft-max of fwd+bwd."""
    lg = fwd + bwd
    m = np.max(lg, axis=1, keepdims=True)
    with np.errstate(under="ignore", invalid="ignore"):
        lse = np.log(np.sum(np.exp(lg - m), axis=1, keepdims=True)) + m
        return np.exp(lg - lse)


# ------------------------------------------------------------------------ Viterbi
def viterbi(startprob, transmat, logB, tie="high"):
    """_hmmc.cpp ``viterbi`` → (log_prob, states int64 (T,)).

    Lattice: ``d[t,i] = max_j(d[t-1,j] + lA[j,i]) + logB[t,i]``.  Final state:
    first maximum of the last row.  Back-trace: maximum of ``d[t,i] + lA[i,prev]``
    with ties resolved per ``tie`` (module docstring).
    """
    ls, lA = _log(startprob), _log(transmat)
    T, S = logB.shape
    d = np.empty((T, S))
    d[0] = ls + logB[0]
    for t in range(1, T):
        for i in range(S):
            d[t, i] = np.max(d[t - 1] + lA[:, i]) + logB[t, i]
    states = np.empty(T, dtype=np.int64)
    prev = int(np.argmax(d[T - 1]))  # std::max_element → first max
    states[T - 1] = prev
    lp = float(d[T - 1, prev])
    for t in range(T - 2, -1, -1):
        v = d[t] + lA[:, prev]
        if tie == "high":
            best, arg = NEG_INF, 0
            for i in range(S):
                # std::max(pair, pair): replace when (best,arg) < (v[i], i)
                if best < v[i] or (best == v[i] and arg < i):
                    best, arg = v[i], i
            prev = arg
        else:
            prev = int(np.argmax(v))
        states[t] = prev
    return lp, states


def decode(X, startprob, transmat, means, covars, tie="high"):
    """``GaussianHMM.decode(X)`` as called at ``decoder.py:43`` → (log_prob, states)."""
    return viterbi(startprob, transmat, log_density_diag(X, means, covars), tie=tie)


def score(X, lengths, startprob, transmat, means, covars):
    """``GaussianHMM.score(X, lengths)`` (``hmmlearn_hmm.py:104``) — sum of forward log-probs."""
    tot, o = 0.0, 0
    for n in lengths:
        lp, _ = forward_log(startprob, transmat, log_density_diag(X[o:o + n], means, covars))
        tot += lp
        o += n
    return tot


# ------------------------------------------------------------------------- EM fit
def new_stats(S, D):
    return {"nobs": 0, "start": np.zeros(S), "trans": np.zeros((S, S)),
            "post": np.zeros(S), "obs": np.zeros((S, D)), "obs2": np.zeros((S, D))}


def accumulate(stats, X, startprob, transmat, means, covars):
    """One sequence of base.py ``_do_estep`` + hmm.py ``_accumulate_sufficient_statistics``.

    ``X**2`` is evaluated in X's own dtype (float32 for the reference's feature
    arrays — ``hmmlearn_hmm.py:80-81`` concatenates float32) and only then
    promoted by the matmul, exactly as ``posteriors.T @ X**2`` does in numpy.
    Returns the sequence log-prob.
    """
    logB = log_density_diag(X, means, covars)
    lp, fwd = forward_log(startprob, transmat, logB)
    bwd = backward_log(startprob, transmat, logB)
    post = posteriors(fwd, bwd)
    stats["nobs"] += 1
    stats["start"] += post[0]
    if X.shape[0] > 1:
        with np.errstate(under="ignore"):
            stats["trans"] += np.exp(log_xi_sum(fwd, transmat, bwd, logB))
    stats["post"] += post.sum(axis=0)
    stats["obs"] += post.T @ X
    stats["obs2"] += post.T @ (X ** 2)
    return lp


def m_step(stats, startprob, transmat, covars_prior=1e-2, covars_weight=1.0,
           means_prior=0.0, means_weight=0.0, startprob_prior=1.0, transmat_prior=1.0):
    """base.py ``_do_mstep`` + hmm.py ``GaussianHMM._do_mstep`` (diag) → new (startprob, transmat, means, covars)."""
    sp = np.maximum(startprob_prior - 1 + stats["start"], 0)
    sp = np.where(startprob == 0, 0, sp)
    tot = sp.sum()
    sp = sp / (tot if tot != 0 else 1.0)   # hmmlearn.utils.normalize: a zero sum is divided by 1, not by 0
    tm = np.maximum(transmat_prior - 1 + stats["trans"], 0)
    tm = np.where(transmat == 0, 0, tm)
    rs = tm.sum(axis=1)
    rs[rs == 0] = 1
    tm = tm / rs[:, None]
    denom = stats["post"][:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        means = (means_weight * means_prior + stats["obs"]) / (means_weight + denom)
        meandiff = means - means_prior
        c_n = (means_weight * meandiff ** 2 + stats["obs2"]
               - 2 * means * stats["obs"] + means ** 2 * denom)
        c_d = max(covars_weight - 1, 0) + denom
        covars = (covars_prior + c_n) / np.maximum(c_d, 1e-5)
    return sp, tm, means, covars


def fit(X, lengths, startprob, transmat, means, covars, n_iter=15, tol=1e-2):
    """base.py ``fit``: E-step, M-step, then ``monitor_.report`` and the convergence
    test (M-step runs BEFORE the test).  Returns (startprob, transmat, means, covars, history)."""
    hist = []
    S, D = means.shape
    for it in range(n_iter):
        st = new_stats(S, D)
        cur, o = 0.0, 0
        for n in lengths:
            cur += accumulate(st, X[o:o + n], startprob, transmat, means, covars)
            o += n
        startprob, transmat, means, covars = m_step(st, startprob, transmat)
        hist.append(cur)
        if it + 1 == n_iter or (len(hist) >= 2 and hist[-1] - hist[-2] < tol):
            break
    return startprob, transmat, means, covars, hist


# ------------------------------------------------------------- flat start (wrapper)
def flat_start(feature_set, num_states):
    """``hmmlearn_hmm.py:38-78,83-94`` — global mean/variance over concatenated frames,
    bidiagonal transmat with a_ii = exp(-1/(avg_frames_per_state-1)), startprob e_0."""
    X = np.concatenate([f.T for f in feature_set], axis=0)
    gm = np.mean(X, axis=0)
    gv = np.var(X, axis=0)
    S = num_states + 2
    total = sum(f.shape[1] for f in feature_set)
    avg = total / len(feature_set) / num_states
    aii = np.exp(-1 / (avg - 1))
    A = np.zeros((S, S))
    A[0, 1] = 1.0
    for i in range(1, num_states + 1):
        A[i, i] = aii
        A[i, i + 1] = 1 - aii
    A[S - 1, S - 1] = 1.0
    sp = np.zeros(S)
    sp[0] = 1.0
    return sp, A, np.tile(gm, (S, 1)), np.tile(gv, (S, 1))
