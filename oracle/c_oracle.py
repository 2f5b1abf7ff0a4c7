"""ctypes loader for oracle/libhmm_oracle.so (the C restatement; TEST INFRASTRUCTURE ONLY)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libhmm_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "libhmm_oracle.so"])
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        lib = C.CDLL(LIB)
        lib.oracle_viterbi.restype = C.c_double
        lib.oracle_forward_log.restype = C.c_double
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def pack_models(startprob, transmat, means, covars):
    """Host constants exactly as the product packer computes them (numpy logs)."""
    means = np.ascontiguousarray(means, dtype=np.float64)
    var = np.maximum(np.asarray(covars, dtype=np.float64), np.finfo(float).tiny)
    W, S, D = means.shape
    gconst = np.stack([D * np.log(2 * np.pi) + np.log(var[w]).sum(axis=-1) for w in range(W)])
    with np.errstate(divide="ignore"):
        ls = np.log(np.asarray(startprob, dtype=np.float64))
        lt = np.log(np.asarray(transmat, dtype=np.float64))
    return (means, np.ascontiguousarray(var), np.ascontiguousarray(gconst),
            np.ascontiguousarray(ls), np.ascontiguousarray(lt))


def decode_batch(feats, offsets, startprob, transmat, means, covars, tie=1, which=0, sum_order=1):
    """feats [total,D] f32, offsets [N+1] i64 → (scores [N,W], best_word [N], path [total]).
    ``sum_order``: 1 = X is the ``feat.T`` view decoder.py:59 passes (default), 0 = C-contiguous X.
    ``which``: 0 = Viterbi (GaussianHMM.decode), 1 = forward log-likelihood (GaussianHMM.score).
    Threads: OpenMP default (set OMP_NUM_THREADS before the first call to pin it)."""
    lib = load()
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    mu, var, gc, ls, lt = pack_models(startprob, transmat, means, covars)
    W, S, D = mu.shape
    N = offsets.shape[0] - 1
    scores = np.empty((N, W))
    best = np.empty(N, dtype=np.int32)
    path = np.zeros(feats.shape[0], dtype=np.int32)
    lib.oracle_decode_batch(_p(feats), _p(offsets), C.c_int64(N), C.c_int(D), _p(mu), _p(var), _p(gc),
                            _p(ls), _p(lt), C.c_int(W), C.c_int(S), C.c_int(tie), C.c_int(sum_order), C.c_int(which),
                            _p(scores), _p(best), _p(path))
    return scores, best, path


def log_density(X, means, covars, sum_order=0):
    lib = load()
    X = np.ascontiguousarray(X, dtype=np.float32)
    S = means.shape[0]
    mu, var, gc, _, _ = pack_models(np.ones((1, S)), np.ones((1, S, S)), means[None], covars[None])
    T, D = X.shape
    out = np.empty((T, S))
    lib.oracle_log_density_diag(_p(X), C.c_int(T), C.c_int(D), _p(mu), _p(var), _p(gc), C.c_int(S),
                                C.c_int(sum_order), _p(out))
    return out


def matmul_fma_chain(A, B):
    """``A @ B`` with every element one k-ascending fused-multiply-add chain from +0.0 — the order in
    which the golden build's BLAS evaluates ``custom_hmm.py:171`` (oracle/gram_oracle.c)."""
    lib = load()
    A = np.ascontiguousarray(A, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64)
    M, K = A.shape
    K2, N = B.shape
    assert K == K2
    out = np.empty((M, N))
    lib.oracle_matmul_fma_chain(_p(A), _p(B), _p(out), C.c_int(M), C.c_int(K), C.c_int(N))
    return out
