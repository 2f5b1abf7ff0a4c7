"""Seeded synthetic inputs shared by the tests and the golden-vector script.

The reference's dataset is not committed (``assignment2/.gitignore:2-3``), so every
parity test runs on MFCC-like synthetic features: left-to-right segments with
word-dependent means, a c0 offset of about -300 (``pytest_results/training_results.txt:239``)
and, in some utterances, runs of IDENTICAL leading/trailing frames such as the
``top_db`` clip produces on silence (``pytest_results/forward_backward_results.txt:14-19``)
— those create exact ties in the trellis.
"""
from __future__ import annotations

import numpy as np

VOCAB = ["heed", "hid", "head", "had", "hard", "hud", "hod", "hoard", "hood", "whod", "heard"]  # train.py:89-92


def word_prototypes(words, D, n_seg=8, seed=1234):
    rng = np.random.default_rng(seed)
    return {w: rng.normal(0.0, 20.0, (n_seg, D)) for w in words}


def synth_utterance(rng, proto, T, noise=5.0, silence=0):
    """(D,T) float32: n_seg piecewise-constant segment means + Gaussian noise; ``silence``
    identical frames at both ends."""
    n_seg, D = proto.shape
    if T > n_seg - 1:
        cuts = np.sort(rng.choice(np.arange(1, T), n_seg - 1, replace=False))
    else:  # very short utterance: one frame per segment
        cuts = np.arange(1, T)
    seg = np.searchsorted(cuts, np.arange(T), side="right")
    x = proto[seg].T + rng.normal(0.0, noise, (D, T))
    x[0] -= 300.0
    if silence:
        s = x[:, :1].copy()
        x[:, :silence] = s
        x[:, T - silence:] = s
    return x.astype(np.float32)


def synth_feature_set(words=VOCAB, n_per_word=6, D=13, tmin=40, tmax=90, seed=0):
    """dict word -> list of (D,T) float32, plus the flat list in word order (what
    ``load_mfccs`` would return for files named ``<k>_<word>.npy``)."""
    rng = np.random.default_rng(seed)
    protos = word_prototypes(words, D, seed=seed + 1234)
    by_word = {}
    for w in words:
        lst = []
        for k in range(n_per_word):
            T = int(rng.integers(tmin, tmax))
            lst.append(synth_utterance(rng, protos[w], T, silence=4 if (k % 3 == 2 and T >= 12) else 0))
        by_word[w] = lst
    flat = [f for w in words for f in by_word[w]]
    return by_word, flat


def synth_batch(n, T=101, D=13, seed=0, n_seg=8):
    """SURVEY.md §8(d) config-3 style fixed-length batch → (n, T, D) float32 frame-major."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, T, D), dtype=np.float32)
    for u in range(n):
        proto = rng.normal(0.0, 20.0, (n_seg, D))
        out[u] = synth_utterance(rng, proto, T).T
    return out


def trained_like_models(W, n_states, D, seed=7, aii=0.84):
    """W hmmlearn-shaped diag models (startprob, transmat, means, covars) with
    S = n_states+2 emitting states and the reference's bidiagonal topology
    (``hmmlearn_hmm.py:45-78``): distinct state means so paths are informative."""
    rng = np.random.default_rng(seed)
    S = n_states + 2
    sp = np.zeros((W, S))
    sp[:, 0] = 1.0
    A = np.zeros((W, S, S))
    means = np.empty((W, S, D))
    covars = np.empty((W, S, D))
    for w in range(W):
        a = np.clip(aii + rng.normal(0, 0.05, S), 0.5, 0.97)
        A[w, 0, 1] = 1.0
        for i in range(1, n_states + 1):
            A[w, i, i] = a[i]
            A[w, i, i + 1] = 1 - a[i]
        A[w, S - 1, S - 1] = 1.0
        means[w] = rng.normal(0.0, 20.0, (S, D))
        means[w, :, 0] -= 300.0
        covars[w] = rng.uniform(15.0, 60.0, (S, D))
    return sp, A, means, covars
