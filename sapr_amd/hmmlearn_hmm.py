"""Host-side mirror of ``assignment2/hmmlearn_hmm.py`` over the HIP trellis kernels.

The reference wraps ``hmmlearn.hmm.GaussianHMM`` (``hmmlearn_hmm.py:27-43``) and what it pickles
and later decodes with is that GaussianHMM object (``hmmlearn_hmm.py:106``, ``train.py:116-120``,
``decoder.py:26-27,43``).  hmmlearn is not a dependency here: :class:`GaussianHMM` below is a
picklable object with the same attribute names and ``fit / score / decode`` methods whose
arithmetic runs in ``libsapr_hip.so`` (sapr_estep_diag, sapr_forward_diag, sapr_viterbi_*).
Only the tiny M-step (hmmlearn ``base.py _do_mstep`` / ``hmm.py GaussianHMM._do_mstep``) and the
convergence monitor run on the host.

Known, documented deviation: the reference's flat-start ``means_`` / ``covars_`` are
float32 (``np.mean`` / ``np.var`` of float32 frames), so under numpy 1.26 its FIRST E-step evaluates
log-densities in float32; here the flat-start values are the same float32 numbers promoted to
float64 and every E-step is float64.
"""
from __future__ import annotations

import logging
from collections import deque
from typing import List

import numpy as np

from . import _lib
from .mfcc_extract import load_mfccs, load_mfccs_by_word  # noqa: F401  (same import surface as the reference)

logging.getLogger("matplotlib").setLevel(logging.WARNING)


class ConvergenceMonitor:
    """hmmlearn ``base.py ConvergenceMonitor``: ``history`` keeps every reported log-prob
    (``train.py:117`` and ``visualize.py:124`` read ``monitor_.history``)."""

    def __init__(self, tol, n_iter, verbose=False):
        self.tol, self.n_iter, self.verbose = tol, n_iter, verbose
        self.history = deque()
        self.iter = 0

    def _reset(self):
        self.iter = 0
        self.history.clear()

    def report(self, log_prob):
        precision = np.finfo(float).eps ** (1 / 2)
        if self.history and (log_prob - self.history[-1]) < -precision:
            logging.warning("Model is not converging.  Current: %s is not greater than %s. Delta is %s",
                            log_prob, self.history[-1], log_prob - self.history[-1])
        self.history.append(log_prob)
        self.iter += 1

    @property
    def converged(self):
        return (self.iter == self.n_iter
                or (len(self.history) >= 2 and self.history[-1] - self.history[-2] < self.tol))


def m_step(stats, startprob, transmat, params="stmc", startprob_prior=1.0, transmat_prior=1.0,
           means_prior=0.0, means_weight=0.0, covars_prior=1e-2, covars_weight=1.0, means=None, covars=None):
    """hmmlearn base.py ``_do_mstep`` + hmm.py ``GaussianHMM._do_mstep`` (diag).  Structural zeros of
    startprob / transmat stay zero; σ² = (covars_prior + obs² − 2μ·obs + μ²·post) / max(post, 1e-5)."""
    if "s" in params:
        sp = np.maximum(startprob_prior - 1 + stats["start"], 0)
        sp = np.where(startprob == 0, 0, sp)
        tot = sp.sum()
        startprob = sp / (tot if tot != 0 else 1.0)  # hmmlearn.utils.normalize: a zero sum divides by 1
    if "t" in params:
        tm = np.maximum(transmat_prior - 1 + stats["trans"], 0)
        tm = np.where(transmat == 0, 0, tm)
        rs = tm.sum(axis=1)
        rs[rs == 0] = 1
        transmat = tm / rs[:, None]
    denom = stats["post"][:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        if "m" in params:
            means = (means_weight * means_prior + stats["obs"]) / (means_weight + denom)
        if "c" in params:
            meandiff = means - means_prior
            c_n = (means_weight * meandiff ** 2 + stats["obs**2"] - 2 * means * stats["obs"] + means ** 2 * denom)
            c_d = max(covars_weight - 1, 0) + denom
            covars = (covars_prior + c_n) / np.maximum(c_d, 1e-5)
    return startprob, transmat, means, covars


def m_step_batch(rows, S, D, startprob, transmat, means, covars, params="stmc", startprob_prior=1.0,
                 transmat_prior=1.0, means_prior=0.0, means_weight=0.0, covars_prior=1e-2, covars_weight=1.0,
                 S_model=None, D_model=None):
    """:func:`m_step` for W models at once: ``rows[W, width]`` are sapr_estep_diag's statistics rows (kernel state
    count S, kernel feature width D), the model arrays carry a leading word axis.  Every operation is the per-model one
    applied along the trailing axes — the same elementwise arithmetic and the same row reductions, hence the same bits
    as W calls — in a tenth of the interpreter time.  Returns (startprob, transmat, means, covars, logprob[W])."""
    rows = np.asarray(rows, dtype=np.float64)
    W = rows.shape[0]
    m = S if S_model is None else S_model
    dm = D if D_model is None else D_model
    o = 2
    start = rows[:, o:o + S][:, :m]
    o += S
    trans = rows[:, o:o + S * S].reshape(W, S, S)[:, :m, :m]
    o += S * S
    post = rows[:, o:o + S][:, :m]
    o += S
    obs = rows[:, o:o + S * D].reshape(W, S, D)[:, :m, :dm]
    o += S * D
    obs2 = rows[:, o:o + S * D].reshape(W, S, D)[:, :m, :dm]
    logprob = rows[:, 1].copy()
    if "s" in params:
        sp = np.maximum(startprob_prior - 1 + start, 0)
        sp = np.where(startprob == 0, 0, sp)
        tot = sp.sum(axis=1)
        startprob = sp / np.where(tot != 0, tot, 1.0)[:, None]
    if "t" in params:
        tm = np.maximum(transmat_prior - 1 + trans, 0)
        tm = np.where(transmat == 0, 0, tm)
        rs = tm.sum(axis=2)
        rs[rs == 0] = 1
        transmat = tm / rs[:, :, None]
    denom = post[:, :, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        if "m" in params:
            means = (means_weight * means_prior + obs) / (means_weight + denom)
        if "c" in params:
            meandiff = means - means_prior
            c_n = (means_weight * meandiff ** 2 + obs2 - 2 * means * obs + means ** 2 * denom)
            c_d = max(covars_weight - 1, 0) + denom
            covars = (covars_prior + c_n) / np.maximum(c_d, 1e-5)
    return startprob, transmat, means, covars, logprob


def _features_f32(X) -> np.ndarray:
    """The kernels read float32 features — what ``mfcc_extract.py:15-24`` produces and every reference call site
    passes (``hmmlearn_hmm.py:80-81``, ``decoder.py:59``).  hmmlearn itself would compute with a float64 ``X`` at full
    width, so silently narrowing one would change results: values that do not survive the round trip through
    float32 are refused instead (float64 arrays holding float32 values, integers etc. pass unchanged)."""
    Xa = np.asarray(X)
    out = np.ascontiguousarray(Xa, dtype=np.float32)
    if Xa.dtype != np.float32 and Xa.size and not np.array_equal(out.astype(Xa.dtype, copy=False), Xa, equal_nan=True):
        raise ValueError(f"features of dtype {Xa.dtype} do not round-trip through float32: the HIP kernels compute "
                         "on float32 features (the reference's MFCCs are float32); cast explicitly if the loss is "
                         "intended")
    return out


class GaussianHMM:
    """hmmlearn-shaped diagonal-Gaussian HMM (every state emits) on the HIP kernels."""

    def __init__(self, n_components=1, covariance_type="diag", min_covar=1e-3, startprob_prior=1.0,
                 transmat_prior=1.0, means_prior=0, means_weight=0, covars_prior=1e-2, covars_weight=1,
                 algorithm="viterbi", random_state=None, n_iter=10, tol=1e-2, verbose=False, params="stmc",
                 init_params="stmc", implementation="log"):
        if covariance_type != "diag":
            raise ValueError("only covariance_type='diag' is implemented (hmmlearn_hmm.py:29)")
        if implementation != "log":
            raise ValueError("only implementation='log' is implemented (hmmlearn_hmm.py:32)")
        if algorithm != "viterbi":
            raise ValueError("only algorithm='viterbi' is implemented")
        self.n_components, self.covariance_type, self.min_covar = n_components, covariance_type, min_covar
        self.startprob_prior, self.transmat_prior = startprob_prior, transmat_prior
        self.means_prior, self.means_weight = means_prior, means_weight
        self.covars_prior, self.covars_weight = covars_prior, covars_weight
        self.algorithm, self.random_state, self.n_iter, self.tol, self.verbose = \
            algorithm, random_state, n_iter, tol, verbose
        self.params, self.init_params, self.implementation = params, init_params, implementation
        self.monitor_ = ConvergenceMonitor(self.tol, self.n_iter, self.verbose)
        # which back-trace tie-break GaussianHMM.decode uses (oracle/hmmlearn_oracle.py docstring)
        self.tie_break = "high"

    # hmmlearn exposes full matrices through covars_ and keeps the (S, D) array in _covars_
    @property
    def covars_(self):
        return np.array([np.diag(c) for c in self._covars_])

    @covars_.setter
    def covars_(self, covars):
        covars = np.array(covars, copy=True)
        if covars.ndim != 2 or np.any(covars <= 0):
            raise ValueError("'diag' covars must be a positive (n_components, n_features) array")
        self._covars_ = covars

    # ---- validation (hmmlearn _check) -----------------------------------------------------
    def _check(self):
        self.startprob_ = np.asarray(self.startprob_)
        self.transmat_ = np.asarray(self.transmat_)
        self.means_ = np.asarray(self.means_)
        S = self.n_components
        if len(self.startprob_) != S or not np.allclose(self.startprob_.sum(), 1.0):
            raise ValueError("startprob_ must have length n_components and sum to 1.0")
        if self.transmat_.shape != (S, S) or not np.allclose(self.transmat_.sum(axis=1), 1.0):
            raise ValueError("rows of transmat_ must sum to 1.0")
        if self.means_.shape[0] != S or self._covars_.shape != self.means_.shape:
            raise ValueError("means_ / covars_ shape mismatch")
        self.n_features = self.means_.shape[1]

    def _pack(self):
        from .trellis import DiagModelPack
        # one model, scored or decoded on its own (decode / score below): the exact kernels' operands only
        return DiagModelPack.from_params(self.startprob_[None], self.transmat_[None],
                                         np.asarray(self.means_, dtype=np.float64)[None],
                                         np.asarray(self._covars_, dtype=np.float64)[None], exact_only=True)

    @staticmethod
    def _split(X, lengths):
        X = np.asarray(X)
        if X.ndim != 2:
            raise ValueError("X must be 2-D (n_samples, n_features)")
        if lengths is None:
            lengths = [X.shape[0]]
        lengths = [int(n) for n in lengths]
        if sum(lengths) != X.shape[0]:
            raise ValueError("lengths do not sum to n_samples")
        return X, lengths

    # ---- GaussianHMM.decode (decoder.py:43) -----------------------------------------------
    def decode(self, X, lengths=None, algorithm=None):
        """Viterbi: ``(log_prob, state_sequence)``.  The order of numpy's sum inside the log-density
        depends on X's memory layout (oracle/hmmlearn_oracle.py): a C-contiguous X reduces pair-wise,
        the ``feat.T`` view decoder.py:59 passes reduces left to right — both reproduced."""
        from .trellis import FeatureBatch, viterbi_decode
        self._check()
        Xa = np.asarray(X)
        sum_order = _lib.SUM_PAIRWISE if Xa.flags.c_contiguous else _lib.SUM_TVIEW
        Xa, lengths = self._split(Xa, lengths)
        if len(lengths) > 1 and sum_order == _lib.SUM_TVIEW:
            sum_order = _lib.SUM_PAIRWISE  # row slices of a transposed view: treat as contiguous copies
        feats = _features_f32(Xa)
        import torch
        dev = _lib.require_gpu()
        batch = FeatureBatch.from_packed(torch.from_numpy(feats).to(dev), np.asarray(lengths))
        tie = _lib.TIE_HIGH if self.tie_break == "high" else _lib.TIE_LOW
        res = viterbi_decode(batch, self._pack(), tie=tie, sum_order=sum_order, word_sel=np.zeros(len(lengths)))
        log_prob = float(res.best_score.sum().item()) if len(lengths) > 1 else float(res.best_score[0].item())
        return log_prob, res.path.cpu().numpy().astype(np.int64)

    def predict(self, X, lengths=None):
        return self.decode(X, lengths)[1]

    # ---- GaussianHMM.score (hmmlearn_hmm.py:104) ------------------------------------------
    def score(self, X, lengths=None):
        from .trellis import FeatureBatch, forward_loglik
        self._check()
        Xa, lengths = self._split(X, lengths)
        import torch
        dev = _lib.require_gpu()
        feats = _features_f32(Xa)
        batch = FeatureBatch.from_packed(torch.from_numpy(feats).to(dev), np.asarray(lengths))
        ll = forward_loglik(batch, self._pack(), np.zeros(len(lengths), dtype=np.int64))
        return float(ll.sum().item())

    # ---- GaussianHMM.fit (hmmlearn_hmm.py:103) --------------------------------------------
    def fit(self, X, lengths=None):
        """Baum-Welch: per iteration one batched E-step on the GPU (all sequences at once), one
        all-reduce of the statistics when torch.distributed is initialised (each rank passes its own
        shard of sequences), M-step, ``monitor_.report`` and the convergence test."""
        fit_models([self], [self._split(X, lengths)])
        return self

    # pickling: plain attributes only (no device handles are ever stored on the object)


def fit_models(models: List[GaussianHMM], data) -> None:
    """Train several word models together: ``data[w] = (X_w, lengths_w)`` (this rank's shard).
    One E-step launch sequence covers every word's utterances; converged models stop updating."""
    import torch
    from . import dist as sdist
    from .trellis import DiagModelPack, EStep, FeatureBatch
    dev = _lib.require_gpu()
    W = len(models)
    for m in models:
        m._check()
        m.monitor_ = ConvergenceMonitor(m.tol, m.n_iter, m.verbose)
    S, D = models[0].n_components, models[0].n_features
    feats, lengths, utt_model = [], [], []
    for w, (X, ln) in enumerate(data):
        X = _features_f32(X)
        if X.shape[0]:
            feats.append(X)
        lengths += list(ln)
        utt_model += [w] * len(ln)
    packed = np.concatenate(feats, axis=0) if feats else np.zeros((0, D), np.float32)
    batch = FeatureBatch.from_packed(torch.from_numpy(packed).to(dev), np.asarray(lengths, dtype=np.int64))
    estep = EStep(batch, np.asarray(utt_model), W, S)
    active = [True] * W
    max_iter = max(m.n_iter for m in models)
    for _ in range(max_iter):
        if not any(active):
            break
        pack = DiagModelPack.from_models(models, device=dev, exact_only=True)  # never decoded with
        stats = estep.run(pack)
        sdist.allreduce_sum_(stats)
        host = stats.cpu().numpy()
        hyper = [(m.params, m.startprob_prior, m.transmat_prior, m.means_weight, m.covars_prior, m.covars_weight)
                 for m in models]
        if W > 1 and all(h == hyper[0] for h in hyper) and all(np.ndim(m.means_prior) == 0 for m in models) \
                and len({float(m.means_prior) for m in models}) == 1:
            # one vectorised M-step for the whole vocabulary (same bits as the per-model calls, a tenth of the time)
            m0 = models[0]
            new = m_step_batch(host, estep.S, estep.D,
                               np.stack([np.asarray(m.startprob_, dtype=np.float64) for m in models]),
                               np.stack([np.asarray(m.transmat_, dtype=np.float64) for m in models]),
                               np.stack([np.asarray(m.means_, dtype=np.float64) for m in models]),
                               np.stack([np.asarray(m._covars_, dtype=np.float64) for m in models]),
                               m0.params, m0.startprob_prior, m0.transmat_prior, m0.means_prior, m0.means_weight,
                               m0.covars_prior, m0.covars_weight, S_model=estep.S_model, D_model=estep.batch.D_model)
            for w, m in enumerate(models):
                if not active[w]:
                    continue
                m.startprob_, m.transmat_, m.means_, m._covars_ = new[0][w], new[1][w], new[2][w], new[3][w]
                m.monitor_.report(float(new[4][w]))
                if m.monitor_.converged:
                    active[w] = False
            continue
        for w, m in enumerate(models):
            if not active[w]:
                continue
            st = estep.split(host[w])
            m.startprob_, m.transmat_, m.means_, cov = m_step(
                st, m.startprob_, m.transmat_, m.params, m.startprob_prior, m.transmat_prior, m.means_prior,
                m.means_weight, m.covars_prior, m.covars_weight, np.asarray(m.means_, dtype=np.float64),
                np.asarray(m._covars_, dtype=np.float64))
            m._covars_ = cov
            m.monitor_.report(st["logprob"])
            if m.monitor_.converged:
                active[w] = False


class HMMLearnModel:
    """Same constructor, attributes and ``fit`` contract as the reference wrapper
    (``hmmlearn_hmm.py:11-108``): flat start from the global mean / variance of ``feature_set``,
    bidiagonal transitions with a_ii = exp(-1/(avg_frames_per_state-1)), startprob = e_0."""

    def __init__(self, num_states: int = 8, model_name: str = None, n_iter: int = 15, min_covar: float = 0.01):
        self.model_name = model_name
        self.num_states = num_states
        self.total_states = num_states + 2

        self.all_features = load_mfccs("feature_set")
        from .custom_hmm import pack_features
        packed = pack_features(self.all_features)   # one host->HBM copy for the three passes of the flat start
        self.global_mean = self.calc_global_mean(packed)
        self.global_cov = self.calc_global_cov(packed)

        self.model = GaussianHMM(n_components=self.total_states, covariance_type="diag", n_iter=n_iter,
                                 params="stmc", implementation="log", min_covar=min_covar, init_params="")
        self.model.means_ = np.tile(self.global_mean, (self.total_states, 1))
        self.model.covars_ = np.tile(self.global_cov, (self.total_states, 1))
        self.model.transmat_ = self.initialize_transmat()
        self.model.startprob_ = np.zeros(self.total_states)
        self.model.startprob_[0] = 1.0

    def initialize_transmat(self) -> np.ndarray:
        total_frames = sum(f.shape[1] for f in self.all_features)
        num_sequences = len(self.all_features)
        avg_frames = total_frames / num_sequences
        avg_frames_per_state = avg_frames / self.num_states
        aii = np.exp(-1 / (avg_frames_per_state - 1))
        aij = 1 - aii
        print("\nTransition probability initialization:")
        print(f"Total frames: {total_frames}")
        print(f"Number of sequences: {num_sequences}")
        print(f"Average frames per sequence: {avg_frames:.2f}")
        print(f"Average frames per state: {avg_frames_per_state:.2f}")
        print(f"Self-transition probability (aii): {aii:.3f}")
        print(f"Next-state transition probability (aij): {aij:.3f}")
        S = self.total_states
        transmat = np.zeros((S, S))
        transmat[0, 1] = 1.0
        for i in range(1, self.num_states + 1):
            transmat[i, i] = aii
            transmat[i, i + 1] = aij
        transmat[S - 1, S - 1] = 1.0
        return transmat

    def prepare_data(self, feature_set: List[np.ndarray]) -> np.ndarray:
        return np.concatenate([f.T for f in feature_set], axis=0)

    @staticmethod
    def _column_sums(feature_set, center=None):
        """Σ_frames x (``center`` None) or Σ_frames (x − center)² of the concatenated float32 features on the
        GPU, in numpy's own order for ``np.mean`` / ``np.var`` over axis 0 of a float32 array: one sequential
        float32 chain per coefficient (``sapr_colsum_f32``).  Returns (float32 sums [D], number of frames)."""
        import torch
        from .custom_hmm import pack_features
        pk = pack_features(feature_set)
        out = torch.empty(pk.D, dtype=torch.float32, device=pk.feats.device)
        c = None if center is None else torch.from_numpy(np.ascontiguousarray(center, dtype=np.float32)).to(pk.feats.device)
        _lib.check(_lib.load().sapr_colsum_f32(_lib.ptr(pk.feats), pk.total_frames, pk.D, _lib.ptr(c), _lib.ptr(out),
                                               _lib.current_stream()), "sapr_colsum_f32")
        return out.cpu().numpy(), pk.total_frames

    def calc_global_mean(self, feature_set: List[np.ndarray]) -> np.ndarray:
        """``np.mean(X, axis=0)`` of the float32 frames (hmmlearn_hmm.py:83-87): float32 result, bit-identical to
        numpy's (sequential float32 accumulation, then numpy's own division)."""
        sums, n = self._column_sums(feature_set)
        global_mean = np.true_divide(sums, n, out=sums, casting="unsafe")
        print(f"Global mean shape: {global_mean.shape}")
        return global_mean

    def calc_global_cov(self, feature_set: List[np.ndarray]) -> np.ndarray:
        """``np.var(X, axis=0)`` (hmmlearn_hmm.py:89-94): numpy's two-pass float32 form, mean first."""
        sums, n = self._column_sums(feature_set)
        mean = np.true_divide(sums, n, out=sums, casting="unsafe")
        sq, _ = self._column_sums(feature_set, center=mean)
        global_cov = np.true_divide(sq, n, out=sq, casting="unsafe")
        print(f"Global variance shape: {global_cov.shape}")
        print(f"Variance range: [{global_cov.min():.6f}, {global_cov.max():.6f}]")
        return global_cov

    def fit(self, feature_set: List[np.ndarray]):
        logging.info(f"Training {self.model_name} HMM using hmmlearn in {self.model.n_iter} iterations...")
        X = self.prepare_data(feature_set)
        lengths = [f.shape[1] for f in feature_set]
        try:
            self.model.fit(X, lengths)
            log_likelihood = self.model.score(X, lengths)
            return self.model, log_likelihood
        except Exception as e:  # the reference logs and returns None (hmmlearn_hmm.py:107-108)
            logging.error(f"Error occurred while training {self.model_name} HMM: {e}")
