"""Host-side mirror of ``assignment2/custom_hmm.py`` over the HIP kernels of ``custom.hip``.

Same class name, constructor, attributes (``A``, ``B["mean"]``, ``B["covariance"]``, ``pi``,
``global_mean``, ``global_covariance`` …) and method signatures as the reference ``HMM``; the
arithmetic of every method runs on the GPU, including the reference's load-bearing quirks
(Gram-row-sum emission term, non-emitting entry/exit states, per-frame renormalised xi, decode over
``features.shape[0]`` frames).  Host work is what the reference also does per call with numpy/LAPACK
on tiny per-state matrices: ``inv`` / ``slogdet`` of ``cov + 1e-6 I`` (``custom_hmm.py:160-165``),
``log(A)``, the S divisions of ``update_A`` and the symmetrise / floor step of ``update_B``.

Objects pickle as plain numpy attributes (``train.py:74-78`` → ``decoder.py:26-27``).
"""
from __future__ import annotations

import ctypes as C
import logging
from typing import List, Tuple

import os

import numpy as np

from . import _lib

_EPS = 1e-6  # custom_hmm.py:160


def _dev(a, dtype=np.float64):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(_lib.require_gpu())


def _dev_many(arrays):
    """Float64 arrays as device tensors through ONE host-to-device copy (views into one buffer): a Baum-Welch loop
    uploads its five small model arrays every iteration."""
    import torch
    parts = [np.ascontiguousarray(a, dtype=np.float64) for a in arrays]
    flat = torch.from_numpy(np.concatenate([a.ravel() for a in parts])).to(_lib.require_gpu())
    out, at = [], 0
    for a in parts:
        out.append(flat[at:at + a.size].view(a.shape))
        at += a.size
    return out


class PackedFeatures:
    """A feature list on the device in the kernels' layout: ``feats[total_frames, D]`` float32
    (frame-major), ``offsets[N+1]`` int64, host ``lens``.  Every method that takes the reference's
    ``features_list`` also takes one of these, so a caller that trains or decodes repeatedly on the same
    utterances packs (one concatenate, one host→HBM copy, one device transpose) once."""

    def __init__(self, feats, offsets, lens):
        self.feats, self.offsets, self.lens = feats, offsets, np.asarray(lens, dtype=np.int64)

    def __len__(self):
        return int(self.lens.shape[0])

    @property
    def total_frames(self):
        return int(self.lens.sum())

    @property
    def max_T(self):
        return int(self.lens.max()) if len(self) else 0

    @property
    def D(self):
        return int(self.feats.shape[1])


def pack_features(features_list) -> "PackedFeatures":
    """list of (D,T) arrays (or an already packed batch) → :class:`PackedFeatures`."""
    import torch
    if isinstance(features_list, PackedFeatures):
        return features_list
    if hasattr(features_list, "feats") and hasattr(features_list, "lengths"):  # trellis.FeatureBatch
        fb = features_list
        feats = fb.feats
        if getattr(fb, "D_model", fb.D) != fb.D:   # padded for the diagonal trellis kernels: these take any width
            feats = feats[:, :fb.D_model].contiguous()
        return PackedFeatures(feats, fb.offsets, fb.lengths)
    dev = _lib.require_gpu()
    arrs = [np.asarray(f) for f in features_list]
    lens = np.asarray([a.shape[1] for a in arrs], dtype=np.int64)
    offs = np.zeros(len(arrs) + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    if arrs:
        # one (D, total) block on the host, transposed to frame-major on the device
        dt = np.concatenate(arrs, axis=1).astype(np.float32, copy=False)
        feats = torch.from_numpy(np.ascontiguousarray(dt)).to(dev).t().contiguous()
    else:
        feats = torch.zeros((0, 1), dtype=torch.float32, device=dev)
    return PackedFeatures(feats, torch.from_numpy(offs).to(dev), lens)


def _pack_features(features_list):
    p = pack_features(features_list)
    return p.feats, p.offsets, p.lens


def model_arrays(models):
    """Per-call preparation the reference does inside compute_emission_matrix / forward:
    inv and slogdet of (cov + 1e-6 I) per emitting state, log(A)."""
    S, D = models[0].B["mean"].shape
    W = len(models)
    means = np.stack([m.B["mean"] for m in models]).astype(np.float64)
    inv = np.zeros((W, S, D, D))
    cterm = np.zeros((W, S))
    # (one stacked call each: numpy runs the same LAPACK routine matrix by matrix — the same bits as a loop over the
    # states, a third of its time)
    cov = np.stack([m.B["covariance"][1:S - 1] for m in models]).astype(np.float64) + _EPS * np.eye(D)
    inv[:, 1:S - 1] = np.linalg.inv(cov)
    cterm[:, 1:S - 1] = D * np.log(2 * np.pi) + np.linalg.slogdet(cov)[1]
    A = np.stack([m.A for m in models]).astype(np.float64)
    with np.errstate(divide="ignore"):
        logA = np.log(A)
    return means, inv, cterm, A, logA


class HMM:
    def __init__(self, num_states: int, num_obs: int, feature_set: list = None, model_name: str = None,
                 var_floor_factor: float = 0.001):
        assert num_states > 0, "Number of states must be greater than 0."
        assert num_obs > 0, "Number of observations must be greater than 0."
        self.model_name = model_name
        self.num_states = num_states
        self.num_obs = num_obs
        self.var_floor_factor = var_floor_factor
        self.total_states = num_states + 2
        self.pi = np.zeros(self.total_states)
        self.pi[0] = 1.0
        if feature_set is not None:
            dims_ok = feature_set.D == num_obs if isinstance(feature_set, PackedFeatures) else \
                all(feature.shape[0] == num_obs for feature in feature_set)
            assert dims_ok, "All features must have the same dimension as the number of observations."
            self.init_parameters(feature_set)

    # ------------------------------------------------------------------ flat start (:35-116)
    def init_parameters(self, feature_set: list) -> None:
        feature_set = pack_features(feature_set)  # one host→HBM copy for the three passes below
        self.global_mean = self.calculate_means(feature_set)
        self.global_covariance = self.calculate_covariance(feature_set, self.global_mean)
        self.global_covariance *= np.eye(self.num_obs)
        var_floor = self.var_floor_factor * np.mean(np.diag(self.global_covariance))
        np.fill_diagonal(self.global_covariance, np.maximum(np.diag(self.global_covariance), var_floor))
        self.A = self.initialize_transitions(feature_set, self.num_states)
        means = np.tile(self.global_mean, (self.total_states, 1))
        covars = np.zeros((self.total_states, self.num_obs, self.num_obs))
        for i in range(self.total_states):
            covars[i] = self.global_covariance.copy()
        self.B = {"mean": means, "covariance": covars}

    def _global_workspace(self, n_utts, total_frames, device):
        import ctypes as C
        import torch
        nb = C.c_size_t(0)
        _lib.check(_lib.load().sapr_custom_global_workspace_bytes(n_utts, total_frames, self.num_obs, C.byref(nb)),
                   "sapr_custom_global_workspace_bytes")
        return torch.empty(max(int(nb.value), 8), dtype=torch.uint8, device=device), int(nb.value)

    def calculate_means(self, feature_set: list) -> np.ndarray:
        """Global mean: per-utterance float32 row sums accumulated in float64 (``:70-80``) — on the GPU,
        in the reference's order, then (when distributed) summed over ranks."""
        from . import dist as sdist
        lib = _lib.load()
        feats, offs, lens = _pack_features(feature_set)
        out = _dev(np.zeros(self.num_obs))
        ws, nb = self._global_workspace(len(feature_set), int(lens.sum()), feats.device)
        _lib.check(lib.sapr_custom_global_sum(_lib.ptr(feats), _lib.ptr(offs), len(feature_set), self.num_obs,
                                              _lib.ptr(out), _lib.ptr(ws), nb, _lib.current_stream()),
                   "sapr_custom_global_sum")
        tot = sdist.allreduce_sum_numpy(np.r_[out.cpu().numpy(), float(lens.sum())])
        return tot[:-1] / tot[-1]

    def calculate_covariance(self, feature_set: list, mean: np.ndarray) -> np.ndarray:
        from . import dist as sdist
        lib = _lib.load()
        feats, offs, lens = _pack_features(feature_set)
        D = self.num_obs
        out = _dev(np.zeros(D * D))
        ws, nb = self._global_workspace(len(feature_set), int(lens.sum()), feats.device)
        _lib.check(lib.sapr_custom_global_cov(_lib.ptr(feats), int(lens.sum()), D, _lib.ptr(_dev(mean)),
                                              _lib.ptr(out), _lib.ptr(ws), nb, _lib.current_stream()),
                   "sapr_custom_global_cov")
        tot = sdist.allreduce_sum_numpy(np.r_[out.cpu().numpy(), float(lens.sum())])
        return tot[:-1].reshape(D, D) / tot[-1]

    def initialize_transitions(self, feature_set: list, num_states: int) -> np.ndarray:
        from . import dist as sdist
        lens = pack_features(feature_set).lens if isinstance(feature_set, PackedFeatures) else \
            np.asarray([f.shape[1] for f in feature_set])
        cnt = sdist.allreduce_sum_numpy(np.array([float(lens.sum()), float(len(feature_set))]))
        avg_frames_per_state = cnt[0] / (cnt[1] * num_states)
        aii = np.exp(-1 / (avg_frames_per_state - 1))
        total_states = num_states + 2
        A = np.zeros((total_states, total_states))
        A[0, 1] = 1.0
        for i in range(1, num_states + 1):
            A[i, i] = aii
            A[i, i + 1] = 1 - aii
        A[-1, -1] = 1.0
        return A

    # ------------------------------------------------------------------ printers (:118-144, :324-349)
    def print_parameters(self):
        print("HMM Parameters:")
        print(f"\nN (states): {self.num_states}")
        print(f"\nM (observation dim): {self.num_obs}")
        print(f"\nπ (initial state distribution): {self.pi.round(3)}")
        print("\nA (transition matrix):")
        self.print_matrix(self.A, "Transition Matrix", col="To", idx="From")

    def print_emission_parameters(self, precision: int = 3) -> None:
        import pandas as pd
        print("\nMeans (each row is a state, each column is an MFCC coefficient):")
        print(pd.DataFrame(self.B["mean"]).round(precision))
        print("\nVariances (diagonal of each state's covariance):")
        print(pd.DataFrame(np.diagonal(self.B["covariance"], axis1=1, axis2=2)).round(precision))

    def print_matrix(self, matrix: np.ndarray, title: str, col="T", idx="State", start_idx=0, start_col=0) -> None:
        if matrix.ndim == 2:
            import pandas as pd
            print(f"\n{title}:")
            print(pd.DataFrame(matrix, columns=[f"{col} {i + start_col}" for i in range(matrix.shape[1])],
                               index=[f"{idx} {i + start_idx}" for i in range(matrix.shape[0])]))
        else:
            logging.warning("Method only supports 2D matrices.")

    # ------------------------------------------------------------------ per-method API
    def _piece(self, op, T, x=None, E=None, alpha=None, beta=None, scale=None):
        import torch
        lib = _lib.load()
        S, D = self.total_states, self.num_obs
        means, inv, cterm, A, logA = model_arrays([self])
        d = [_dev(a) for a in (means, inv, cterm, A, logA)]
        z = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=d[0].device)  # noqa: E731
        tE = _dev(E) if E is not None else z(T, S)
        tal = _dev(alpha) if alpha is not None else z(T, S)
        tbe = _dev(beta) if beta is not None else z(T, S)
        tga, txi, tsc = z(T, S), z(max(T - 1, 1), S, S), z(1)
        if scale is not None:
            tsc[0] = float(scale)
        tx = _dev(np.ascontiguousarray(np.asarray(x).T), np.float32) if x is not None else None
        _lib.check(lib.sapr_custom_piece(op, _lib.ptr(tx), T, D, S, *[_lib.ptr(a) for a in d], _lib.ptr(tE),
                                         _lib.ptr(tal), _lib.ptr(tbe), _lib.ptr(tga), _lib.ptr(txi), _lib.ptr(tsc),
                                         _lib.current_stream()), "sapr_custom_piece")
        return tE, tal, tbe, tga, txi, tsc

    def compute_emission_matrix(self, features):
        """(T, total_states) log "densities"; entry/exit columns are -inf (``:146-174``).  Evaluated in the
        reference's own order (two fused-multiply-add chains + numpy's pair-wise row sum of the (T,T) Gram
        matrix, see custom.hip): bit-identical to the reference on the golden build."""
        import torch
        features = np.asarray(features)
        if features.ndim != 2 or features.shape[0] != self.num_obs:
            # the reference's broadcast in `features - mean[j, :, None]` fails for other shapes (:157)
            raise ValueError(f"operands could not be broadcast together with shapes {features.shape} "
                             f"({self.num_obs},1)")
        lib = _lib.load()
        S, D, T = self.total_states, self.num_obs, features.shape[1]
        pk = pack_features([features])
        means, inv, cterm, _, _ = model_arrays([self])
        d = [_dev(a) for a in (means, inv, cterm)]
        E = torch.full((T, S), float("-inf"), dtype=torch.float64, device=pk.feats.device)
        _lib.check(lib.sapr_custom_emission_exact(_lib.ptr(pk.feats), _lib.ptr(pk.offsets), 1, 1, D, S, 0, T,
                                                  *[_lib.ptr(a) for a in d], _lib.ptr(E), _lib.current_stream()),
                   "sapr_custom_emission_exact")
        return E.cpu().numpy()

    def forward(self, emission_matrix: np.ndarray) -> tuple:
        T = emission_matrix.shape[0]
        _, al, _, _, _, sc = self._piece(1, T, E=emission_matrix)
        return al.cpu().numpy(), sc.cpu().numpy()[0]

    def backward(self, emission_matrix: np.ndarray, scale_factor: float) -> np.ndarray:
        T = emission_matrix.shape[0]
        return self._piece(2, T, E=emission_matrix, scale=scale_factor)[2].cpu().numpy()

    def compute_gamma(self, alpha: np.ndarray, beta: np.ndarray) -> np.ndarray:
        return self._piece(3, alpha.shape[0], alpha=alpha, beta=beta)[3].cpu().numpy()

    def compute_xi(self, alpha: np.ndarray, beta: np.ndarray, emission_matrix: np.ndarray) -> np.ndarray:
        T = alpha.shape[0]
        xi = self._piece(4, T, E=emission_matrix, alpha=alpha, beta=beta)[4].cpu().numpy()
        return xi[: T - 1]

    # ------------------------------------------------------------------ M-step (:351-400)
    def update_A(self, aggregated_xi, aggregated_gamma) -> None:
        self.A[0, 1] = 1.0
        for i in range(1, self.total_states - 1):
            if aggregated_gamma[i] > 0:
                self.A[i, i] = aggregated_xi[i, i] / aggregated_gamma[i]
                self.A[i, i + 1] = 1.0 - self.A[i, i]
        self.A[-1, -1] = 1.0

    def update_B(self, features_list: list, gamma_per_seq: list) -> None:
        """Two-pass means / full covariances on the GPU from the given posteriors; symmetrise and floor
        the diagonal on the host."""
        feats, offs, lens = _pack_features(features_list)
        gamma = _dev(np.concatenate([np.asarray(g, dtype=np.float64) for g in gamma_per_seq], axis=0))
        means, covs = self._update_b_device(feats, offs, len(features_list), gamma)
        self.B["mean"], self.B["covariance"] = means, covs

    def _moments_applicable(self) -> bool:
        return (self.num_obs == 13 and self.total_states <= 16
                and os.environ.get("SAPR_CUSTOM_FOLD", "") != "ordered")

    def _moments_launch(self, feats, offs, n_utts, gamma, lane_slots, ws=None, ws_bytes=0):
        """Enqueues sapr_custom_update_b_moments; returns the device tensor [16 * 112] it fills (no synchronisation)."""
        import ctypes as C
        import torch
        lib = _lib.load()
        S, D, dev = self.total_states, self.num_obs, feats.device
        if ws is None:
            nb = C.c_size_t(0)
            _lib.check(lib.sapr_custom_update_b_workspace_bytes(n_utts, 1, D, S, C.byref(nb)),
                       "sapr_custom_update_b_workspace_bytes")
            ws, ws_bytes = torch.empty(max(int(nb.value), 8), dtype=torch.uint8, device=dev), int(nb.value)
        # (`ws` may be dropped by the caller right away: the launch is on the current stream and the caching allocator
        # hands a freed block only to later work of the same stream)
        center = torch.from_numpy(np.ascontiguousarray(self.global_mean, dtype=np.float64).reshape(-1)).to(dev)
        mom = torch.zeros(16 * 112, dtype=torch.float64, device=dev)
        _lib.check(lib.sapr_custom_update_b_moments(_lib.ptr(feats), _lib.ptr(offs), n_utts, D, S, _lib.ptr(gamma),
                                                    lane_slots, _lib.ptr(center), _lib.ptr(mom), _lib.ptr(ws), ws_bytes,
                                                    _lib.current_stream()), "sapr_custom_update_b_moments")
        return mom

    def _moments_finish(self, mom_host):
        """(means, covariances) from the (all-reduced) moments, or None where the one-pass form is ill conditioned."""
        S, D = self.total_states, self.num_obs
        m = np.asarray(mom_host, dtype=np.float64).reshape(16, 112)[:S]
        occ = m[:, 104].copy()
        means, covs = np.zeros((S, D)), np.zeros((S, D, D))
        c = np.asarray(self.global_mean, dtype=np.float64).reshape(-1)
        live = np.nonzero(occ[1:S - 1] > 0)[0] + 1          # emitting states that were visited
        if live.size:                                      # (all of them at once: the same operations per entry)
            iu = np.triu_indices(D)
            d = m[live, 91:104] / occ[live, None]
            s2 = np.zeros((live.size, D, D))
            s2[:, iu[0], iu[1]] = m[live, :91]
            s2[:, iu[1], iu[0]] = m[live, :91]
            means[live] = c + d
            covs[live] = s2 / occ[live, None, None] - d[:, :, None] * d[:, None, :]
        # S2/occ - d d^T is a difference of two terms of size ~d^2: it loses log10(d^2 / var) digits and is not
        # positive semi-definite by construction.  Where a state sits far from the centre relative to its spread
        # (or the difference came out indefinite) the reference's own two passes are run instead — for the whole
        # update, and every rank takes the same branch because the moments are the all-reduced ones
        # (SAPR_CUSTOM_FOLD=moments forces the one-pass form: tests).
        if os.environ.get("SAPR_CUSTOM_FOLD", "") == "moments" or not self._moments_ill_conditioned(m, occ, covs):
            return self._floor_covariances(means, covs, occ)
        return None

    def _update_b_device(self, feats, offs, n_utts, gamma, lane_slots=0, two_pass=False):
        """update_B on the device.  The reference is two-pass (covariances about the NEW means), so a
        sharded run needs two sums over ranks: {Σγx, Σγ} → means, then Σγ(x-μ)(x-μ)ᵀ → covariances."""
        import ctypes as C
        import torch
        from . import dist as sdist
        lib = _lib.load()
        S, D = self.total_states, self.num_obs
        dev = feats.device
        nb = C.c_size_t(0)
        _lib.check(lib.sapr_custom_update_b_workspace_bytes(n_utts, 1, D, S, C.byref(nb)),
                   "sapr_custom_update_b_workspace_bytes")
        ws = torch.empty(max(int(nb.value), 8), dtype=torch.uint8, device=dev)
        st = _lib.current_stream()
        if not two_pass and self._moments_applicable():
            # both passes from ONE read of the data: posterior-weighted moments about the global mean on the float64
            # matrix cores, one sum over ranks, then mean = c + s1/occ, cov = S2/occ - (s1/occ)(s1/occ)^T — the
            # reference's values (custom_hmm.py:366-400) to a rounding; SAPR_CUSTOM_FOLD=ordered keeps its two passes
            mom = self._moments_launch(feats, offs, n_utts, gamma, lane_slots, ws, int(nb.value))
            sdist.allreduce_sum_(mom)
            out = self._moments_finish(mom.cpu().numpy())
            if out is not None:
                return out
        # one buffer {sum_x[S][D], occ[S]} so that pass 1 is a single all-reduce
        p1 = torch.zeros(S * D + S, dtype=torch.float64, device=dev)
        means, occ = p1[:S * D], p1[S * D:]
        covs = torch.zeros(S * D * D, dtype=torch.float64, device=dev)
        _lib.check(lib.sapr_custom_update_b_sums(_lib.ptr(feats), _lib.ptr(offs), None, n_utts, 1, D, S, _lib.ptr(gamma),
                                                 lane_slots, _lib.ptr(means), _lib.ptr(occ), _lib.ptr(ws),
                                                 int(nb.value), st), "sapr_custom_update_b_sums")
        sdist.allreduce_sum_(p1)
        _lib.check(lib.sapr_custom_normalise(_lib.ptr(means), _lib.ptr(occ), S, D, st), "sapr_custom_normalise")
        _lib.check(lib.sapr_custom_update_b_scatter(_lib.ptr(feats), _lib.ptr(offs), None, n_utts, 1, D, S,
                                                    _lib.ptr(gamma), lane_slots, _lib.ptr(means), _lib.ptr(covs),
                                                    _lib.ptr(ws), int(nb.value), st), "sapr_custom_update_b_scatter")
        sdist.allreduce_sum_(covs)
        _lib.check(lib.sapr_custom_normalise(_lib.ptr(covs), _lib.ptr(occ), S, D * D, st), "sapr_custom_normalise")
        host = torch.cat([p1, covs]).cpu().numpy()  # one D2H for {means, occ, covs}
        means, occ = host[:S * D].reshape(S, D).copy(), host[S * D:S * D + S].copy()
        covs = host[S * D + S:].reshape(S, D, D).copy()
        return self._floor_covariances(means, covs, occ)

    #: the one-pass covariance is kept while max_d d_d^2 / var_d <= this (it then carries >= 12 of float64's 16 digits)
    MOMENTS_MAX_CANCELLATION = 1.0e4

    def _moments_ill_conditioned(self, m, occ, covs) -> bool:
        S, D = self.total_states, self.num_obs
        live = [j for j in range(1, S - 1) if occ[j] > 0]
        for j in live:
            d = m[j, 91:104] / occ[j]
            var = np.diagonal(covs[j])
            if np.any(var <= 0) or np.any(d * d > self.MOMENTS_MAX_CANCELLATION * var):
                return True
        if live:
            sym = (covs[live] + np.transpose(covs[live], (0, 2, 1))) / 2
            try:   # positive definite <=> the Cholesky factorisation exists (one stacked LAPACK call)
                np.linalg.cholesky(sym)
            except np.linalg.LinAlgError:
                return True
        return False

    def _floor_covariances(self, means, covs, occ):
        """custom_hmm.py:392-399: symmetrise and floor the variances of the states that were visited."""
        S, D = self.total_states, self.num_obs
        var_floor = self.var_floor_factor * np.mean(np.diagonal(self.global_covariance))
        live = np.nonzero(np.asarray(occ)[1:S - 1] > 0)[0] + 1
        if live.size:
            sym = (covs[live] + np.transpose(covs[live], (0, 2, 1))) / 2
            idx = np.arange(D)
            sym[:, idx, idx] = np.maximum(sym[:, idx, idx], var_floor)
            covs[live] = sym
        return means, covs

    # ------------------------------------------------------------------ Baum-Welch (:402-460)
    def baum_welch(self, features_list: list, max_iter: int = 15, tol: float = 1e-4):
        """EM with the reference's order: E-step over all sequences (one launch), history append,
        convergence test BEFORE the M-step, then update_A / update_B."""
        import torch
        from . import dist as sdist
        lib = _lib.load()
        print(f"\nTraining `{self.model_name}` HMM using Baum-Welch algorithm...")
        S, D = self.total_states, self.num_obs
        feats, offs, lens = _pack_features(features_list)
        N, total = len(features_list), int(lens.sum())
        z = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=feats.device)  # noqa: E731
        # lattices in the lane-contiguous layout [max_T][S][slots] (nobody outside the kernels reads them)
        slots = -(-max(N, 1) // 64) * 64
        max_T = int(lens.max()) if N else 1
        # (uninitialised: every element a kernel reads — frames t < T of live slots — is written by a kernel first;
        # zero-filling 4 x 0.8 GB per call cost 0.4 ms per iteration of a three-iteration run)
        E, al, be, ga = (torch.empty(max_T * S * slots, dtype=torch.float64, device=feats.device) for _ in range(4))
        utt_out = z(N, 2 + S + S * S)
        # the features once more in slot-major order [max_T][D][slots]: the E-step kernel's 13 values per frame and
        # state are then coalesced rows instead of 64 private reads per wavefront load (they do not change between
        # iterations: staged once per call)
        feat_t = torch.empty(max_T * D * slots, dtype=torch.float32, device=feats.device)
        # ... and every utterance's sum over its frames (the reference's Gram row sum needs it; same for every iteration)
        frame_sums = torch.empty(D * slots, dtype=torch.float64, device=feats.device)
        _lib.check(lib.sapr_custom_stage_features(_lib.ptr(feats), _lib.ptr(offs), N, D, max_T, slots, _lib.ptr(feat_t),
                                                  _lib.ptr(frame_sums), _lib.current_stream()),
                   "sapr_custom_stage_features")
        folded = z(2 + S + S * S)
        prev_log_likelihood = float("-inf")
        log_likelihood_history = []
        for iteration in range(max_iter):
            arrs = _dev_many(model_arrays([self]))
            _lib.check(lib.sapr_custom_estep_staged(_lib.ptr(feats), _lib.ptr(offs), None, N, D, S, 1,
                                                    *[_lib.ptr(a) for a in arrs], slots, _lib.ptr(E), _lib.ptr(al),
                                                    _lib.ptr(be), _lib.ptr(ga), None, _lib.ptr(utt_out),
                                                    _lib.ptr(feat_t), _lib.ptr(frame_sums), _lib.current_stream()),
                       "sapr_custom_estep_staged")
            # the reference's accumulation order over sequences (custom_hmm.py:434-439: one sequence after
            # another) as a fixed-order fold on the device: only 2 + S + S*S doubles cross PCIe per iteration
            _lib.check(lib.sapr_custom_fold_rows(_lib.ptr(utt_out), N, 2 + S + S * S, _lib.ptr(folded),
                                                 _lib.current_stream()), "sapr_custom_fold_rows")
            # (round 4) update_B's moments need the posteriors only: their kernel is enqueued behind the fold, so that
            # an iteration waits for the device ONCE — one device-to-host copy of {folded sums, moments} and, when
            # sharded, ONE sum over ranks of both — instead of twice (a converged last iteration wastes one launch)
            one_sync = N > 0 and self._moments_applicable()
            if one_sync:
                mom = self._moments_launch(feats, offs, N, ga, slots)
                both = torch.cat([folded, mom]).cpu().numpy()
                fo, mom_host = both[:2 + S + S * S], both[2 + S + S * S:]
            else:
                fo, mom_host = folded.cpu().numpy(), None
            aggregated_gamma = fo[2:2 + S].copy()
            aggregated_xi = fo[2 + S:].reshape(S, S).copy()
            total_log_likelihood = float(fo[0]) if N else 0
            if sdist.is_distributed():  # utterance shards: one sum of {LL, Σγ, Σξ} (and the moments) per EM iteration
                vec = np.r_[total_log_likelihood, aggregated_gamma, aggregated_xi.ravel()]
                tot = sdist.allreduce_sum_numpy(np.r_[vec, mom_host] if one_sync else vec)
                total_log_likelihood, aggregated_gamma = float(tot[0]), tot[1:1 + S]
                aggregated_xi = tot[1 + S:1 + S + S * S].reshape(S, S)
                if one_sync:
                    mom_host = tot[1 + S + S * S:]
            log_likelihood_history.append(total_log_likelihood)
            print(f"Iteration {iteration + 1}, Log-Likelihood: {total_log_likelihood:.2f}")
            if abs(total_log_likelihood - prev_log_likelihood) < tol:
                print(f"Converged after {iteration + 1} iterations!")
                break
            prev_log_likelihood = total_log_likelihood
            self.update_A(aggregated_xi, aggregated_gamma)
            out = self._moments_finish(mom_host) if one_sync else None
            if out is None:   # other shapes, SAPR_CUSTOM_FOLD=ordered, or an ill-conditioned one-pass covariance
                out = self._update_b_device(feats, offs, N, ga, lane_slots=slots, two_pass=one_sync)
            self.B["mean"], self.B["covariance"] = out
        print("Training complete!")
        return log_likelihood_history

    # ------------------------------------------------------------------ Viterbi (:462-514)
    def decode(self, features: np.ndarray) -> Tuple[float, List[int]]:
        """Returns ``(log_prob, path)`` like the reference (its annotation says otherwise); the trellis
        runs over ``features.shape[0]`` frames of a ``(D, T)`` array — the reference's quirk."""
        scores, paths = decode_batch([self], [features])
        return float(scores[0, 0]), [int(s) for s in paths[0][0]]


def decode_batch(models, features_list, with_best: bool = False):
    """Every utterance against every custom model in one launch sequence → ``(scores [N,W], paths [N,W,Tq])``;
    ``with_best=True`` adds Decoder.decode_sequence's arg-max over the models, evaluated on the device:
    ``(scores, paths, best_word [N] (-1 = none), best_score [N], best_path [N,Tq])``."""
    import torch
    lib = _lib.load()
    m0 = models[0]
    S, D = m0.total_states, m0.num_obs
    if isinstance(features_list, PackedFeatures) or hasattr(features_list, "feats"):
        pk = pack_features(features_list)
        if pk.D != D:
            raise ValueError(f"operands could not be broadcast together with shapes ({pk.D},T) ({D},1)")
        short = pk.lens[pk.lens < D]
    else:
        for f in features_list:
            f = np.asarray(f)
            if f.ndim != 2 or f.shape[0] != D:
                raise ValueError(f"operands could not be broadcast together with shapes {f.shape} ({D},1)")
        pk = pack_features(features_list)
        short = pk.lens[pk.lens < D]
    if short.size:
        # reference: emission_matrix[t, j] with t up to features.shape[0]-1 (:500)
        raise IndexError(f"index {int(short[0])} is out of bounds for axis 0 with size {int(short[0])}")
    N, W, Tq = len(pk), len(models), D
    dev = pk.feats.device
    arrs = _dev_many(model_arrays(models))
    e_rows = torch.empty(max(N * W * Tq * S, 1), dtype=torch.float64, device=dev)
    scores = torch.zeros((N, W), dtype=torch.float64, device=dev)
    paths = torch.zeros((N, W, Tq), dtype=torch.int32, device=dev)
    bw = torch.full((N,), -1, dtype=torch.int32, device=dev) if with_best else None
    bs = torch.full((N,), float("-inf"), dtype=torch.float64, device=dev) if with_best else None
    bp = torch.zeros((N, Tq), dtype=torch.int32, device=dev) if with_best else None
    _lib.check(lib.sapr_custom_decode(_lib.ptr(pk.feats), _lib.ptr(pk.offsets), N, W, D, S, m0.num_states, Tq,
                                      *[_lib.ptr(a) for a in arrs], _lib.ptr(e_rows), _lib.ptr(scores),
                                      _lib.ptr(paths), _lib.ptr(bw), _lib.ptr(bs), _lib.ptr(bp),
                                      _lib.current_stream()), "sapr_custom_decode")
    if with_best:
        return _lib.to_host(scores, paths, bw, bs, bp)
    return _lib.to_host(scores, paths)
