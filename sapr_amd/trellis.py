"""Device-side containers and launches for the HMM trellis kernels.

* :class:`FeatureBatch` – a ragged batch of utterances in HBM: frame-major
  ``feats[total_frames, D]`` float32 + ``offsets[N+1]`` int64 + a length-sorted
  ``order`` (so the 64 lanes of a wavefront walk trellises of similar length).
  Built from the reference's per-utterance ``(D, T)`` numpy arrays
  (``mfcc_extract.py:15-24``) or directly from the MFCC kernel's output.
* :class:`DiagModelPack` – W diagonal-Gaussian word models (float64) laid out for
  the kernels, with every host-side constant evaluated by numpy exactly as hmmlearn
  evaluates it (``np.log(transmat)``, ``nf*log(2*pi) + log(covars).sum(-1)``).
* :func:`viterbi_decode` – two launches (scores + back-trace) through the C ABI.

PyTorch is used for device memory and streams only.
"""
from __future__ import annotations

import os

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib

_TINY = np.finfo(float).tiny


def _torch():
    import torch
    return torch


@dataclass
class FeatureBatch:
    feats: "object"      # torch float32 [total_frames, D] (device)
    offsets: "object"    # torch int64 [N+1] (device)
    order: "object"      # torch int32 [N] (device), utterances sorted by length (desc)
    lengths: np.ndarray  # host int64 [N]
    max_T: int
    D: int               # columns of `feats` = the feature width the kernels run at (kernel_dims)
    D_model: int = 0     # the caller's feature width (D >= D_model: zero columns appended, see kernel_dims)

    def __post_init__(self):
        if not self.D_model:
            self.D_model = self.D

    @property
    def n_utts(self) -> int:
        return int(self.lengths.shape[0])

    @property
    def total_frames(self) -> int:
        return int(self.lengths.sum())

    @staticmethod
    def from_arrays(utterances, layout: str = "DT", device=None) -> "FeatureBatch":
        """Pack a list of numpy arrays: ``layout="DT"`` = the reference's channel-first
        ``(D, T)`` (custom path), ``"TD"`` = frame-major ``(T, D)`` (what decoder.py:59
        hands to hmmlearn)."""
        torch = _torch()
        device = device or _lib.require_gpu()
        if len(utterances) == 0:
            raise ValueError("empty utterance list")
        if layout == "DT":
            mats = [np.ascontiguousarray(np.asarray(f).T, dtype=np.float32) for f in utterances]
        elif layout == "TD":
            mats = [np.ascontiguousarray(np.asarray(f), dtype=np.float32) for f in utterances]
        else:
            raise ValueError(f"unknown layout {layout!r}")
        D = mats[0].shape[1]
        for m in mats:
            if m.ndim != 2 or m.shape[1] != D:
                raise ValueError("all utterances must share the feature dimension")
        lengths = np.asarray([m.shape[0] for m in mats], dtype=np.int64)
        packed = np.concatenate(mats, axis=0) if lengths.sum() else np.zeros((0, D), np.float32)
        return FeatureBatch.from_packed(torch.from_numpy(packed).to(device), lengths)

    @staticmethod
    def from_packed(feats, lengths) -> "FeatureBatch":
        """``feats`` already on the device as [total_frames, D] float32; ``lengths`` host ints."""
        torch = _torch()
        lengths = np.asarray(lengths, dtype=np.int64)
        offs = np.zeros(lengths.shape[0] + 1, dtype=np.int64)
        np.cumsum(lengths, out=offs[1:])
        if feats.dtype != torch.float32 or feats.dim() != 2 or not feats.is_contiguous():
            raise ValueError("feats must be a contiguous float32 [total_frames, D] tensor")
        if feats.shape[0] != offs[-1]:
            raise ValueError("feats rows do not match sum(lengths)")
        order = np.argsort(-lengths, kind="stable").astype(np.int32)
        dev = feats.device
        d_model = int(feats.shape[1])
        dk = kernel_dims(d_model)
        if dk != d_model and d_model > 0:   # zero columns up to the next instantiated width (one device copy)
            wide = torch.zeros((feats.shape[0], dk), dtype=torch.float32, device=dev)
            wide[:, :d_model] = feats
            feats = wide
        return FeatureBatch(feats=feats, offsets=torch.from_numpy(offs).to(dev),
                            order=torch.from_numpy(order).to(dev), lengths=lengths,
                            max_T=int(lengths.max()) if lengths.size else 0, D=int(feats.shape[1]), D_model=d_model)


def is_bidiagonal(transmat: np.ndarray) -> bool:
    """True when only A[i,i] and A[i,i+1] are non-zero (hmmlearn_hmm.py:45-78 topology;
    hmmlearn's M-step keeps structural zeros, so trained models stay bidiagonal)."""
    S = transmat.shape[-1]
    mask = np.eye(S, dtype=bool) | np.eye(S, k=1, dtype=bool)
    return bool(np.all(transmat[..., ~mask] == 0))


KERNEL_STATES = (10, 18)  # state counts the trellis kernels are instantiated for (8 / 16 emitting + entry/exit)


def kernel_states(S: int) -> int:
    """Smallest instantiated state count that holds an S-state model.  Smaller models are padded with
    unreachable dummy states (start probability 0, no transition into them, self-loop 1, unit
    Gaussians): every candidate through a dummy state is -inf, exp(-inf) terms add exact zeros to the
    log-sum-exps, so scores, paths, log-likelihoods and sufficient statistics of the real states keep
    their bits.  Larger S is passed through unchanged and the C library refuses it."""
    for k in KERNEL_STATES:
        if S <= k:
            return k
    return S


KERNEL_DIMS = (13, 39)  # feature widths the trellis kernels are instantiated for (MFCC, MFCC + delta + delta-delta)


def kernel_dims(D: int) -> int:
    """Smallest instantiated feature width that holds D-dimensional features.  Narrower models run padded: the extra
    dimensions carry mean 0, variance 1 and zero features, so each adds (0 - 0)^2 / 1 = +0.0 to the quadratic form
    and log(1) = 0 to the log-determinant — the Gaussian constant is evaluated on the caller's D first.  In the
    left-to-right summation order (the ``feat.T`` view decoder.py:59 passes: SUM_TVIEW, utterances of two frames or
    more) and for D < 8, where numpy's pair-wise order IS left to right (SUM_SEQ), every score keeps its bits; where
    numpy reduces 8 <= D terms pair-wise — a C-contiguous array, or a one-frame utterance in either layout — the padded
    sum associates differently from numpy's (an ulp-level difference: paths can differ only at exact near-ties).
    Wider D is passed through unchanged and the C library refuses it."""
    for k in KERNEL_DIMS:
        if D <= k:
            return k
    return D


@dataclass
class DiagModelPack:
    means: "object"      # [W,S,D] f64
    vars: "object"       # [W,S,D] f64  (covars floored at float64 tiny, hmmlearn stats.py)
    gconst: "object"     # [W,S]   f64  nf*log(2*pi) + sum(log(vars))
    log_start: "object"  # [W,S]   f64
    log_trans: "object"  # [W,S,S] f64
    W: int
    S: int
    D: int
    topology: int
    blob: "object" = None     # device uint8: sapr_diag_pack output ({mean, var, 1/var hi, 1/var lo} interleaved ...)
    fast_div: int = 0         # 1 = parameters inside the proven domain of the FMA division
    flags: int = 0            # sapr_diag_pack's bit mask (PACK_FAST_DIV | PACK_BOUND_OK)
    S_model: int = 0          # states of the caller's models (S >= S_model: padding, see kernel_states)
    D_model: int = 0          # feature width of the caller's models (D >= D_model: padding, see kernel_dims)

    def _build_blob(self, exact_only=False):
        torch = _torch()
        lib = _lib.load()
        n = C.c_size_t(0)
        _lib.check(lib.sapr_diag_pack_bytes(self.W, self.S, self.D, C.byref(n)), "sapr_diag_pack_bytes")
        self.blob = torch.empty(int(n.value), dtype=torch.uint8, device=self.means.device)
        ok = C.c_int32(_lib.PACK_EXACT_ONLY if exact_only else 0)
        _lib.check(lib.sapr_diag_pack(_lib.ptr(self.means), _lib.ptr(self.vars), _lib.ptr(self.gconst),
                                      _lib.ptr(self.log_start), _lib.ptr(self.log_trans), self.W, self.S,
                                      self.D, _lib.ptr(self.blob), int(n.value), C.byref(ok),
                                      _lib.current_stream()), "sapr_diag_pack")
        self.flags = int(ok.value)
        self.fast_div = 1 if self.flags & _lib.PACK_FAST_DIV else 0
        return self

    @property
    def prunable(self) -> bool:
        """True when ``sapr_viterbi_decode_pruned`` accepts this pack (bidiagonal, inside the bound's domain)."""
        need = _lib.PACK_BOUND_OK | _lib.PACK_BIDIAG
        return self.topology == _lib.TOPO_BIDIAG and (self.flags & need) == need

    @staticmethod
    def from_params(startprob, transmat, means, covars, device=None, exact_only=False) -> "DiagModelPack":
        """Arrays with a leading word axis: startprob [W,S], transmat [W,S,S], means/covars [W,S,D].
        ``exact_only``: a pack for the E-step / scoring only (no operands for the pruned decoder's bounding pass:
        ``prunable`` is False) — what a Baum-Welch loop builds once per iteration."""
        torch = _torch()
        device = device or _lib.require_gpu()
        startprob = np.asarray(startprob, dtype=np.float64)
        transmat = np.asarray(transmat, dtype=np.float64)
        means = np.ascontiguousarray(means, dtype=np.float64)
        covars = np.asarray(covars, dtype=np.float64)
        W, S, D = means.shape
        if startprob.shape != (W, S) or transmat.shape != (W, S, S) or covars.shape != (W, S, D):
            raise ValueError("inconsistent model shapes")
        S_model, Sk = S, kernel_states(S)
        if Sk != S:  # pad with unreachable states (kernel_states)
            pad = Sk - S
            startprob = np.concatenate([startprob, np.zeros((W, pad))], axis=1)
            tm = np.zeros((W, Sk, Sk))
            tm[:, :S, :S] = transmat
            tm[:, np.arange(S, Sk), np.arange(S, Sk)] = 1.0
            transmat = tm
            means = np.ascontiguousarray(np.concatenate([means, np.zeros((W, pad, D))], axis=1))
            covars = np.concatenate([covars, np.ones((W, pad, D))], axis=1)
            S = Sk
        var = np.maximum(covars, _TINY)
        # evaluated per model exactly like hmmlearn: scalar + (S,) array — on the caller's D, before any padding
        # (one call over the word axis: the same row reductions along the contiguous last axis, the same bits as W calls)
        gconst = D * np.log(2 * np.pi) + np.log(var).sum(axis=-1)
        D_model, Dk = D, kernel_dims(D)
        if Dk != D:  # pad with (mean 0, variance 1) dimensions (kernel_dims)
            means = np.ascontiguousarray(np.concatenate([means, np.zeros((W, S, Dk - D))], axis=2))
            var = np.concatenate([var, np.ones((W, S, Dk - D))], axis=2)
            D = Dk
        with np.errstate(divide="ignore"):
            log_start = np.log(startprob)
            log_trans = np.log(transmat)
        topo = _lib.TOPO_BIDIAG if (is_bidiagonal(transmat) and S <= 32) else _lib.TOPO_DENSE

        # one upload for the five arrays (a Baum-Welch loop packs every iteration: five small pageable copies cost more
        # than the kernels that read them); the device tensors are views into it
        parts = [np.ascontiguousarray(a, dtype=np.float64) for a in (means, var, gconst, log_start, log_trans)]
        flat = torch.from_numpy(np.concatenate([a.ravel() for a in parts])).to(device)
        views, at = [], 0
        for a in parts:
            views.append(flat[at:at + a.size].view(a.shape))
            at += a.size
        return DiagModelPack(means=views[0], vars=views[1], gconst=views[2], log_start=views[3], log_trans=views[4],
                             W=W, S=S, D=D, topology=topo, S_model=S_model, D_model=D_model)._build_blob(exact_only)

    @staticmethod
    def from_models(models, device=None, exact_only=False) -> "DiagModelPack":
        """``models``: objects with hmmlearn's attribute names (startprob_, transmat_, means_, _covars_)."""
        sp = np.stack([np.asarray(m.startprob_, dtype=np.float64) for m in models])
        tm = np.stack([np.asarray(m.transmat_, dtype=np.float64) for m in models])
        mu = np.stack([np.asarray(m.means_, dtype=np.float64) for m in models])
        cv = np.stack([np.asarray(m._covars_, dtype=np.float64) for m in models])
        return DiagModelPack.from_params(sp, tm, mu, cv, device=device, exact_only=exact_only)


def _check_dims(batch: FeatureBatch, pack: "DiagModelPack"):
    if batch.D_model != (pack.D_model or pack.D) or batch.D != pack.D:
        raise ValueError(f"feature dim {batch.D_model} != model dim {pack.D_model or pack.D}")


def _exact_sum_order(batch: FeatureBatch, sum_order: int) -> int:
    """Below 8 terms numpy's pair-wise reduction is the plain left-to-right loop in every layout: features narrower
    than 8 that run padded (kernel_dims) keep their bits in that order only — for one-frame utterances as well, where
    SUM_TVIEW would reduce the padded row pair-wise."""
    if batch.D_model != batch.D and batch.D_model < 8:
        return _lib.SUM_SEQ
    return sum_order


@dataclass
class ViterbiResult:
    scores: "object"      # [N,W] f64 — log_prob of GaussianHMM.decode per (utterance, word)
    last_state: "object"  # [N,W] i32
    best_word: "object"   # [N] i32 (decoder.py:42-47 arg-max; -1 if no score beats -inf)
    best_score: "object"  # [N] f64
    path: "object"        # [total_frames] i32 — state sequence of the selected word


class PrunedDecoder:
    """Pre-allocated ``sapr_viterbi_decode_pruned`` over a fixed batch geometry (decoder.py:35-49 semantics:
    best word, its score, its state path — bit-identical to the all-vocabulary evaluation)."""

    def __init__(self, n_utts, max_T, total_frames, pack: DiagModelPack, device, approx: str = None):
        """``approx``: "auto" (default; env SAPR_APPROX overrides) runs the bounding pass on the matrix cores
        where the pack allows it (PACK_GEMM_OK), "valu" keeps it on the vector ALU — same outputs either way."""
        torch = _torch()
        self.lib = _lib.load()
        approx = approx or os.environ.get("SAPR_APPROX", "auto")
        if approx not in ("auto", "valu"):
            raise ValueError(f"approx must be 'auto' or 'valu', got {approx!r}")
        self.flag_mask = ~_lib.PACK_GEMM_OK if approx == "valu" else ~0
        if not pack.prunable:
            raise _lib.SaprHipError("model pack is not prunable (dense topology or variances outside [1e-20, 1e20])")
        self.N, self.max_T, self.pack = int(n_utts), int(max_T), pack
        n = C.c_size_t(0)
        _lib.check(self.lib.sapr_viterbi_pruned_workspace_bytes(self.N, pack.W, pack.S, self.max_T, C.byref(n)),
                   "sapr_viterbi_pruned_workspace_bytes")
        self.ws_bytes = int(n.value)
        self.workspace = torch.empty(max(self.ws_bytes, 1), dtype=torch.uint8, device=device)
        # result buffers of its own unless the caller passes `out` to launch() (total_frames == 0: none)
        self.best_word = self.best_score = self.path = None
        if total_frames:
            self.best_word = torch.empty(self.N, dtype=torch.int32, device=device)
            self.best_score = torch.empty(self.N, dtype=torch.float64, device=device)
            self.path = torch.empty(int(total_frames), dtype=torch.int32, device=device)

    def launch(self, feats, offsets, order, tie, sum_order, stream, out=None):
        """``offsets`` may be a slice [lo : hi + 1] of a larger batch's table: frame offsets stay absolute, so
        ``feats`` and the path buffer are the whole batch's and best_word / best_score the slice [lo : hi]."""
        p = self.pack
        bw, bs, path = out if out is not None else (self.best_word, self.best_score, self.path)
        _lib.check(self.lib.sapr_viterbi_decode_pruned(
            _lib.ptr(feats), _lib.ptr(offsets), _lib.ptr(order), self.N, p.D, self.max_T, _lib.ptr(p.blob), p.W, p.S,
            tie, sum_order, p.flags & self.flag_mask, _lib.ptr(self.workspace), self.ws_bytes, _lib.ptr(bw),
            _lib.ptr(bs), _lib.ptr(path), stream), "sapr_viterbi_decode_pruned")

    def views(self):
        """(approx_score, approx_eps, exact_score, cand_slot [N,W], cand_count [W]) as torch tensors over the
        workspace — for tests and diagnostics."""
        torch = _torch()
        ptrs = [C.c_void_p() for _ in range(5)]
        _lib.check(self.lib.sapr_viterbi_pruned_views(self.N, self.pack.W, self.max_T, _lib.ptr(self.workspace),
                                                      *[C.byref(q) for q in ptrs]), "sapr_viterbi_pruned_views")
        base = self.workspace.data_ptr()
        W = self.pack.W

        def view(q, dtype, shape):
            off = q.value - base
            nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
            return self.workspace[off:off + nbytes].view(dtype).view(*shape)
        return (view(ptrs[0], torch.float64, (self.N, W)), view(ptrs[1], torch.float64, (self.N, W)),
                view(ptrs[2], torch.float64, (self.N, W)), view(ptrs[3], torch.int32, (self.N, W)),
                view(ptrs[4], torch.int32, (W,)))


def viterbi_decode_best(batch: FeatureBatch, pack: DiagModelPack, tie: int = _lib.TIE_HIGH,
                        sum_order: int = _lib.SUM_TVIEW):
    """Best word, its score and its path per utterance through the pruned decoder when the pack allows it,
    through the all-vocabulary evaluation otherwise (same bits either way) → (best_word, best_score, path)."""
    _check_dims(batch, pack)
    sum_order = _exact_sum_order(batch, sum_order)
    if not pack.prunable or batch.n_utts == 0:
        r = viterbi_decode(batch, pack, tie=tie, sum_order=sum_order)
        return r.best_word, r.best_score, r.path
    dec = PrunedDecoder(batch.n_utts, batch.max_T, batch.total_frames, pack, batch.feats.device)
    dec.launch(batch.feats, batch.offsets, batch.order, tie, sum_order, _lib.current_stream())
    return dec.best_word, dec.best_score, dec.path


def viterbi_decode(batch: FeatureBatch, pack: DiagModelPack, tie: int = _lib.TIE_HIGH,
                   sum_order: int = _lib.SUM_TVIEW, word_sel=None,
                   want_path: bool = True, fast_div=None) -> ViterbiResult:
    """All W models over all utterances: scores, arg-max word and its state path.

    ``sum_order`` names the numpy reduction order hmmlearn's log-density would use for the
    caller's array layout: ``SUM_TVIEW`` for the ``feat_seq.T`` view decoder.py:59 passes
    (default, the decoder path), ``SUM_PAIRWISE`` for a C-contiguous ``(T, D)`` array."""
    torch = _torch()
    lib = _lib.load()
    _check_dims(batch, pack)
    sum_order = _exact_sum_order(batch, sum_order)
    dev = batch.feats.device
    N, W, S = batch.n_utts, pack.W, pack.S
    nbytes = C.c_size_t(0)
    _lib.check(lib.sapr_viterbi_workspace_bytes(N, W, S, batch.max_T, pack.topology, C.byref(nbytes)),
               "sapr_viterbi_workspace_bytes")
    ws = torch.empty(max(int(nbytes.value), 1), dtype=torch.uint8, device=dev)
    scores = torch.empty((N, W), dtype=torch.float64, device=dev)
    last = torch.empty((N, W), dtype=torch.int32, device=dev)
    stream = _lib.current_stream()
    _lib.check(lib.sapr_viterbi_diag_scores(
        _lib.ptr(batch.feats), _lib.ptr(batch.offsets), _lib.ptr(batch.order), N, batch.D, batch.max_T,
        _lib.ptr(pack.blob), W, S, pack.topology, tie, sum_order,
        pack.fast_div if fast_div is None else int(fast_div), _lib.ptr(ws), nbytes.value,
        _lib.ptr(scores), _lib.ptr(last), stream), "sapr_viterbi_diag_scores")
    best_word = torch.empty(N, dtype=torch.int32, device=dev)
    best_score = torch.empty(N, dtype=torch.float64, device=dev)
    path = torch.empty(batch.total_frames, dtype=torch.int32, device=dev) if want_path else None
    if word_sel is not None:
        word_sel = torch.as_tensor(word_sel, dtype=torch.int32, device=dev).contiguous()
    _lib.check(lib.sapr_viterbi_backtrace(
        _lib.ptr(batch.offsets), _lib.ptr(batch.order), N, batch.max_T, W, S, pack.topology,
        _lib.ptr(ws), nbytes.value, _lib.ptr(scores), _lib.ptr(last), _lib.ptr(word_sel),
        _lib.ptr(best_word), _lib.ptr(best_score), _lib.ptr(path), stream), "sapr_viterbi_backtrace")
    return ViterbiResult(scores, last, best_word, best_score, path)


# ------------------------------------------------------------------------------------------
# forward scoring / Baum-Welch E-step (estep.hip)
# ------------------------------------------------------------------------------------------
_TILE = 256


@dataclass
class TileLayout:
    """Utterances grouped into 256-slot tiles that share one word model (sorted by model, then by
    length so a wavefront's 64 trellises have similar T)."""
    slot_utt: "object"        # device int32 [n_tiles*256], -1 = empty
    tile_model: "object"      # device int32 [n_tiles]
    model_tile_off: "object"  # device int32 [W+1]
    n_tiles: int

    @staticmethod
    def build(lengths, utt_model, W, device):
        torch = _torch()
        lengths = np.asarray(lengths, dtype=np.int64)
        utt_model = np.asarray(utt_model, dtype=np.int64)
        slots, tile_model, off = [], [], [0]
        for w in range(W):
            idx = np.nonzero(utt_model == w)[0]
            idx = idx[np.argsort(-lengths[idx], kind="stable")]
            n_t = (idx.size + _TILE - 1) // _TILE
            pad = np.full(n_t * _TILE, -1, dtype=np.int32)
            pad[: idx.size] = idx
            slots.append(pad)
            tile_model += [w] * n_t
            off.append(off[-1] + n_t)
        slot_utt = np.concatenate(slots) if slots else np.zeros(0, np.int32)
        t = torch.from_numpy
        return TileLayout(t(slot_utt).to(device), t(np.asarray(tile_model, dtype=np.int32)).to(device),
                          t(np.asarray(off, dtype=np.int32)).to(device), len(tile_model))


def stats_width(S, D):
    return 2 + S + S * S + S + 2 * S * D


def split_stats(row, S, D, S_model=None, D_model=None):
    """One model's row of sapr_estep_diag's stats (kernel state count S, kernel feature width D) → hmmlearn's stats
    dict for the model's own S_model <= S states and D_model <= D dimensions (dummy padding states and the zero
    feature columns carry exact zeros and are dropped)."""
    o = 0
    m = S if S_model is None else S_model
    dm = D if D_model is None else D_model

    def take(n):
        nonlocal o
        v = row[o:o + n]
        o += n
        return v
    nobs, logprob = take(1)[0], take(1)[0]
    return {"nobs": nobs, "logprob": logprob, "start": take(S)[:m].copy(),
            "trans": take(S * S).reshape(S, S)[:m, :m].copy(), "post": take(S)[:m].copy(),
            "obs": take(S * D).reshape(S, D)[:m, :dm].copy(), "obs**2": take(S * D).reshape(S, D)[:m, :dm].copy()}


def forward_loglik(batch: FeatureBatch, pack: DiagModelPack, utt_model, layout: TileLayout = None):
    """log P(utterance | model utt_model[u]) for every utterance (GaussianHMM.score per sequence)."""
    torch = _torch()
    lib = _lib.load()
    dev = batch.feats.device
    layout = layout or TileLayout.build(batch.lengths, utt_model, pack.W, dev)
    loglik = torch.zeros(batch.n_utts, dtype=torch.float64, device=dev)
    _lib.check(lib.sapr_forward_diag(_lib.ptr(batch.feats), _lib.ptr(batch.offsets), _lib.ptr(layout.slot_utt),
                                     _lib.ptr(layout.tile_model), layout.n_tiles, batch.D, _lib.ptr(pack.blob),
                                     pack.W, pack.S, pack.topology, pack.fast_div, _lib.ptr(loglik),
                                     _lib.current_stream()), "sapr_forward_diag")
    return loglik


class EStep:
    """Pre-allocated E-step over a fixed batch / tile layout (one launch sequence per EM iteration).  The first
    ``run`` stages the batch's features in slot-major order inside the workspace and later runs reuse that copy:
    the feature tensor must not be modified in place between iterations (build a new ``EStep`` for new data)."""

    def __init__(self, batch: FeatureBatch, utt_model, W, S):
        torch = _torch()
        self.lib = _lib.load()
        self.S_model, S = S, kernel_states(S)  # statistics rows are laid out for the kernel's state count
        self.batch, self.W, self.S, self.D = batch, W, S, batch.D
        dev = batch.feats.device
        self.layout = TileLayout.build(batch.lengths, utt_model, W, dev)
        n = C.c_size_t(0)
        _lib.check(self.lib.sapr_fb_workspace_bytes(batch.n_utts, self.layout.n_tiles, S, self.D, batch.max_T,
                                                    C.byref(n)), "sapr_fb_workspace_bytes")
        self.ws_bytes = int(n.value)
        self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self.loglik = torch.zeros(batch.n_utts, dtype=torch.float64, device=dev)
        self.stats = torch.zeros((W, stats_width(S, self.D)), dtype=torch.float64, device=dev)
        self._staged = False   # the workspace holds the slot-major feature copy after the first run()

    def split(self, row):
        """hmmlearn-style stats dict of one model's (host) row."""
        return split_stats(row, self.S, self.D, self.S_model, self.batch.D_model)

    def run(self, pack: DiagModelPack):
        """Returns the device stats tensor [W, width] (caller all-reduces across ranks, then M-step)."""
        if pack.S != self.S:
            raise ValueError(f"model pack has {pack.S} kernel states, this E-step was laid out for {self.S}")
        b, lay = self.batch, self.layout
        _lib.check(self.lib.sapr_estep_diag(
            _lib.ptr(b.feats), _lib.ptr(b.offsets), _lib.ptr(lay.slot_utt), _lib.ptr(lay.tile_model),
            _lib.ptr(lay.model_tile_off), b.n_utts, lay.n_tiles, b.D, b.max_T, _lib.ptr(pack.blob), pack.W,
            pack.S, pack.topology, pack.fast_div | (_lib.ESTEP_STAGED if self._staged else 0),
            _lib.ptr(self.workspace), self.ws_bytes, _lib.ptr(self.loglik),
            _lib.ptr(self.stats), _lib.current_stream()), "sapr_estep_diag")
        self._staged = True    # same batch, same workspace: later iterations skip the feature copy
        return self.stats
