"""Pre-allocated PCM → MFCC → all-vocabulary Viterbi pipeline (the BASELINE metric's hot path).

Everything a step needs (offset tables, feature buffer, back-pointer workspace, score / path
buffers) is allocated once; :meth:`RecognizerPipeline.run` is a fixed sequence of asynchronous kernel
launches on the current stream and never touches the host allocator — the batched equivalent of
``extract_mfcc`` (mfcc_extract.py:10-27) followed by ``Decoder.decode_sequence`` for every
utterance (decoder.py:35-49).

Two decode modes with identical ``best_word`` / ``best_score`` / ``path`` bits:

* ``pruned`` (default when the model pack allows it): ``sapr_viterbi_decode_pruned`` — a float32 bounding
  pass over the whole vocabulary, the exact lattice only for the words that can still win;
* ``full``: ``sapr_viterbi_diag_scores`` + ``sapr_viterbi_backtrace`` — every word's exact score is
  materialised in ``self.scores`` (what ``GaussianHMM.decode`` would return for each model).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .frontend import MfccPlan, num_frames
from .trellis import DiagModelPack, PrunedDecoder


class RecognizerPipeline:
    def __init__(self, plan: MfccPlan, pack: DiagModelPack, sample_lengths, tie=_lib.TIE_HIGH,
                 sum_order=_lib.SUM_TVIEW, device=None, mode: str = "auto"):
        import torch
        self.torch = torch
        self.lib = _lib.load()
        dev = device or _lib.require_gpu()
        if plan.d_out != pack.D:
            raise ValueError(f"front-end emits {plan.d_out}-dim features, models expect {pack.D}")
        self.plan, self.pack, self.tie, self.sum_order = plan, pack, tie, sum_order
        sl = np.asarray(sample_lengths, dtype=np.int64)
        self.n_utts = int(sl.shape[0])
        fr = num_frames(sl, plan.hop_length).astype(np.int64)
        if not plan.two_pass and fr.max() > plan.max_frames:
            raise ValueError("utterance longer than the plan's max_frames")
        self.frames = fr
        self.total_frames = int(fr.sum())
        self.total_samples = int(sl.sum())
        self.max_T = int(fr.max())
        so = np.zeros(self.n_utts + 1, dtype=np.int64)
        np.cumsum(sl, out=so[1:])
        fo = np.zeros(self.n_utts + 1, dtype=np.int64)
        np.cumsum(fr, out=fo[1:])
        order = np.argsort(-fr, kind="stable").astype(np.int32)
        t = torch.from_numpy
        self.sample_offsets = t(so).to(dev)
        self.frame_offsets = t(fo).to(dev)
        self.order = t(order).to(dev)
        if mode not in ("auto", "pruned", "full"):
            raise ValueError("mode must be 'auto', 'pruned' or 'full'")
        if mode == "pruned" and not pack.prunable:
            raise _lib.SaprHipError("mode='pruned' needs a bidiagonal model pack inside the bound's domain")
        self.mode = "pruned" if (mode != "full" and pack.prunable) else "full"
        self.feats = torch.empty((self.total_frames, plan.d_out), dtype=torch.float32, device=dev)
        self.mfcc_ws, self.mfcc_ws_bytes = plan.workspace(self.total_frames, self.n_utts, dev)
        if self.mode == "pruned":
            self.pruned = PrunedDecoder(self.n_utts, self.max_T, self.total_frames, pack, dev)
            self.best_word, self.best_score, self.path = (self.pruned.best_word, self.pruned.best_score,
                                                          self.pruned.path)
            self.scores = self.last_state = None
            return
        nbytes = C.c_size_t(0)
        _lib.check(self.lib.sapr_viterbi_workspace_bytes(self.n_utts, pack.W, pack.S, self.max_T,
                                                         pack.topology, C.byref(nbytes)),
                   "sapr_viterbi_workspace_bytes")
        self.ws_bytes = int(nbytes.value)
        self.workspace = torch.empty(max(self.ws_bytes, 1), dtype=torch.uint8, device=dev)
        self.scores = torch.empty((self.n_utts, pack.W), dtype=torch.float64, device=dev)
        self.last_state = torch.empty((self.n_utts, pack.W), dtype=torch.int32, device=dev)
        self.best_word = torch.empty(self.n_utts, dtype=torch.int32, device=dev)
        self.best_score = torch.empty(self.n_utts, dtype=torch.float64, device=dev)
        self.path = torch.empty(self.total_frames, dtype=torch.int32, device=dev)

    def launch_decode(self, stream):
        """Everything after the front-end: best word, score and path of every utterance."""
        if self.mode == "pruned":
            self.pruned.launch(self.feats, self.frame_offsets, self.order, self.tie, self.sum_order, stream)
        else:
            self.launch_viterbi(stream)
            self.launch_backtrace(stream)

    # the launches, separately callable so bench.py can bracket each with events
    def launch_mfcc(self, pcm, stream):
        _lib.check(self.lib.sapr_mfcc_batch(self.plan._h, _lib.ptr(pcm), _lib.ptr(self.sample_offsets),
                                            _lib.ptr(self.frame_offsets), self.n_utts, self.total_frames,
                                            _lib.ptr(self.feats), 0, _lib.ptr(self.mfcc_ws), self.mfcc_ws_bytes,
                                            stream), "sapr_mfcc_batch")

    def launch_viterbi(self, stream):
        p = self.pack
        _lib.check(self.lib.sapr_viterbi_diag_scores(
            _lib.ptr(self.feats), _lib.ptr(self.frame_offsets), _lib.ptr(self.order), self.n_utts, p.D,
            self.max_T, _lib.ptr(p.blob), p.W, p.S, p.topology, self.tie, self.sum_order, p.fast_div,
            _lib.ptr(self.workspace), self.ws_bytes, _lib.ptr(self.scores), _lib.ptr(self.last_state),
            stream), "sapr_viterbi_diag_scores")

    def launch_backtrace(self, stream):
        p = self.pack
        _lib.check(self.lib.sapr_viterbi_backtrace(
            _lib.ptr(self.frame_offsets), _lib.ptr(self.order), self.n_utts, self.max_T, p.W, p.S,
            p.topology, _lib.ptr(self.workspace), self.ws_bytes, _lib.ptr(self.scores),
            _lib.ptr(self.last_state), None, _lib.ptr(self.best_word), _lib.ptr(self.best_score),
            _lib.ptr(self.path), stream), "sapr_viterbi_backtrace")

    def run(self, pcm):
        """pcm: device float32 [total_samples].  Results land in self.best_word / best_score / path."""
        if pcm.shape[0] != self.total_samples:
            raise ValueError("pcm length does not match the pipeline's sample_lengths")
        stream = _lib.current_stream()
        self.launch_mfcc(pcm, stream)
        self.launch_decode(stream)
        return self.best_word, self.best_score, self.path
