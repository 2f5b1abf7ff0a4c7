"""Pre-allocated PCM → MFCC → all-vocabulary Viterbi pipeline (the BASELINE metric's hot path).

Everything a step needs (offset tables, feature buffer, back-pointer workspace, score / path
buffers) is allocated once; :meth:`RecognizerPipeline.run` is a fixed sequence of asynchronous kernel
launches on the current stream and never touches the host allocator — the batched equivalent of
``extract_mfcc`` (mfcc_extract.py:10-27) followed by ``Decoder.decode_sequence`` for every
utterance (decoder.py:35-49).

Two decode modes with identical ``best_word`` / ``best_score`` / ``path`` bits:

* ``pruned`` (default when the model pack allows it): ``sapr_viterbi_decode_pruned`` — a float32 bounding
  pass over the whole vocabulary, the exact lattice only for the words that can still win.  ``pieces`` > 1 cuts
  the batch into contiguous utterance ranges decoded on separate HIP streams, so that one piece's exact pass
  (~1.5 wavefronts per SIMD) could share the chip with the next piece's bounding pass; measured on MI355X it
  does not pay (2.45 ms with 1 piece, 2.39 with 2, 2.79 with 4 per 100 000 utterances), so the default is 1;
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
                 sum_order=_lib.SUM_TVIEW, device=None, mode: str = "auto", pieces: int = 1):
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
            self.best_word = torch.empty(self.n_utts, dtype=torch.int32, device=dev)
            self.best_score = torch.empty(self.n_utts, dtype=torch.float64, device=dev)
            self.path = torch.empty(self.total_frames, dtype=torch.int32, device=dev)
            self.scores = self.last_state = None
            # contiguous utterance ranges, each with its own decoder workspace, length-sorted order and stream
            n_p = max(1, min(int(pieces), self.n_utts // 4096))
            cuts = np.linspace(0, self.n_utts, n_p + 1).astype(np.int64)
            self._pieces = []
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                lo, hi = int(lo), int(hi)
                dec = PrunedDecoder(hi - lo, int(fr[lo:hi].max()), 0, pack, dev)
                order_p = t(np.argsort(-fr[lo:hi], kind="stable").astype(np.int32)).to(dev)
                stream = torch.cuda.Stream(device=dev) if n_p > 1 else None
                self._pieces.append((lo, hi, dec, order_p, stream))
            self.pruned = self._pieces[0][2] if n_p == 1 else None
            self._ev_in = torch.cuda.Event()
            self._ev_out = [torch.cuda.Event() for _ in self._pieces]
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
            torch = self.torch
            if len(self._pieces) == 1:
                lo, hi, dec, order_p, _ = self._pieces[0]
                dec.launch(self.feats, self.frame_offsets, order_p, self.tie, self.sum_order, stream,
                           out=(self.best_word, self.best_score, self.path))
                return
            main = torch.cuda.ExternalStream(stream.value) if stream.value else torch.cuda.default_stream(self.feats.device)
            self._ev_in.record(main)
            for (lo, hi, dec, order_p, st), ev in zip(self._pieces, self._ev_out):
                st.wait_event(self._ev_in)
                dec.launch(self.feats, self.frame_offsets[lo:hi + 1], order_p, self.tie, self.sum_order,
                           C.c_void_p(st.cuda_stream),
                           out=(self.best_word[lo:hi], self.best_score[lo:hi], self.path))
                ev.record(st)
            for ev in self._ev_out:
                main.wait_event(ev)
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

    def pruned_views(self):
        """(approx_score, approx_eps, exact_score, cand_slot) as [n_utts, W] tensors in utterance order and the
        per-word survivor counts [W], gathered from the pieces' workspaces — tests / diagnostics."""
        torch = self.torch
        v = [dec.views() for _, _, dec, _, _ in self._pieces]
        return tuple(torch.cat([x[i] for x in v], dim=0) for i in range(4)) + (sum(x[4] for x in v),)

    def run(self, pcm):
        """pcm: device float32 [total_samples].  Results land in self.best_word / best_score / path."""
        if pcm.shape[0] != self.total_samples:
            raise ValueError("pcm length does not match the pipeline's sample_lengths")
        stream = _lib.current_stream()
        self.launch_mfcc(pcm, stream)
        self.launch_decode(stream)
        return self.best_word, self.best_score, self.path
