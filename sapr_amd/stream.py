"""Chunked recogniser for corpora that do not fit in HBM at once (BASELINE configs[4]: 1 M utterances).

The reference walks one file at a time (``mfcc_extract.py:35-49`` then ``decoder.py:58-70``).  Here a
corpus is a sequence of CHUNKS of 16-bit PCM in host memory (what WAV files hold; 2 bytes per sample
across PCIe instead of 4).  Per chunk:

    copy stream      int16 PCM  host (pinned) --H2D--> one of two device buffers
    compute stream   sapr_pcm16_to_f32 -> sapr_mfcc_batch -> sapr_viterbi_decode_pruned
                     -> best word / score / state path --D2H--> pinned result buffers

The H2D of chunk k+1 runs while chunk k computes (two HIP streams, events for the two hand-offs), so
in steady state a chunk costs max(PCIe time, kernel time).  One rank owns a contiguous range of chunks
(``dist.shard_range``): no collective on this path.

PyTorch is used for device memory, pinned host memory, streams and events only.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .frontend import MfccPlan
from .pipeline import RecognizerPipeline
from .trellis import DiagModelPack


@dataclass
class StreamReport:
    n_utts: int = 0
    frames: int = 0
    wall_s: float = 0.0           # PCIe-inclusive: first H2D issued -> last result on the host
    kernel_s: float = 0.0         # sum over chunks of HIP-event time around the compute launches
    h2d_s: float = 0.0            # sum over chunks of HIP-event time around the PCM upload
    chunk_kernel_ms: list = field(default_factory=list)

    @property
    def frames_per_s_pcie_inclusive(self):
        return self.frames / self.wall_s if self.wall_s else 0.0

    @property
    def frames_per_s_kernels_only(self):
        return self.frames / self.kernel_s if self.kernel_s else 0.0


class StreamingRecognizer:
    """``chunks`` for :meth:`run` is an iterable of ``(pcm16, sample_lengths)``: a 1-D int16 array / tensor of the
    chunk's utterances back to back (pinned host memory makes the upload asynchronous) and the per-utterance
    sample counts.  Chunks with the same length table share one pre-allocated :class:`RecognizerPipeline`."""

    def __init__(self, plan: MfccPlan, pack: DiagModelPack, device=None, mode: str = "auto"):
        import torch
        self.torch = torch
        self.dev = device or _lib.require_gpu()
        self.plan, self.pack, self.mode = plan, pack, mode
        self.lib = _lib.load()
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.compute_stream = torch.cuda.Stream(device=self.dev)
        self._pipes = {}
        self._dev16 = [None, None]
        self._pcmf = None
        self._host = {}

    def _host_results(self, slot, pipe):
        """Pinned host buffers for one slot's results (two slots: chunk k's results are read on the host
        while chunk k+1 computes), grown on demand."""
        torch = self.torch
        want = (pipe.best_word, pipe.best_score, pipe.path)
        have = self._host.get(slot)
        if have is None or any(h.shape[0] < t.shape[0] for h, t in zip(have, want)):
            have = tuple(torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in want)
            self._host[slot] = have
        return tuple(h[: t.shape[0]] for h, t in zip(have, want))

    def _pipeline(self, sample_lengths):
        sl = np.asarray(sample_lengths, dtype=np.int64)
        key = (sl.shape[0], int(sl.sum()), hash(sl.tobytes()))
        if key not in self._pipes:
            self._pipes[key] = RecognizerPipeline(self.plan, self.pack, sl, device=self.dev, mode=self.mode)
        return self._pipes[key]

    def run(self, chunks, on_result=None, keep_results: bool = True):
        """Returns ``(results, report)``; ``results`` is a list of ``(best_word, best_score, path)`` numpy
        arrays per chunk (or empty when ``keep_results`` is False; ``on_result(k, bw, bs, path)`` is called
        with host arrays either way)."""
        torch = self.torch
        rep = StreamReport()
        results = []
        ev_up = [torch.cuda.Event(enable_timing=True) for _ in range(2)]      # upload of slot s done
        ev_up0 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev_free = [torch.cuda.Event() for _ in range(2)]                       # slot s consumed by the conversion
        pending = None   # (k, pipe, host tensors, events) of the chunk whose results are still in flight

        def finish(p):
            k, pipe, hb, e0, e1, e2 = p
            e2.synchronize()   # kernels done AND results on the host
            rep.kernel_s += e0.elapsed_time(e1) * 1e-3
            rep.chunk_kernel_ms.append(e0.elapsed_time(e1))
            out = tuple(h.numpy() for h in hb)
            if on_result is not None:
                on_result(k, *out)
            if keep_results:
                results.append(tuple(o.copy() for o in out))

        t_start = None
        rep_chunks = 0
        for k, (pcm16, sample_lengths) in enumerate(chunks):
            pcm16 = torch.as_tensor(pcm16)
            if pcm16.dtype != torch.int16 or pcm16.dim() != 1:
                raise ValueError("a chunk's PCM must be a 1-D int16 array")
            pipe = self._pipeline(sample_lengths)
            if pcm16.shape[0] != pipe.total_samples:
                raise ValueError("chunk PCM length does not match its sample_lengths")
            s = k % 2
            n = pcm16.shape[0]
            # The buffers are allocated on the default stream but used on the copy and compute streams only.  Before one
            # is REPLACED by a larger one (a later chunk is longer) every launch that may still read or write the old
            # block must have finished, or the caching allocator could hand that block to someone else mid-flight:
            # drain the device first (rare: chunks of a corpus normally share one size), and tell the allocator which
            # streams use the new block.
            if self._dev16[s] is None or self._dev16[s].shape[0] < n:
                torch.cuda.synchronize(self.dev)
                self._dev16[s] = torch.empty(n, dtype=torch.int16, device=self.dev)
                self._dev16[s].record_stream(self.copy_stream)
                self._dev16[s].record_stream(self.compute_stream)
            if self._pcmf is None or self._pcmf.shape[0] < n:
                torch.cuda.synchronize(self.dev)
                self._pcmf = torch.empty(n, dtype=torch.float32, device=self.dev)
                self._pcmf.record_stream(self.compute_stream)
            if t_start is None:
                torch.cuda.synchronize(self.dev)
                t_start = time.perf_counter()
            # ---- upload on the copy stream (waits until the conversion of chunk k-2 has read this slot)
            with torch.cuda.stream(self.copy_stream):
                if k >= 2:
                    self.copy_stream.wait_event(ev_free[s])
                ev_up0[s].record(self.copy_stream)
                self._dev16[s][:n].copy_(pcm16, non_blocking=True)
                ev_up[s].record(self.copy_stream)
            # ---- compute stream
            cs = self.compute_stream
            st = _lib.C.c_void_p(cs.cuda_stream)
            with torch.cuda.stream(cs):
                cs.wait_event(ev_up[s])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cs)
                pcmf = self._pcmf[:n]
                _lib.check(self.lib.sapr_pcm16_to_f32(_lib.ptr(self._dev16[s]), n, _lib.ptr(pcmf), st),
                           "sapr_pcm16_to_f32")
                ev_free[s].record(cs)
                pipe.launch_mfcc(pcmf, st)
                pipe.launch_decode(st)
                e1.record(cs)
                # results of the PREVIOUS chunk were copied out of this pipeline's buffers before these
                # launches were enqueued (same stream), so buffers can be shared between chunks
                hb = self._host_results(s, pipe)
                for h, t in zip(hb, (pipe.best_word, pipe.best_score, pipe.path)):
                    h.copy_(t, non_blocking=True)
                e2 = torch.cuda.Event()
                e2.record(cs)
            if pending is not None:
                finish(pending)
            pending = (k, pipe, hb, e0, e1, e2)
            rep_chunks += 1
            rep.n_utts += pipe.n_utts
            rep.frames += pipe.total_frames
        if pending is not None:
            finish(pending)
        torch.cuda.synchronize(self.dev)
        rep.wall_s = time.perf_counter() - t_start if t_start is not None else 0.0
        for s in range(min(2, rep_chunks)):   # (a slot a one-chunk run never used has no recorded events)
            rep.h2d_s += ev_up0[s].elapsed_time(ev_up[s]) * 1e-3   # last upload of each slot only: a sample
        return results, rep


def shard_chunks(n_chunks: int, rank: int = None, world: int = None):
    """Contiguous range of chunk indices owned by this rank (utterance shards need no collective)."""
    from .dist import shard_range
    return range(*shard_range(n_chunks, rank, world))
