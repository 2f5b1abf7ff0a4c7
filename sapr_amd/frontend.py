"""Device-side MFCC front-end: plan object + batched launch through the C ABI.

Presets (same numbers as ``oracle/mfcc_oracle.py``):

* ``REFERENCE`` – what ``mfcc_extract.py:12-23`` resolves to with librosa 0.10.2 defaults
  (sr 22 050, n_fft 2048, win 661, hop 220, 128 Slaney mels, 13 coefficients, top_db 80).
* ``BENCH`` – BASELINE.json's north-star configuration (sr 16 000, n_fft 512, win 400, hop 160,
  40 mels, 13 coefficients; optional pre-emphasis 0.97 and delta / delta-delta → 39 dims).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

REFERENCE = dict(sr=22050, n_fft=2048, win_length=661, hop_length=220, n_mels=128, n_mfcc=13,
                 top_db=80.0, preemph=0.0, deltas=False)
BENCH = dict(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=40, n_mfcc=13,
             top_db=80.0, preemph=0.0, deltas=False)
BENCH39 = dict(BENCH, preemph=0.97, deltas=True)


def num_frames(n_samples, hop_length):
    """center=True ⇒ 1 + floor(n / hop) (librosa.stft)."""
    return 1 + np.asarray(n_samples) // hop_length


class MfccPlan:
    """Tables for one MFCC configuration, resident on the current device."""

    def __init__(self, sr=22050, n_fft=2048, win_length=661, hop_length=220, n_mels=128, n_mfcc=13,
                 top_db=80.0, preemph=0.0, deltas=False, fmin=0.0, fmax=None, max_frames=128):
        """``max_frames`` > 0: fused mode (an utterance's log-mel matrix stays in LDS; utterances of at
        most that many frames); ``max_frames=0``: two-pass mode through an HBM workspace, any length.
        A fused layout that does not fit the 160 KiB of LDS silently becomes two-pass."""
        _lib.require_gpu()
        lib = _lib.load()
        self.cfg = dict(sr=sr, n_fft=n_fft, win_length=win_length, hop_length=hop_length, n_mels=n_mels,
                        n_mfcc=n_mfcc, top_db=top_db, preemph=preemph, deltas=bool(deltas))
        self.hop_length = int(hop_length)
        self.max_frames = int(max_frames)
        self._h = C.c_void_p()
        _lib.check(lib.sapr_mfcc_plan_create(float(sr), n_fft, win_length, hop_length, n_mels, n_mfcc,
                                             float(fmin), float(fmax or 0.0), float(top_db), float(preemph),
                                             1 if deltas else 0, self.max_frames, C.byref(self._h)),
                   "sapr_mfcc_plan_create")
        d_out, mf, lds = C.c_int32(), C.c_int32(), C.c_int64()
        _lib.check(lib.sapr_mfcc_plan_info(self._h, C.byref(d_out), C.byref(mf), C.byref(lds), None),
                   "sapr_mfcc_plan_info")
        self.d_out, self.lds_bytes = int(d_out.value), int(lds.value)
        self.two_pass = int(mf.value) == 0
        self._ws = None

    @property
    def handle(self) -> int:
        """The C plan handle as an integer (what ``torch.ops.sapr.mfcc_batch`` takes)."""
        return int(self._h.value)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.load().sapr_mfcc_plan_destroy(h)
            except Exception:
                pass

    def __call__(self, pcm, sample_lengths, grid_blocks=0):
        """``pcm``: device float32 [total_samples]; returns (feats [total_frames, d_out] device
        float32, frame_lengths host int64)."""
        import torch
        lib = _lib.load()
        sample_lengths = np.asarray(sample_lengths, dtype=np.int64)
        if pcm.dtype != torch.float32 or pcm.dim() != 1 or not pcm.is_contiguous():
            raise ValueError("pcm must be a contiguous 1-D float32 tensor")
        if pcm.shape[0] != sample_lengths.sum():
            raise ValueError("pcm length does not match sum(sample_lengths)")
        frames = num_frames(sample_lengths, self.hop_length).astype(np.int64)
        if not self.two_pass and frames.size and frames.max() > self.max_frames:
            raise ValueError(f"utterance of {int(frames.max())} frames exceeds the plan's max_frames="
                             f"{self.max_frames}")
        if self.cfg["deltas"] and frames.size and frames.min() < 9:
            raise ValueError("delta features need at least 9 frames per utterance (librosa.feature.delta)")
        so = np.zeros(sample_lengths.shape[0] + 1, dtype=np.int64)
        np.cumsum(sample_lengths, out=so[1:])
        fo = np.zeros_like(so)
        np.cumsum(frames, out=fo[1:])
        dev = pcm.device
        so_d, fo_d = torch.from_numpy(so).to(dev), torch.from_numpy(fo).to(dev)
        out = torch.empty((int(fo[-1]), self.d_out), dtype=torch.float32, device=dev)
        ws, ws_bytes = self.workspace(int(fo[-1]), sample_lengths.shape[0], dev)
        _lib.check(lib.sapr_mfcc_batch(self._h, _lib.ptr(pcm), _lib.ptr(so_d), _lib.ptr(fo_d),
                                       sample_lengths.shape[0], int(fo[-1]), _lib.ptr(out), int(grid_blocks),
                                       _lib.ptr(ws), ws_bytes, _lib.current_stream()), "sapr_mfcc_batch")
        return out, frames

    def workspace(self, total_frames, n_utts, device):
        """(tensor or None, bytes): the two-pass log-mel workspace, cached and grown on demand."""
        import torch
        n = C.c_size_t(0)
        _lib.check(_lib.load().sapr_mfcc_workspace_bytes(self._h, int(total_frames), int(n_utts), C.byref(n)),
                   "sapr_mfcc_workspace_bytes")
        if n.value == 0:
            return None, 0
        if self._ws is None or self._ws.numel() < n.value or self._ws.device != device:
            self._ws = torch.empty(int(n.value), dtype=torch.uint8, device=device)
        return self._ws, int(n.value)


def mfcc_batch(signals, plan: MfccPlan):
    """List of 1-D arrays → list of (d_out, T) float32 numpy arrays (reference layout).  float arrays are
    samples in [-1, 1); if every array is int16 (raw 16-bit PCM) the batch is uploaded as int16 — half the
    PCIe traffic — and scaled by 1/32768 on the device (``sapr_pcm16_to_f32``), same values as the host
    conversion."""
    import torch
    dev = _lib.require_gpu()
    lens = np.asarray([len(s) for s in signals], dtype=np.int64)
    if len(signals) and all(np.asarray(s).dtype == np.int16 for s in signals):
        raw = torch.from_numpy(np.concatenate([np.asarray(s, dtype=np.int16) for s in signals])).to(dev)
        pcm = torch.empty(raw.numel(), dtype=torch.float32, device=dev)
        _lib.check(_lib.load().sapr_pcm16_to_f32(_lib.ptr(raw), raw.numel(), _lib.ptr(pcm), _lib.current_stream()),
                   "sapr_pcm16_to_f32")
    else:
        packed = np.concatenate([np.asarray(s, dtype=np.float32) for s in signals]) if len(signals) else \
            np.zeros(0, np.float32)
        pcm = torch.from_numpy(packed).to(dev)
    feats, frames = plan(pcm, lens)
    (host,) = _lib.to_host(feats)
    out, o = [], 0
    for t in frames:
        out.append(np.ascontiguousarray(host[o:o + t].T))
        o += t
    return out
