"""CPU oracle: restatement of the librosa MFCC chain invoked by the reference.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

**PARITY UNPINNED against the reference itself** (pinned piecewise, see below).  The arithmetic lives in
librosa 0.10.2.post1
(``/root/reference/assignment2/poetry.lock:679-680``; numpy 1.26.4, scipy 1.13.1),
un-vendored and absent from the build container.  The reference's only call
site is ``mfcc_extract.py:12-23``::

    y, sr = librosa.load(path)                         # mono, 22 050 Hz, float32
    librosa.feature.mfcc(y=y, sr=sr, n_mfcc=13, win_length=int(0.03*sr),
                         hop_length=int(0.01*sr), window="hamming", center=True)

and its tests pin only the output shape (``tests/test_mfcc_extract.py:31-45``).
This file restates librosa's published algorithm (``core/spectrum.py`` ``stft`` /
``_spectrogram`` / ``power_to_db``, ``filters.py`` ``mel``, ``core/convert.py``
``mel_frequencies``, ``feature/spectral.py`` ``melspectrogram`` / ``mfcc``,
``feature/utils.py`` ``delta``) with librosa's dtype flow (float64 window*frame →
rFFT → complex64 → float32 power/mel/dB/DCT).  ``tests/test_oracle_mfcc.py`` pins every piece
against code that shares nothing with this file: the numbers librosa publishes in its own
docstrings (``hz_to_mel``, ``mel_to_hz``, ``mel_frequencies(n_mels=40)``, ``filters.mel``),
``scipy.signal.stft`` and ``torch.stft`` for framing/centring/window/rFFT, scipy's
``get_window`` / ``fft.dct`` / ``savgol_filter``, an end-to-end float64 recomputation, and (round 4) the
librosa-compatible routines ``transformers.audio_utils`` ships for the Whisper feature extractor —
``mel_filter_bank(norm="slaney", mel_scale="slaney")``, ``window_function``, ``spectrogram(center=True,
pad_mode="constant", power=2, log_mel="dB", db_range=80)`` — whose filterbank equals this file's to 4e-16 and
whose log-mel spectrogram, through scipy's DCT, gives the 13 cepstra of both presets to 4.1e-5.

Two presets (``sapr_amd/mfcc_extract.py`` carries the same numbers):

* ``REFERENCE`` – what ``mfcc_extract.py:12-23`` resolves to with librosa defaults:
  sr 22 050, n_fft 2048, win 661, hop 220, 128 Slaney mels, 13 coefficients,
  top_db 80, no pre-emphasis, no deltas.
* ``BENCH`` – BASELINE.json's north-star configuration, which has NO reference
  semantics: sr 16 000, n_fft 512, win 400 (25 ms), hop 160 (10 ms), 40 mels,
  13 coefficients, optional pre-emphasis 0.97 (``y'[0] = y[0]``, i.e. x[-1] := 0)
  and optional delta / delta-delta (Savitzky-Golay width 9, ``mode="interp"``,
  librosa.feature.delta semantics).
"""
from __future__ import annotations

import numpy as np
import scipy.fft
import scipy.signal

REFERENCE = dict(sr=22050, n_fft=2048, win_length=661, hop_length=220, n_mels=128,
                 n_mfcc=13, top_db=80.0, preemph=0.0, deltas=False)
BENCH = dict(sr=16000, n_fft=512, win_length=400, hop_length=160, n_mels=40,
             n_mfcc=13, top_db=80.0, preemph=0.0, deltas=False)


# ------------------------------------------------------------------- mel filterbank
def hz_to_mel(f):
    """librosa.hz_to_mel (Slaney): linear below 1 kHz (200/3 Hz per mel), log above."""
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(f >= min_log_hz, min_log_mel + np.log(f / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None, dtype=np.float32):
    """librosa.filters.mel(htk=False, norm="slaney") → (n_mels, 1+n_fft//2)."""
    if fmax is None:
        fmax = sr / 2.0
    nb = 1 + n_fft // 2
    w = np.zeros((n_mels, nb), dtype=dtype)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    w *= enorm[:, None].astype(dtype)
    return w


def dct_matrix(n_mfcc, n_mels, dtype=np.float64):
    """Rows 0..n_mfcc-1 of the orthonormal DCT-II matrix (scipy.fftpack.dct norm="ortho")."""
    n = np.arange(n_mels)
    k = np.arange(n_mfcc)[:, None]
    m = 2.0 * np.cos(np.pi * k * (2 * n + 1) / (2.0 * n_mels))
    m[0] *= np.sqrt(1.0 / (4 * n_mels))
    m[1:] *= np.sqrt(1.0 / (2 * n_mels))
    return m.astype(dtype)


def padded_window(win_length, n_fft):
    """Periodic Hamming (scipy get_window(fftbins=True)) zero-padded centred to n_fft (librosa util.pad_center)."""
    w = scipy.signal.get_window("hamming", win_length, fftbins=True)
    lpad = (n_fft - win_length) // 2
    out = np.zeros(n_fft)
    out[lpad:lpad + win_length] = w
    return out


# ------------------------------------------------------------------------- chain
def preemphasis(y, coef):
    """North-star addition (no reference semantics): y'[n] = y[n] - coef*y[n-1], y'[0] = y[0]; float32 arithmetic."""
    y = np.asarray(y, dtype=np.float32)
    out = y.copy()
    out[1:] = y[1:] - np.float32(coef) * y[:-1]
    return out


def num_frames(n_samples, hop_length):
    """center=True ⇒ 1 + floor(n / hop)."""
    return 1 + n_samples // hop_length


def power_spectrogram(y, n_fft, win_length, hop_length):
    """|stft|**2 with center=True zero padding → (1+n_fft//2, T) float32 (librosa _spectrogram, power=2)."""
    y = np.asarray(y, dtype=np.float32)
    yp = np.pad(y, n_fft // 2, mode="constant")
    T = num_frames(len(y), hop_length)
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(T)[None, :]
    frames = yp[idx]                                            # (n_fft, T) float32
    win = padded_window(win_length, n_fft)[:, None]             # float64
    spec = np.fft.rfft(win * frames, axis=0).astype(np.complex64)  # numpy.fft in double, stored complex64
    return (np.abs(spec) ** 2.0).astype(np.float32)


def power_to_db(S, amin=1e-10, top_db=80.0):
    """librosa.power_to_db(ref=1.0) in S's dtype; top_db clip against the max of the WHOLE array."""
    S = np.asarray(S)
    log_spec = (10.0 * np.log10(np.maximum(np.asarray(amin, S.dtype), S))).astype(S.dtype)
    log_spec -= np.asarray(10.0 * np.log10(np.maximum(amin, 1.0)), S.dtype)
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - np.asarray(top_db, S.dtype))
    return log_spec


def delta(data, order, width=9):
    """librosa.feature.delta(width=9, order, axis=-1, mode="interp")."""
    return scipy.signal.savgol_filter(data, width, deriv=order, polyorder=order, axis=-1, mode="interp")


def mfcc(y, sr=22050, n_fft=2048, win_length=661, hop_length=220, n_mels=128, n_mfcc=13,
         top_db=80.0, preemph=0.0, deltas=False):
    """Full chain → (n_mfcc [*3 if deltas], T) float32 channel-first, like ``mfcc_extract.py:15-24``."""
    y = np.asarray(y, dtype=np.float32)
    if preemph:
        y = preemphasis(y, preemph)
    P = power_spectrogram(y, n_fft, win_length, hop_length)
    M = mel_filterbank(sr, n_fft, n_mels)
    mel = (M @ P).astype(np.float32)
    db = power_to_db(mel, top_db=top_db)
    c = scipy.fft.dct(db, axis=0, type=2, norm="ortho")[:n_mfcc].astype(np.float32)
    if deltas:
        c = np.concatenate([c, delta(c, 1), delta(c, 2)], axis=0).astype(np.float32)
    return c


def delta_edge_matrices(order, width=9):
    """Dense (T,T)-free description of :func:`delta`: returns (interior taps (width,),
    head (width//2, width), tail (width//2, width)) so that
    ``out[t] = taps . x[t-h : t+h+1]`` in the interior and ``out[:h] = head @ x[:width]``,
    ``out[-h:] = tail @ x[-width:]`` at the edges (polynomial fit of mode="interp")."""
    h = width // 2
    taps = scipy.signal.savgol_coeffs(width, order, deriv=order, use="dot")
    head = np.stack([scipy.signal.savgol_coeffs(width, order, deriv=order, pos=p, use="dot") for p in range(h)])
    tail = np.stack([scipy.signal.savgol_coeffs(width, order, deriv=order, pos=width - h + p, use="dot")
                     for p in range(h)])
    return taps, head, tail


# ------------------------------------------------------------- synthetic workloads
def synth_utterances(n, n_samples=16000, sr=16000, seed=0):
    """SURVEY.md §8(d) config 2 generator: three random sinusoids (100-4000 Hz, random
    phase, amplitudes U(0.05,0.3)) + N(0,0.01^2) noise, first/last 100 ms zeroed in
    10 % of utterances.  Returns (n, n_samples) float32."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples, dtype=np.float64) / sr
    out = np.empty((n, n_samples), dtype=np.float32)
    for u in range(n):
        f = rng.uniform(100.0, 4000.0, 3)
        ph = rng.uniform(0, 2 * np.pi, 3)
        a = rng.uniform(0.05, 0.3, 3)
        y = (a[:, None] * np.sin(2 * np.pi * f[:, None] * t[None] + ph[:, None])).sum(0)
        y += rng.normal(0, 0.01, n_samples)
        if rng.uniform() < 0.1:
            k = sr // 10
            y[:k] = 0
            y[-k:] = 0
        out[u] = y.astype(np.float32)
    return out
