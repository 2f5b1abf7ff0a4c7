"""Host-side mirror of ``assignment2/mfcc_extract.py`` over the HIP MFCC kernel.

Same function names, arguments, return layout (``(13, T)`` float32, channel-first) and error
behaviour as the reference; the arithmetic of ``librosa.feature.mfcc`` (``mfcc_extract.py:15-23``)
runs in ``libsapr_hip.so`` (``sapr_mfcc_batch``).  File IO stays on the host like the reference's.

Audio ingest: WAV files (PCM 8/16/24/32-bit or IEEE float) are parsed here;
anything else (the reference's dataset is mp3) needs a decoder this image does not have and raises.
``librosa.load`` resamples to 22 050 Hz with soxr_hq, whose filter cannot be reproduced here; other
rates go through a polyphase Kaiser-windowed FIR with ``scipy.signal.resample_poly``'s definition,
run on the GPU (``sapr_resample_poly``; only the tap design is host work) — numerically close to,
not identical with, soxr.
"""
from __future__ import annotations

import logging
import os
import struct
from fractions import Fraction

import numpy as np

from . import _lib

TARGET_SR = 22050  # librosa.load default (mfcc_extract.py:12)
_plans = {}


def read_wav(path: str, raw16: bool = False):
    """Minimal RIFF/WAVE reader → (float32 mono signal in [-1, 1), sample_rate).  ``raw16=True`` returns
    mono 16-bit PCM files as the int16 samples themselves (``frontend.mfcc_batch`` converts on the GPU)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file (only WAV input is supported; mp3 needs a decoder)")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:  # WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first word
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    tag, ch, sr, bits = fmt
    if tag == 1:
        if bits == 8:
            x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            if raw16 and ch == 1:
                return np.frombuffer(pcm, dtype="<i2").astype(np.int16), int(sr)
            x = np.frombuffer(pcm, dtype="<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(pcm[: len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v & 0x800000, v - 0x1000000, v)
            x = v.astype(np.float32) / 8388608.0
        elif bits == 32:
            x = np.frombuffer(pcm, dtype="<i4").astype(np.float32) / 2147483648.0
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    elif tag == 3:
        x = np.frombuffer(pcm, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag}")
    if ch > 1:
        x = x[: x.size // ch * ch].reshape(-1, ch).mean(axis=1).astype(np.float32)  # librosa to_mono
    return np.ascontiguousarray(x, dtype=np.float32), int(sr)


def resample_design(up: int, down: int):
    """Taps and trimming of ``scipy.signal.resample_poly(x, up, down)`` (window=("kaiser", 5.0)):
    returns (taps float32 — scaled by ``up`` and left-padded so the output aligns —, n_pre_remove)."""
    from scipy.signal import firwin
    max_rate = max(up, down)
    half_len = 10 * max_rate
    h = firwin(2 * half_len + 1, 1.0 / max_rate, window=("kaiser", 5.0)) * up
    n_pre_pad = down - half_len % down
    n_pre_remove = (half_len + n_pre_pad) // down
    return np.concatenate([np.zeros(n_pre_pad), h]).astype(np.float32), int(n_pre_remove)


def resample_batch(signals, sr_in: int, sr_out: int):
    """Polyphase resampling of a list of float32 signals on the GPU → list of float32 arrays."""
    import torch
    fr = Fraction(sr_out, sr_in)
    up, down = fr.numerator, fr.denominator
    if up == down:
        return [np.asarray(s, dtype=np.float32) for s in signals]
    lib, dev = _lib.load(), _lib.require_gpu()
    taps, n_pre_remove = resample_design(up, down)
    n_in = np.asarray([len(s) for s in signals], dtype=np.int64)
    n_out = -(-n_in * up // down)
    io, oo = np.zeros(len(signals) + 1, np.int64), np.zeros(len(signals) + 1, np.int64)
    np.cumsum(n_in, out=io[1:])
    np.cumsum(n_out, out=oo[1:])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    x = t(np.concatenate([np.asarray(s, dtype=np.float32) for s in signals]) if len(signals) else np.zeros(0, np.float32))
    y = torch.empty(int(oo[-1]), dtype=torch.float32, device=dev)
    out = []
    for lo in range(0, len(signals), 65535):  # grid.y limit
        hi = min(lo + 65535, len(signals))
        tio, too = t(io[lo:hi + 1]), t(oo[lo:hi + 1])
        _lib.check(lib.sapr_resample_poly(_lib.ptr(x), _lib.ptr(tio), _lib.ptr(too), hi - lo,
                                          int(n_out[lo:hi].max()) if hi > lo else 0, up, down, _lib.ptr(t(taps)),
                                          len(taps), n_pre_remove, _lib.ptr(y), _lib.current_stream()),
                   "sapr_resample_poly")
    host = y.cpu().numpy()
    for u in range(len(signals)):
        out.append(host[oo[u]:oo[u + 1]].copy())
    return out


def load_audio(path: str, sr: int = TARGET_SR, raw16: bool = False):
    """``librosa.load(path)`` stand-in: mono float32 at ``sr`` Hz (resampled on the GPU if needed).
    ``raw16=True`` hands back the int16 samples of a mono 16-bit file that needs no resampling."""
    y, sr_in = read_wav(path, raw16=raw16)
    if sr_in != sr:
        if y.dtype == np.int16:
            y = y.astype(np.float32) / 32768.0
        y = resample_batch([y], sr_in, sr)[0]
    return y, sr


def _plan_for(n_frames: int):
    """Reference-preset plan whose LDS log-mel matrix holds at least ``n_frames`` frames."""
    from .frontend import REFERENCE, MfccPlan
    cap = max(32, -(-n_frames // 16) * 16)
    if cap not in _plans:
        _plans[cap] = MfccPlan(**REFERENCE, max_frames=cap)
    return _plans[cap]


def extract_mfcc(audio_path: str) -> np.ndarray:
    """13 MFCCs, 30 ms Hamming windows every 10 ms at 22 050 Hz → ``(13, T)`` float32."""
    try:
        from .frontend import mfcc_batch
        y, sr = load_audio(audio_path)
        hop = int(0.01 * sr)
        return mfcc_batch([y], _plan_for(1 + len(y) // hop))[0]
    except Exception as e:
        logging.error(f"Error processing {audio_path}: {str(e)}")
        raise


class AudioDecodeUnavailable(RuntimeError):
    """The folder holds compressed audio this build cannot decode (no mp3 decoder in the image)."""


def extract_mfccs(input_folder: str, output_folder: str) -> str:
    """Every ``.wav`` file of ``input_folder`` → ``<stem>.npy``; per-file failures are logged and skipped
    (``mfcc_extract.py:47-49``).  The whole folder is one kernel launch.

    The reference lists ``.mp3`` files (``mfcc_extract.py:36``) and decodes them through librosa/audioread.
    There is no mp3 decoder here (and none may be installed), so a folder that holds mp3 files is refused
    LOUDLY instead of being "processed" into an empty feature directory: transcode to 16-bit WAV first
    (any sample rate; resampling to 22 050 Hz happens on the GPU)."""
    logging.debug(f"Extracting MFCCs from {input_folder} to {output_folder}...")
    listing = os.listdir(input_folder)
    mp3 = [f for f in listing if f.endswith(".mp3")]
    if mp3:
        raise AudioDecodeUnavailable(
            f"{input_folder} holds {len(mp3)} .mp3 file(s) (e.g. {mp3[0]}): sapr_amd reads PCM WAV only; "
            "transcode the set to .wav (the MFCC path is unchanged from PCM onwards)")
    os.makedirs(output_folder, exist_ok=True)
    names, signals = [], []
    for file in listing:
        if file.endswith(".wav"):
            try:
                y, _ = load_audio(os.path.join(input_folder, file), raw16=True)
                names.append(file)
                signals.append(y)
            except Exception as e:
                logging.error(f"Failed to process {file}: {str(e)}")
                continue
    processed_files = 0
    if signals:
        from .frontend import mfcc_batch
        if not all(y.dtype == np.int16 for y in signals):  # mixed folder: one sample format for the batch
            signals = [y.astype(np.float32) / 32768.0 if y.dtype == np.int16 else y for y in signals]
        hop = int(0.01 * TARGET_SR)
        feats = mfcc_batch(signals, _plan_for(max(1 + len(y) // hop for y in signals)))
        for file, m in zip(names, feats):
            np.save(os.path.join(output_folder, os.path.splitext(file)[0] + ".npy"), m)
            processed_files += 1
            logging.debug(f"Processed {file} ({processed_files} files done)")
    logging.info(f"Completed processing {processed_files} files")
    return output_folder


def load_mfcc(file_path: str) -> np.ndarray:
    try:
        return np.load(file_path)
    except Exception as e:
        logging.error(f"Failed to load MFCC from {file_path}: {str(e)}")
        raise


def load_mfccs(directory_path: str) -> list:
    """All ``.npy`` feature arrays of a directory, in ``os.listdir`` order (``mfcc_extract.py:63-79``)."""
    return [load_mfcc(os.path.join(directory_path, n)) for n in os.listdir(directory_path) if n.endswith(".npy")]


def load_mfccs_by_word(directory_path: str, word: str) -> list:
    """Arrays whose file name ends in ``_<word>.npy`` (``mfcc_extract.py:82-89``)."""
    out = []
    for n in os.listdir(directory_path):
        if n.endswith(".npy") and n.split("_")[-1].split(".")[0] == word:
            out.append(load_mfcc(os.path.join(directory_path, n)))
    return out


if __name__ == "__main__":
    extract_mfccs("dev_set", "feature_set")
    extract_mfccs("eval_set", "eval_feature_set")
