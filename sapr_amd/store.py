"""Packed ragged feature store.

The reference keeps one ``<speaker>_<word>.npy`` of shape ``(13, T)`` per utterance
(``mfcc_extract.py:51,63-89``) and re-reads the directory for every word model.  The kernels want
the layout of ``FeatureBatch``: all frames frame-major in one ``[total_frames, D]`` float32 block
plus an offset table.  This module writes and memory-maps exactly that, so a training or decoding
job uploads one contiguous buffer (a single host→HBM copy) instead of thousands of small files:

    <store>/frames.npy    float32 [total_frames, D]   (memory-mapped)
    <store>/offsets.npy   int64   [N+1]
    <store>/names.txt     one utterance name per line (file stem, e.g. ``sp01_heed``)

Word selection follows the reference's file-name rule: the text after the last ``_``
(``mfcc_extract.py:86``).
"""
from __future__ import annotations

import os

import numpy as np


class FeatureStore:
    def __init__(self, frames, offsets, names):
        self.frames, self.offsets, self.names = frames, np.asarray(offsets, dtype=np.int64), list(names)
        if self.offsets.ndim != 1 or self.offsets.size != len(self.names) + 1 or self.offsets[0] != 0:
            raise ValueError("offsets must be [N+1], start at 0 and match names")
        if np.any(np.diff(self.offsets) < 0) or self.offsets[-1] != self.frames.shape[0]:
            raise ValueError("offsets must be non-decreasing and end at the number of frames")

    # ---- writing ------------------------------------------------------------------------
    @staticmethod
    def write(path: str, utterances, names, layout: str = "DT") -> "FeatureStore":
        """``utterances``: list of ``(D, T)`` arrays (``layout="DT"``, the reference's) or ``(T, D)``."""
        if len(utterances) != len(names):
            raise ValueError("one name per utterance")
        if not utterances:
            raise ValueError("empty utterance list")
        mats = [np.asarray(u, dtype=np.float32).T if layout == "DT" else np.asarray(u, dtype=np.float32) for u in utterances]
        D = mats[0].shape[1]
        if any(m.ndim != 2 or m.shape[1] != D for m in mats):
            raise ValueError("all utterances must share the feature dimension")
        offsets = np.zeros(len(mats) + 1, np.int64)
        np.cumsum([m.shape[0] for m in mats], out=offsets[1:])
        os.makedirs(path, exist_ok=True)
        frames = np.lib.format.open_memmap(os.path.join(path, "frames.npy"), mode="w+", dtype=np.float32,
                                           shape=(int(offsets[-1]), D))
        for m, lo in zip(mats, offsets[:-1]):
            frames[lo:lo + m.shape[0]] = m
        frames.flush()
        np.save(os.path.join(path, "offsets.npy"), offsets)
        with open(os.path.join(path, "names.txt"), "w") as f:
            f.write("".join(n + "\n" for n in names))
        return FeatureStore.open(path)

    @staticmethod
    def pack_directory(feature_dir: str, path: str) -> "FeatureStore":
        """Every ``.npy`` of a reference-style feature directory, in sorted name order."""
        files = sorted(n for n in os.listdir(feature_dir) if n.endswith(".npy"))
        return FeatureStore.write(path, [np.load(os.path.join(feature_dir, n)) for n in files],
                                  [os.path.splitext(n)[0] for n in files])

    # ---- reading ------------------------------------------------------------------------
    @staticmethod
    def open(path: str) -> "FeatureStore":
        # copy-on-write mapping: torch wants a writable buffer; the file itself is never modified
        frames = np.load(os.path.join(path, "frames.npy"), mmap_mode="c")
        offsets = np.load(os.path.join(path, "offsets.npy"))
        with open(os.path.join(path, "names.txt")) as f:
            names = [ln.rstrip("\n") for ln in f]
        return FeatureStore(frames, offsets, names)

    def __len__(self):
        return len(self.names)

    @property
    def lengths(self) -> np.ndarray:
        return np.diff(self.offsets)

    def utterance(self, i: int) -> np.ndarray:
        """``(D, T)`` like ``load_mfcc`` returns (a transposed view of the mapped block)."""
        return self.frames[self.offsets[i]:self.offsets[i + 1]].T

    def word_of(self, i: int) -> str:
        return self.names[i].split("_")[-1].split(".")[0]

    def indices_for_word(self, word: str) -> np.ndarray:
        return np.asarray([i for i in range(len(self)) if self.word_of(i) == word], dtype=np.int64)

    def select(self, indices) -> "FeatureStore":
        """In-memory sub-store (e.g. one word's training set, or one rank's shard)."""
        idx = np.asarray(indices, dtype=np.int64)
        lens = self.lengths[idx]
        offs = np.zeros(idx.size + 1, np.int64)
        np.cumsum(lens, out=offs[1:])
        frames = np.empty((int(offs[-1]), self.frames.shape[1]), np.float32)
        for k, i in enumerate(idx):
            frames[offs[k]:offs[k + 1]] = self.frames[self.offsets[i]:self.offsets[i + 1]]
        return FeatureStore(frames, offs, [self.names[i] for i in idx])

    def shard(self, rank: int, world: int) -> "FeatureStore":
        """Contiguous utterance range of one rank (same split as ``dist.shard_range``)."""
        from .dist import shard_range
        lo, hi = shard_range(len(self), rank, world)
        return FeatureStore(self.frames[self.offsets[lo]:self.offsets[hi]], self.offsets[lo:hi + 1] - self.offsets[lo],
                            self.names[lo:hi])

    def training_data(self, words):
        """``[(X_w, lengths_w) for w in words]`` — the ``data`` argument of ``hmmlearn_hmm.fit_models``
        (X_w = this store's frames of word w, frame-major, like ``HMMLearnModel.prepare_data``)."""
        out = []
        for w in words:
            sub = self.select(self.indices_for_word(w))
            out.append((sub.frames, sub.lengths))
        return out

    def to_batch(self, device=None):
        """One host→HBM copy of the frame block → ``trellis.FeatureBatch``."""
        import torch
        from . import _lib
        from .trellis import FeatureBatch
        device = device or _lib.require_gpu()
        host = torch.from_numpy(np.ascontiguousarray(self.frames))
        return FeatureBatch.from_packed(host.to(device), self.lengths)
