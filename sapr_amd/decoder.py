"""Host-side mirror of ``assignment2/decoder.py``: same ``Decoder`` API and result dictionaries,
but every utterance is scored against every word model in ONE kernel launch sequence instead of W Python
calls to ``model.decode`` per utterance (``decoder.py:42-47``).  ``decode_sequence`` returns the best word, ITS
score and ITS states (``decoder.py:35-49``) — the other W - 1 exact scores are thrown away — so the hmmlearn
models go through ``sapr_viterbi_decode_pruned`` (bounding pass over the vocabulary, exact lattice only for the
words that can still win; same bits as scoring every word) whenever the model pack allows it, and through
``sapr_viterbi_diag_scores`` + ``sapr_viterbi_backtrace`` otherwise; the reference's from-scratch models use
``sapr_custom_decode``.
"""
from __future__ import annotations

import logging
import pickle
from pathlib import Path
from typing import Dict, List, Tuple

import numpy as np

from . import _lib
from .mfcc_extract import load_mfccs_by_word


class Decoder:
    def __init__(self, models_dir: str = "trained_models", implementation: str = "hmmlearn", n_iter: int = 15):
        self.models_dir = Path(models_dir)
        self.implementation = implementation
        self.n_iter = n_iter
        self.models: Dict = {}
        self.vocab: List[str] = []
        self._pack = None
        self.load_models()

    def load_models(self) -> None:
        impl_dir = self.models_dir / self.implementation
        pattern = f"*_{self.implementation}_{self.n_iter}.pkl"
        for model_path in impl_dir.glob(pattern):  # vocabulary order = glob order, like the reference
            word = model_path.stem.split("_")[0]
            with open(model_path, "rb") as f:
                self.models[word] = pickle.load(f)
                self.vocab.append(word)
        if not self.models:
            raise ValueError(f"No models found in {impl_dir} with pattern {pattern}")
        logging.info(f"Loaded {len(self.models)} models from {impl_dir} for words: {', '.join(self.vocab)}")

    # ---- batched core -------------------------------------------------------------------------
    def _model_list(self):
        return [self.models[w] for w in self.models]  # dict order = load order (decoder.py:42)

    def decode_batch(self, feature_list: List[np.ndarray]) -> List[Tuple[str, float, object]]:
        """``feature_list``: (D, T) arrays as stored by mfcc_extract (the ``feat_seq`` of
        decoder.py:58).  Returns one ``(word, score, states)`` per utterance with decode_sequence's
        semantics: first strict maximum over the models in load order."""
        words = list(self.models)
        if self.implementation == "custom":
            return self._decode_custom(feature_list)
        from .trellis import FeatureBatch
        return self._decode_feature_batch(FeatureBatch.from_arrays(feature_list, layout="DT"))

    def decode_store(self, store) -> List[Tuple[str, float, object]]:
        """Every utterance of a packed ``store.FeatureStore`` (one host→HBM copy, one launch
        sequence); same tuples as ``decode_batch``."""
        if self.implementation == "custom":
            return self._decode_custom(store.to_batch())
        return self._decode_feature_batch(store.to_batch())

    def _decode_custom(self, features) -> List[Tuple[str, float, object]]:
        """The reference's from-scratch models: emission rows, trellis and the arg-max over the models
        (decoder.py:42-47, first strict maximum in load order) all on the device."""
        from .custom_hmm import decode_batch
        words = list(self.models)
        _, _, bw, bs, bp = decode_batch(self._model_list(), features, with_best=True)
        return [(words[w], float(sc), [int(x) for x in p]) if w >= 0 else (None, float("-inf"), None)
                for w, sc, p in zip(bw, bs, bp)]

    def _decode_feature_batch(self, batch) -> List[Tuple[str, float, object]]:
        from .trellis import DiagModelPack, viterbi_decode_best
        words = list(self.models)
        if self._pack is None:
            self._pack = DiagModelPack.from_models(self._model_list())
        tie = _lib.TIE_HIGH if getattr(self._model_list()[0], "tie_break", "high") == "high" else _lib.TIE_LOW
        # decoder.py:59 hands hmmlearn the transposed VIEW of the (D,T) array → numpy's left-to-right sum.
        # Pruned decoder when the pack is prunable, all-vocabulary evaluation otherwise: identical outputs.
        best_word, best_score, best_path = viterbi_decode_best(batch, self._pack, tie=tie, sum_order=_lib.SUM_TVIEW)
        bw, bs, path = _lib.to_host(best_word, best_score, best_path)
        offs = np.r_[0, np.cumsum(batch.lengths)].tolist()
        path = path.astype(np.int64)  # one conversion for the batch; the per-utterance results are views of it
        bw_l, bs_l = bw.tolist(), bs.tolist()
        return [(words[w], sc, path[lo:hi]) if w >= 0 else (None, float("-inf"), None)
                for w, sc, lo, hi in zip(bw_l, bs_l, offs[:-1], offs[1:])]

    # ---- the reference's API ------------------------------------------------------------------
    def decode_sequence(self, features: np.ndarray) -> Tuple[str, float, List[int]]:
        """``features`` is the (T, D) view decoder.py:59 builds."""
        return self.decode_batch([np.asarray(features).T])[0]

    @staticmethod
    def _rows(word: str, decoded) -> List[Dict]:
        """Result dictionaries of one word's samples (schema of decoder.py:62-69)."""
        return [{"sample_index": i, "true_word": word, "predicted_word": pred, "log_likelihood": score,
                 "correct": pred == word, "state_sequence": states}
                for i, (pred, score, states) in enumerate(decoded, start=1)]

    def decode_word_samples(self, word: str, feature_set: str = "feature_set") -> List[Dict]:
        if word not in self.vocab:
            raise ValueError(f"Word '{word}' not in vocabulary: {self.vocab}")
        features = load_mfccs_by_word(feature_set, word)
        return self._rows(word, self.decode_batch(features) if features else [])

    def decode_vocabulary(self, feature_set: str = "feature_set", verbose: bool = True) -> Dict[str, List[Dict]]:
        """Every word's samples (decoder.py:74-93) — loaded word by word like the reference, decoded as ONE
        batch (one launch sequence for the whole vocabulary), reported word by word."""
        per_word = [load_mfccs_by_word(feature_set, word) for word in self.vocab]
        flat = [f for feats in per_word for f in feats]
        decoded = self.decode_batch(flat) if flat else []
        all_results, start = {}, 0
        for word, feats in zip(self.vocab, per_word):
            rows = self._rows(word, decoded[start:start + len(feats)])
            start += len(feats)
            all_results[word] = rows
            if verbose:
                hits = sum(1 for r in rows if r["correct"])
                print(f"\nResults for '{word}':")
                print(f"Accuracy: {hits}/{len(rows)} ({hits/len(rows):.1%})")
                for r in rows:
                    mark = "✓" if r["correct"] else "✗"
                    print(f"\nSample {r['sample_index']}:\nPredicted: {r['predicted_word']}\n"
                          f"Log likelihood: {r['log_likelihood']:.2f}\nCorrect: {mark}")
        return all_results


if __name__ == "__main__":
    decoder = Decoder(implementation="hmmlearn", n_iter=15)
    results = decoder.decode_vocabulary()
    all_predictions = [result["correct"] for word_results in results.values() for result in word_results]
    print(f"\nOverall accuracy: {sum(all_predictions) / len(all_predictions):.1%}")
