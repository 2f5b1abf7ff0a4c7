"""Host-side mirror of the metric part of ``assignment2/eval.py``: the same function
names, arguments and return dictionary, driven by the batched ``sapr_amd.decoder.Decoder``; the confusion
matrix and the accuracy are computed with numpy in scikit-learn's conventions (``eval.py:34-35``), the
matplotlib/seaborn plot of ``eval.py:40-95`` is out of scope.
"""
from __future__ import annotations

import logging
from typing import Dict, List, Tuple

import numpy as np

from .decoder import Decoder


def extract_labels(all_results: Dict) -> Tuple[List[str], List[str]]:
    """``eval.py:16-25``: true / predicted words in vocabulary order, then sample order."""
    true_labels, predicted_labels = [], []
    for results in all_results.values():
        for result in results:
            true_labels.append(result["true_word"])
            predicted_labels.append(result["predicted_word"])
    return true_labels, predicted_labels


def confusion_matrix(y_true, y_pred) -> np.ndarray:
    """``sklearn.metrics.confusion_matrix(y_true, y_pred)`` without ``labels=``: rows / columns are the
    SORTED DISTINCT labels that occur in either list (so a word nobody says or predicts has no row)."""
    y_true, y_pred = np.asarray(y_true), np.asarray(y_pred)
    labels = np.unique(np.concatenate([y_true, y_pred])) if y_true.size else np.zeros(0, dtype=np.int64)
    idx = {v: i for i, v in enumerate(labels.tolist())}
    cm = np.zeros((len(labels), len(labels)), dtype=np.int64)
    for t, p in zip(y_true.tolist(), y_pred.tolist()):
        cm[idx[t], idx[p]] += 1
    return cm


def calculate_metrics(true_labels: List[str], predicted_labels: List[str], vocab: List[str]) -> Tuple[np.ndarray, float]:
    """``eval.py:28-38``: labels → vocabulary indices, confusion matrix, accuracy."""
    label_mapping = {word: idx for idx, word in enumerate(vocab)}
    true_idx = [label_mapping[label] for label in true_labels]          # KeyError for an unknown word, like the reference
    pred_idx = [label_mapping[label] for label in predicted_labels]
    cm = confusion_matrix(true_idx, pred_idx)
    accuracy = float(np.mean(np.asarray(true_idx) == np.asarray(pred_idx))) if true_idx else float("nan")
    return cm, accuracy


def log_per_word_accuracy(all_results: Dict) -> None:
    logging.info("\nPer-word accuracy:")
    for word, results in all_results.items():
        word_correct = sum(r["correct"] for r in results)
        word_total = len(results)
        logging.info(f"{word}: {word_correct / word_total:.2%}")


def eval_hmm(implementation: str = "hmmlearn", feature_set_path: str = "eval_feature_set", model_iter: int = 15) -> Dict:
    """``eval.py:108-134`` without the plot: every utterance of every word decoded in batched launches."""
    import pandas as pd
    decoder = Decoder(implementation=implementation, n_iter=model_iter)
    all_results = decoder.decode_vocabulary(feature_set_path, verbose=False)
    true_labels, predicted_labels = extract_labels(all_results)
    cm, accuracy = calculate_metrics(true_labels, predicted_labels, decoder.vocab)
    cm_df = pd.DataFrame(cm, index=decoder.vocab, columns=decoder.vocab)  # raises like the reference if a word never occurs
    logging.info(f"\nConfusion Matrix:\n{cm_df}")
    logging.info(f"\nOverall Accuracy: {accuracy:.2%}")
    log_per_word_accuracy(all_results)
    return {"results": all_results, "accuracy": accuracy, "confusion_matrix": cm_df, "true_labels": true_labels,
            "predicted_labels": predicted_labels}
