"""``from eval import eval_hmm`` → the drop-in (metrics only; the reference's plots need seaborn)."""
from sapr_amd.eval import *  # noqa: F401,F403
from sapr_amd.eval import calculate_metrics, confusion_matrix, eval_hmm, extract_labels, log_per_word_accuracy  # noqa: F401
