"""Frame-level ROC AUC, the accuracy metric of the reference (``roc_auc_score(true_labels,
boosted_probs)`` at ``vad/evaluate.py:65``, mean over files at ``:134``).  Rank-based
(Mann-Whitney U with mid-ranks for ties), numpy only -- sklearn is not assumed on the GPU box."""
from __future__ import annotations

import numpy as np


def roc_auc(labels, scores) -> float:
    labels = np.asarray(labels).astype(bool).ravel()
    scores = np.asarray(scores, dtype=np.float64).ravel()
    n_pos = int(labels.sum())
    n_neg = labels.size - n_pos
    if n_pos == 0 or n_neg == 0:
        raise ValueError("AUC needs both classes")
    order = np.argsort(scores, kind="mergesort")
    s = scores[order]
    ranks = np.empty(labels.size, dtype=np.float64)
    # mid-ranks for ties
    boundaries = np.flatnonzero(np.r_[True, s[1:] != s[:-1], True])
    for lo, hi in zip(boundaries[:-1], boundaries[1:]):
        ranks[order[lo:hi]] = 0.5 * (lo + hi - 1) + 1.0
    return float((ranks[labels].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))
