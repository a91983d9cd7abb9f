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


# ---- the rest of the evaluate harness (vad/evaluate.py:48-80, vad/metrics.py:16-131), numpy only -----------------
def detect_boundaries(frames):
    """vad/metrics.py:117-126: start / end frame indices of the voiced segments."""
    frames = np.asarray(frames).astype(np.int64)
    boundaries = np.append(frames, 0) - np.append(0, frames)
    starts = np.where(boundaries == 1)[0]
    ends = np.where(boundaries == -1)[0] - 1
    return starts, ends, len(starts)


def _boundary_accuracy(frames_true, frames_pred, boundaries, num_segments, L, start: bool) -> float:
    # vad/metrics.py:58-104: weight 1 on the frames at/after a start boundary (at/before an end boundary) inside +-L
    n = len(frames_true)
    total = 0.0
    for bnd in boundaries:
        lo, hi = max(int(bnd) - L, 0), min(int(bnd) + L, n)
        num = den = 0
        for idx in range(lo, hi):
            wgt = 1 if ((idx - bnd) if start else (bnd - idx)) >= 0 else 0
            num += wgt * (1 if frames_pred[idx] == frames_true[idx] else 0)
            den += wgt
        total += num / den
    return total / num_segments if num_segments > 0 else 0


def vad_accuracy(frames_true, frames_pred, L: int = 5):
    """vad/metrics.py:23-55 -> (VACC, ACC, SBA, EBA, BP); VACC = harmonic mean of the other four."""
    from statistics import harmonic_mean

    frames_true = np.asarray(frames_true).astype(np.int64)
    frames_pred = np.asarray(frames_pred).astype(np.int64)
    acc = float((frames_true == frames_pred).mean())
    starts, ends, n_true = detect_boundaries(frames_true)
    _, _, n_pred = detect_boundaries(frames_pred)
    sba = _boundary_accuracy(frames_true, frames_pred, starts, n_true, L, True)
    eba = _boundary_accuracy(frames_true, frames_pred, ends, n_true, L, False)
    bp = n_true / (2 * n_pred) * (sba + eba) if n_pred > 0 else 0
    return harmonic_mean([acc, sba, eba, bp]), acc, sba, eba, bp


def roc_curve_points(labels, scores):
    """(fpr, tpr) at every distinct threshold, starting at (0, 0) -- sklearn.metrics.roc_curve without
    drop_intermediate (dropping collinear points does not change a piecewise-linear interpolation)."""
    labels = np.asarray(labels).astype(bool).ravel()
    scores = np.asarray(scores, dtype=np.float64).ravel()
    order = np.argsort(-scores, kind="mergesort")
    s, y = scores[order], labels[order]
    distinct = np.r_[np.flatnonzero(np.diff(s)), len(s) - 1]
    tps = np.cumsum(y)[distinct]
    fps = 1 + distinct - tps
    tpr = np.r_[0.0, tps / max(tps[-1], 1)]
    fpr = np.r_[0.0, fps / max(fps[-1], 1)]
    return fpr, tpr


def equal_error_rate(labels, scores) -> float:
    """vad/metrics.py:16-20: the x in [0,1] where 1 - x = TPR(x) on the linearly interpolated ROC."""
    fpr, tpr = roc_curve_points(labels, scores)
    f = lambda x: 1.0 - x - np.interp(x, fpr, tpr)  # noqa: E731
    lo, hi = 0.0, 1.0
    for _ in range(200):  # bisection (the reference uses scipy.optimize.brentq; same root to 1e-12)
        mid = 0.5 * (lo + hi)
        if f(mid) > 0:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def precision_recall(labels, predictions):
    labels = np.asarray(labels).astype(bool)
    predictions = np.asarray(predictions).astype(bool)
    tp = float((labels & predictions).sum())
    return (tp / predictions.sum() if predictions.sum() else 0.0), (tp / labels.sum() if labels.sum() else 0.0)
