#!/usr/bin/env python3
"""Golden values for the evaluate row: the reference's vad.metrics (vad_accuracy, equal_error_rate) and the sklearn
calls of vad/evaluate.py:65-80, run here on deterministic inputs (tests/golden/metric_cases.py).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_metrics.py
"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from metric_cases import metric_case  # noqa: E402
from sklearn.metrics import accuracy_score, precision_score, recall_score, roc_auc_score  # noqa: E402
from vad.metrics import equal_error_rate, vad_accuracy  # noqa: E402  (reference, unmodified)

cases = []
for c in range(12):
    y, P = metric_case(c)
    single, boosted = P[:, 3] > 0.5, P.mean(axis=1) > 0.5
    cases.append({"seed": c, "auc": float(roc_auc_score(y, P.mean(axis=1))), "accuracy": float(accuracy_score(y, boosted)),
                  "precision": float(precision_score(y, boosted)), "recall": float(recall_score(y, boosted)),
                  "vad_accuracy_single": [float(v) for v in vad_accuracy(y, single)],
                  "vad_accuracy_boosted": [float(v) for v in vad_accuracy(y, boosted)],
                  "eer_single": float(equal_error_rate(y, single)), "eer_boosted": float(equal_error_rate(y, boosted)),
                  "eer_scores": float(equal_error_rate(y, P.mean(axis=1)))})
out = Path(__file__).resolve().parent / "golden_metrics.json"
out.write_text(json.dumps(cases, indent=0))
print("wrote", out, len(cases))
