"""``evaluate`` command: the reference's ``python main.py evaluate`` (``vad/evaluate.py:20-190``) on the MI355X
path.  Same inputs (a JSON-lines data list of ``{"audio_path", "voice_activity_path"}`` pairs, paths relative to
``data_dir``), same per-file and total metrics under the same keys, same output file layout (first line = totals,
then one line per file).  Metrics are numpy-only restatements (``metrics.py``); sklearn is not required.

Reference quirk kept: the un-prefixed ``auc / accuracy / precision / recall`` are computed from the BOOSTED
probabilities exactly like their ``boosted_*`` twins (``vad/evaluate.py:65-68`` vs ``:72-75``); only
``vacc / sba / eba / bp / eer`` use the single-frame (middle window slot) predictions.
"""
from __future__ import annotations

import json
import random
from collections import OrderedDict
from pathlib import Path
from typing import Callable, List, Optional

import numpy as np

from .data_models import VoiceActivity
from .metrics import equal_error_rate, precision_recall, roc_auc, vad_accuracy

METRIC_KEYS = ("auc", "accuracy", "precision", "recall", "vacc", "sba", "eba", "bp", "eer")


def load_data_list(path: Path) -> List[dict]:
    """vad/data_models/vad_data.py:38-46: one JSON object per line."""
    pairs = []
    with Path(path).open() as fh:
        for line in fh:
            if line.strip():
                d = json.loads(line)
                pairs.append({"audio_path": Path(d["audio_path"]), "voice_activity_path": Path(d["voice_activity_path"])})
    return pairs


def file_metrics(true_labels, all_frame_probabilities, threshold: float) -> "OrderedDict[str, float]":
    """vad/evaluate.py:56-80 for one file: [N, W] probabilities + label vector -> the 18 metric values."""
    p = np.asarray(all_frame_probabilities)
    y = np.asarray(true_labels)
    single = p[:, int(p.shape[1] / 2)][: len(y)]
    single_pred = single > threshold
    boosted = p.mean(axis=1)[: len(y)]
    boosted_pred = boosted > threshold
    y = y[: len(boosted)]  # (the reference would raise in sklearn on a length mismatch; labels are trimmed instead)
    auc = roc_auc(y, boosted)
    accuracy = float((y.astype(bool) == boosted_pred).mean())
    precision, recall = precision_recall(y, boosted_pred)
    vacc, _, sba, eba, bp = vad_accuracy(y, single_pred)
    eer = equal_error_rate(y, single_pred)
    bvacc, _, bsba, beba, bbp = vad_accuracy(y, boosted_pred)
    beer = equal_error_rate(y, boosted_pred)
    out = OrderedDict(auc=auc, accuracy=accuracy, precision=precision, recall=recall, vacc=vacc, sba=sba, eba=eba, bp=bp, eer=eer)
    out.update(boosted_auc=auc, boosted_accuracy=accuracy, boosted_precision=precision, boosted_recall=recall,
               boosted_vacc=bvacc, boosted_sba=bsba, boosted_eba=beba, boosted_bp=bbp, boosted_eer=beer)
    return out


def _report(title: str, m: dict) -> str:
    names = [("AUC", "auc"), ("Accuracy", "accuracy"), ("Precision", "precision"), ("Recall", "recall"), ("VACC", "vacc"),
             ("SBA", "sba"), ("EBA", "eba"), ("BP", "bp"), ("EER", "eer")]
    lines = ["", title]
    lines += [f"{label}: {m[key]:0.2%}" for label, key in names]
    lines += [f"Boosted {label}: {m['boosted_' + key]:0.2%}" for label, key in names]
    return "\n".join(lines) + "\n"


def evaluate_vad_from_scratch(eval_path: Path, checkpoint_path: Optional[Path] = None, output_path: Optional[Path] = None,
                              data_dir: Optional[Path] = None, threshold: float = 0.5, shuffle: bool = False,
                              limit: Optional[int] = None, random_seed: int = 0, device: str = "cuda",
                              probabilities_fn: Optional[Callable[[Path], np.ndarray]] = None, echo=print) -> dict:
    """Arguments as vad/evaluate.py:20-29.  `probabilities_fn(audio_path) -> [N, W]` replaces checkpoint + GPU
    predictor (host-logic tests); otherwise the audio goes WAV -> log-mel -> predict_probabilities on `device`."""
    eval_path = Path(eval_path)
    if probabilities_fn is None:
        from .features import load_wav_mono16k, log_mel
        from .predictor import VADFromScratchPredictor

        predictor = VADFromScratchPredictor.from_checkpoint(checkpoint_path, device)

        def probabilities_fn(path):
            return predictor.predict_probabilities(log_mel(load_wav_mono16k(path), predictor.device))

    data_dir = eval_path.parent if data_dir is None else Path(data_dir)
    pairs = load_data_list(eval_path)
    if shuffle:
        random.seed(random_seed)
        random.shuffle(pairs)
    if limit:
        pairs = pairs[:limit]

    results = []
    for pair in pairs:
        audio_path = data_dir.joinpath(pair["audio_path"])
        voice_activity_path = data_dir.joinpath(pair["voice_activity_path"])
        true_labels = VoiceActivity.load(voice_activity_path).to_labels(100)
        metrics = file_metrics(true_labels, probabilities_fn(audio_path), threshold)
        echo(_report(str(pair["audio_path"]), metrics))
        result = OrderedDict(audio_path=str(audio_path), voice_activity_path=str(voice_activity_path))
        result.update(metrics)
        results.append(result)

    total = {key: float(np.mean([r[key] for r in results])) for key in list(METRIC_KEYS) + ["boosted_" + k for k in METRIC_KEYS]}
    echo(_report("Total:", total))
    if output_path is not None:
        output_path = Path(output_path)
        output_path.parent.mkdir(parents=True, exist_ok=True)
        with output_path.open("w") as fh:
            fh.write(json.dumps(total, ensure_ascii=False) + "\n")
            for r in results:
                fh.write(json.dumps(r, ensure_ascii=False) + "\n")
    return {"total": total, "files": results}
