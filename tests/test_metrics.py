"""Evaluate row: voice_activity_detection_amd.metrics against values produced by the reference's
vad.metrics + the sklearn calls of vad/evaluate.py:65-80 (tests/golden/make_golden_metrics.py)."""
import json
import sys
from datetime import timedelta
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE / "golden"))
from metric_cases import metric_case  # noqa: E402

G = json.loads((HERE / "golden" / "golden_metrics.json").read_text())


def test_evaluate_file_matches_reference_metrics():
    from voice_activity_detection_amd.evaluate import file_metrics
    from voice_activity_detection_amd.metrics import equal_error_rate, vad_accuracy

    for c in G:
        y, P = metric_case(c["seed"])
        r = dict(file_metrics(y, P, 0.5))
        r["acc"] = vad_accuracy(y, P[:, int(P.shape[1] / 2)] > 0.5)[1]
        r["boosted_acc"] = vad_accuracy(y, P.mean(axis=1) > 0.5)[1]
        for k in ("auc", "accuracy", "precision", "recall"):
            assert abs(r[k] - c[k]) < 1e-12, (k, r[k], c[k])
        got_s = [r["vacc"], r["acc"], r["sba"], r["eba"], r["bp"]]
        got_b = [r["boosted_vacc"], r["boosted_acc"], r["boosted_sba"], r["boosted_eba"], r["boosted_bp"]]
        assert np.allclose(got_s, c["vad_accuracy_single"], atol=1e-12) and np.allclose(got_b, c["vad_accuracy_boosted"], atol=1e-12)
        assert abs(r["eer"] - c["eer_single"]) < 1e-9 and abs(r["boosted_eer"] - c["eer_boosted"]) < 1e-9
        assert abs(equal_error_rate(y, P.mean(axis=1)) - c["eer_scores"]) < 1e-9


def test_to_labels():
    from voice_activity_detection_amd.data_models import Activity, VoiceActivity

    va = VoiceActivity(timedelta(seconds=2.0), [Activity(timedelta(seconds=0.25), timedelta(seconds=0.5)),
                                                Activity(timedelta(seconds=1.0), timedelta(seconds=1.995))], None, None)
    lab = va.to_labels(100)  # vad/data_models/voice_activity.py:239-246
    assert lab.shape == (200,) and lab.sum() == 25 + 99 and lab[25] == 1 and lab[24] == 0 and lab[50] == 0 and lab[198] == 1 and lab[199] == 0


def _labels_to_voice_activity(y):
    from voice_activity_detection_amd.data_models import Activity, VoiceActivity

    acts, start = [], None
    for i, v in enumerate(list(y) + [0]):
        if v and start is None:
            start = i
        if not v and start is not None:
            # +5 ms: to_labels truncates seconds * 100 (voice_activity.py:239-246), and 0.29 * 100 < 29 in binary floating point
            acts.append(Activity(timedelta(milliseconds=10 * start + 5), timedelta(milliseconds=10 * i + 5)))
            start = None
    return VoiceActivity(timedelta(milliseconds=10 * len(y) + 5), acts, None, None)


def test_evaluate_command_host_logic(tmp_path):
    """vad/evaluate.py:20-190 restated: data list -> per-file metrics under the reference's keys -> totals = plain means
    -> output file (totals line, then one line per file).  The probabilities come from a stub (no GPU here)."""
    from voice_activity_detection_amd.evaluate import METRIC_KEYS, evaluate_vad_from_scratch

    seeds = [0, 3, 7]
    lines = []
    for s in seeds:
        y, _ = metric_case(s)
        _labels_to_voice_activity(y).save(tmp_path / f"va{s}.json")
        (tmp_path / f"clip{s}.wav").write_bytes(b"")
        lines.append(json.dumps({"audio_path": f"clip{s}.wav", "voice_activity_path": f"va{s}.json"}))
    (tmp_path / "list.jsonl").write_text("\n".join(lines) + "\n")

    def probabilities(path):  # two extra frames past the labels, as the predictor pads (vad/evaluate.py:59,62 slice them off)
        P = metric_case(int(Path(path).stem[4:]))[1]
        return np.concatenate([P, np.full((2, 7), 0.5, np.float32)])

    printed = []
    out = evaluate_vad_from_scratch(tmp_path / "list.jsonl", output_path=tmp_path / "out" / "eval.jsonl", probabilities_fn=probabilities,
                                    echo=printed.append)
    by_seed = {c["seed"]: c for c in G}
    for s, r in zip(seeds, out["files"]):
        c = by_seed[s]
        assert list(r.keys()) == ["audio_path", "voice_activity_path"] + list(METRIC_KEYS) + ["boosted_" + k for k in METRIC_KEYS]
        assert r["audio_path"] == str(tmp_path / f"clip{s}.wav")
        for k in ("auc", "accuracy", "precision", "recall"):
            assert abs(r[k] - c[k]) < 1e-12 and r["boosted_" + k] == r[k]  # reference quirk: both from the boosted scores
        assert np.allclose([r["vacc"], r["sba"], r["eba"], r["bp"]], np.array(c["vad_accuracy_single"])[[0, 2, 3, 4]], atol=1e-12)
        assert np.allclose([r["boosted_vacc"], r["boosted_sba"], r["boosted_eba"], r["boosted_bp"]],
                           np.array(c["vad_accuracy_boosted"])[[0, 2, 3, 4]], atol=1e-12)
        assert abs(r["eer"] - c["eer_single"]) < 1e-9 and abs(r["boosted_eer"] - c["eer_boosted"]) < 1e-9
    for k, v in out["total"].items():
        assert abs(v - np.mean([r[k] for r in out["files"]])) < 1e-15
    written = [json.loads(x) for x in (tmp_path / "out" / "eval.jsonl").read_text().splitlines()]
    assert len(written) == 1 + len(seeds) and written[0] == out["total"] and written[2]["voice_activity_path"].endswith("va3.json")
    assert len(printed) == len(seeds) + 1 and "Total:" in printed[-1] and "Boosted EER:" in printed[0]
    # --limit / --shuffle follow random.seed(random_seed); random.shuffle (vad/evaluate.py:40-44)
    sub = evaluate_vad_from_scratch(tmp_path / "list.jsonl", probabilities_fn=probabilities, shuffle=True, limit=2, random_seed=5,
                                    echo=lambda s: None)
    import random
    order = list(seeds)
    random.seed(5)
    random.shuffle(order)
    assert [Path(r["audio_path"]).stem for r in sub["files"]] == [f"clip{s}" for s in order[:2]]


def test_cli_has_evaluate_subcommand(capsys):
    import pytest
    from voice_activity_detection_amd.__main__ import main

    with pytest.raises(SystemExit):
        main(["evaluate", "--help"])
    assert "eval_path" in capsys.readouterr().out
