#!/usr/bin/env python3
"""Golden vectors for the post-processing row, produced by the REFERENCE functions (read-only at
/root/reference): vad/postprocessing/{trim,convert,split}.py are importable as they are;
vad/util/time_utils.py, vad/data_models/voice_activity.py and vad/predictor.py need in-memory stubs
for absent third-party packages (pysrt, omegaconf, more_itertools, librosa, soundfile, cv2).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_post.py

Stores inputs' seeds and the reference's outputs only (tests/golden/golden_post.json)."""
from __future__ import annotations

import json
import sys
import types
from datetime import timedelta
from itertools import islice
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True


class _Stub(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return lambda *a, **k: None


def ichunked(it, n):
    it = iter(it)
    while True:
        chunk = list(islice(it, n))
        if not chunk:
            return
        yield chunk


mi = types.ModuleType("more_itertools")
mi.ichunked = ichunked
oc = types.ModuleType("omegaconf")
oc.MISSING = "???"
oc.OmegaConf = type("OmegaConf", (), {"create": staticmethod(lambda x: x), "to_container": staticmethod(lambda x, **k: x)})
oc.DictConfig = dict
sys.modules.setdefault("more_itertools", mi)
sys.modules.setdefault("omegaconf", oc)
for name in ("librosa", "librosa.feature", "soundfile", "pysrt", "cv2"):
    sys.modules.setdefault(name, _Stub(name))
sys.modules["pysrt"].SubRipTime = object

from vad.postprocessing.convert import convert_frames_to_samples, convert_samples_to_segments  # noqa: E402
from vad.postprocessing.split import optimal_split_voice_activity  # noqa: E402
from vad.postprocessing.trim import trim_voice_activity  # noqa: E402
from vad.util.time_utils import format_timedelta_to_timecode  # noqa: E402

from voice_activity_detection_amd.seeded import seeded_state_dict  # noqa: E402


def runs(rng, n, p_flip):
    x = np.zeros(n, dtype=bool)
    v = False
    for i in range(n):
        if rng.random() < p_flip:
            v = not v
        x[i] = v
    return x


def main():
    g = {"trim": [], "frames_to_samples": [], "segments": [], "split": [], "timecode": [], "predict": []}
    rng = np.random.default_rng(2024)
    for case in range(24):
        n = int(rng.integers(1, 400))
        pred = runs(rng, n, rng.choice([0.02, 0.1, 0.3]))
        args = dict(min_vally=int(rng.integers(0, 30)), min_hill=int(rng.integers(0, 30)),
                    hang_before=int(rng.integers(0, 12)), hang_over=int(rng.integers(0, 12)))
        out = trim_voice_activity(pred, **args)
        g["trim"].append({"pred": pred.astype(int).tolist(), **args, "out": np.asarray(out).astype(int).tolist()})
    for case in range(10):
        n = int(rng.integers(1, 60))
        kind = case % 2
        frames = runs(rng, n, 0.2) if kind == 0 else rng.random(n)
        sr, hop, win = [(16000, 10, 25), (16000, 10, 10), (100, 10, 25), (8000, 10, 25)][case % 4]
        out = convert_frames_to_samples(frames, sample_rate=sr, hop_ms=hop, window_ms=win)
        g["frames_to_samples"].append({"frames": np.asarray(frames, dtype=float).tolist(), "sr": sr, "hop": hop, "win": win,
                                       "n_out": len(out), "sum": float(out.sum()), "head": out[:50].tolist(),
                                       "tail": out[-50:].tolist(), "every97": out[::97].tolist()})
        if kind == 0:
            segs = convert_samples_to_segments(out, sample_rate=sr)
            g["segments"].append({"frames": np.asarray(frames, dtype=float).tolist(), "sr": sr, "hop": hop, "win": win,
                                  "segments_us": [[int(a / timedelta(microseconds=1)), int(b / timedelta(microseconds=1))] for a, b in segs]})
    for case in range(8):
        n = int(rng.integers(50, 3000))
        pred = runs(rng, n, 0.004).astype(float)
        probs = rng.random(n)
        sr = 10
        max_s = int(rng.integers(3, 40))
        out = optimal_split_voice_activity(pred, probs, max_length_seconds=max_s, sample_rate=sr)
        g["split"].append({"seed": int(case), "n": n, "pred": pred.astype(int).tolist(), "probs": probs.tolist(),
                           "max_s": max_s, "sr": sr, "out": np.asarray(out).astype(int).tolist()})
    for us in [0, 1, 499, 500, 501, 1500, 2500, 999499, 999500, 999999, 1000000, 3599999999, 3600000000, 86399999500,
               10213000, 525000, 1225000, 74938, 123456789]:
        g["timecode"].append([us, format_timedelta_to_timecode(timedelta(microseconds=us))])

    # ---- predict(): the reference's whole post-feature path on its own model (seeded weights)
    from vad.models.self_attention import SelfAttentiveVAD
    from vad.predictor import VADFromScratchPredictor, VADPredictParameters

    from oracle import logmel

    model = SelfAttentiveVAD(80, 3, 128, 0.5)
    st = seeded_state_dict(1234)
    # bias the classifier so that the decision flips along the clip (seeded random weights sit near p = 0.5)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    model.eval()
    ns = types.SimpleNamespace
    fe = ns(config=ns(transform=ns(hop_ms=10, window_ms=25)),
            extract_with_postprocessing=lambda audio: logmel.log_mel(audio.audio))
    config = ns(context_resolution=ns(context_window_half_frames=19, context_window_jump_frames=9), model=ns(name="self-attention"))
    pred = VADFromScratchPredictor(model=model, feature_extractor=fe, device=torch.device("cpu"), config=config)
    from vad.data_models.audio_data import AudioData

    for case, (seconds, params) in enumerate([
        (6.0, dict(split_max_seconds=None, threshold=0.5, min_vally_ms=0, min_hill_ms=0, hang_before_ms=0, hang_over_ms=0,
                   activity_max_seconds=None, return_probs=True, probs_sample_rate=100)),
        (9.5, dict(split_max_seconds=4.0, threshold=0.5, min_vally_ms=80, min_hill_ms=60, hang_before_ms=30, hang_over_ms=50,
                   activity_max_seconds=1, return_probs=False, probs_sample_rate=None)),
        (5.0, dict(split_max_seconds=None, threshold=0.47, min_vally_ms=200, min_hill_ms=100, hang_before_ms=100, hang_over_ms=100,
                   activity_max_seconds=None, return_probs=False, probs_sample_rate=None)),
    ]):
        n = int(seconds * 16000)
        arng = np.random.default_rng(700 + case)
        t = np.arange(n) / 16000.0
        env = (np.sin(2 * np.pi * 0.7 * t + case) > 0).astype(np.float32)
        audio = (env * 0.3 * np.sin(2 * np.pi * (200 + 50 * case) * t) + 0.02 * arng.standard_normal(n)).astype(np.float32)
        p = VADPredictParameters(show_progress_bar=False, **params)
        va = pred.predict(AudioData(audio=audio, sample_rate=16000, duration=timedelta(seconds=n / 16000)), p)
        g["predict"].append({"case": case, "seconds": seconds, "params": params, "json": va.to_json()})
    out = Path(__file__).resolve().parent / "golden_post.json"
    out.write_text(json.dumps(g))
    print("wrote", out, out.stat().st_size, "bytes;", {k: len(v) for k, v in g.items()})
    for pr in g["predict"]:
        print(pr["case"], len(pr["json"]["activities"]), pr["json"]["duration"])


if __name__ == "__main__":
    main()
