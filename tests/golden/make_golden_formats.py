#!/usr/bin/env python3
"""Golden vectors for the VoiceActivity file formats, produced by the REFERENCE class
(vad/data_models/voice_activity.py:37-246, read-only at /root/reference; pysrt is stubbed in memory):
every writer (to_json v0.1 / v0.2 / v0.3, to_milliseconds v0.2 / v0.3) applied to seeded activity lists,
and what every reader (from_json, from_milliseconds) makes of those documents (as v0.3 JSON + to_labels sums).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_formats.py   ->  tests/golden/golden_formats.json"""
from __future__ import annotations

import json
import sys
import types
from datetime import timedelta
from pathlib import Path

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
pysrt = types.ModuleType("pysrt")
pysrt.SubRipTime = object
sys.modules.setdefault("pysrt", pysrt)
if not hasattr(np, "long"):  # the reference pins numpy 1.19 (np.long = int)
    np.long = np.int64

from vad.data_models.voice_activity import (Activity, VoiceActivity, VoiceActivityMillisecondsVersion,  # noqa: E402
                                            VoiceActivityVersion)


def case(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(0, 6))
    cuts = np.sort(rng.integers(0, 3_600_000, 2 * n))  # milliseconds
    acts = [Activity(timedelta(milliseconds=int(cuts[2 * i])), timedelta(milliseconds=int(cuts[2 * i + 1]))) for i in range(n)]
    dur = timedelta(milliseconds=int(cuts[-1] + rng.integers(0, 5000)) if n else 1234)
    probs = [round(float(v), 4) for v in rng.random(5)] if seed % 2 else None
    return VoiceActivity(dur, acts, 100 if probs else None, probs)


def main():
    out = []
    for seed in range(8):
        va = case(seed)
        docs = {
            "json_v0.1": va.to_json(VoiceActivityVersion.v01),
            "json_v0.2": va.to_json(VoiceActivityVersion.v02),
            "json_v0.3": va.to_json(VoiceActivityVersion.v03),
            "ms_v0.2": va.to_milliseconds(VoiceActivityMillisecondsVersion.v02),
            "ms_v0.3": va.to_milliseconds(VoiceActivityMillisecondsVersion.v03),
        }
        read = {}
        for k, d in docs.items():
            back = VoiceActivity.from_json(d) if k.startswith("json") or k == "ms_v0.2" else VoiceActivity.from_milliseconds(d)
            read[k] = {"as_v0.3": back.to_json(), "label_sum": int(back.to_labels(100).sum()), "n_labels": int(len(back.to_labels(100)))}
        read["ms_v0.2_via_from_milliseconds"] = VoiceActivity.from_milliseconds(docs["ms_v0.2"]).to_json()
        out.append({"seed": seed, "docs": docs, "read": read})
    path = Path(__file__).resolve().parent / "golden_formats.json"
    path.write_text(json.dumps(out, indent=1))
    print("wrote", path, len(out), "cases")


if __name__ == "__main__":
    main()
