"""Post-processing of the predict path -- same function names and semantics as the reference's
``vad/postprocessing/{trim,convert,split}.py``, executed by the native host code in libsavad.so
(``csrc/savad_post.h``).  Pinned by goldens produced with the reference functions themselves
(``tests/golden/make_golden_post.py``)."""
from __future__ import annotations

import ctypes
from datetime import timedelta

import numpy as np

from . import _lib


def _ptr(a: np.ndarray):
    return ctypes.c_void_p(a.ctypes.data)


def trim_voice_activity(predictions, min_vally=20, min_hill=20, hang_before=10, hang_over=10):
    """vad/postprocessing/trim.py:4-66.  predictions: 0/1 (or bool) per frame; returns the same dtype."""
    p = np.ascontiguousarray(predictions)
    src = (p != 0).astype(np.uint8)
    out = np.empty_like(src)
    _lib.check(_lib.load().savad_trim_voice_activity(_ptr(src), len(src), int(min_vally), int(min_hill),
                                                     int(hang_before), int(hang_over), _ptr(out)))
    return out.astype(p.dtype)


def convert_frames_to_samples(frames, sample_rate=16000, hop_ms=10, window_ms=10):
    """vad/postprocessing/convert.py:6-24 (float64 result, like numpy.zeros)."""
    f = np.ascontiguousarray(frames, dtype=np.float64)
    lib = _lib.load()
    n = lib.savad_frames_to_samples(_ptr(f), len(f), int(sample_rate), float(hop_ms), float(window_ms), None)
    if n < 0:
        _lib.check(int(n))
    out = np.empty(int(n), dtype=np.float64)
    lib.savad_frames_to_samples(_ptr(f), len(f), int(sample_rate), float(hop_ms), float(window_ms), _ptr(out))
    return out


def segment_indices(samples):
    """(start, end) SAMPLE indices of convert_samples_to_segments' segments."""
    s = np.ascontiguousarray(samples, dtype=np.float64)
    lib = _lib.load()
    cnt = lib.savad_samples_to_segments(_ptr(s), len(s), None, None, 0)
    if cnt < 0:
        _lib.check(cnt)
    starts = np.empty(cnt, dtype=np.int64)
    ends = np.empty(cnt, dtype=np.int64)
    if cnt:
        lib.savad_samples_to_segments(_ptr(s), len(s), _ptr(starts), _ptr(ends), cnt)
    return starts, ends


def convert_samples_to_segments(samples, sample_rate=16000):
    """vad/postprocessing/convert.py:27-61: list of (start, end) timedeltas (seconds = index / sample_rate)."""
    starts, ends = segment_indices(samples)
    return [(timedelta(seconds=int(a) / sample_rate), timedelta(seconds=int(b) / sample_rate)) for a, b in zip(starts, ends)]


def optimal_split_voice_activity(sample_predictions, sample_probs, max_length_seconds=300, sample_rate=16000):
    """vad/postprocessing/split.py:26-78."""
    pred = np.ascontiguousarray(sample_predictions, dtype=np.float64)
    probs = np.ascontiguousarray(sample_probs, dtype=np.float64)
    out = np.empty_like(pred)
    _lib.check(_lib.load().savad_optimal_split(_ptr(pred), _ptr(probs), len(pred), int(max_length_seconds * sample_rate),
                                               _ptr(out)))
    return out
