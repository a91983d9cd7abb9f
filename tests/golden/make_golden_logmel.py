#!/usr/bin/env python3
"""Independent fixture for the log-mel row (parity with librosa itself stays UNPINNED: librosa
0.8.0 is absent from the reference tree and from this image).  What CAN be pinned without it:

  * the STFT half against an independent implementation -- scipy.signal.stft (scipy 1.15 in this image) on the
    reference's own test clip: librosa.stft(center=True, pad_mode="reflect", n_fft=512, win_length=400, hop 160) centres
    a periodic Hann(400) in a 512 frame, which differs from scipy's "window at the start of the zero-padded segment"
    only by a linear phase, so |X|^2 must agree once the signal is reflect-padded by 256 - 56 = 200 samples;
  * the mel half against the closed form of the Slaney scale and the known answers printed in librosa 0.8's own
    docstrings (tests/test_oracle_golden.py::test_logmel_*).

Output: tests/golden/golden_logmel.npz = every 8th frame of log(mel(|STFT_scipy|^2) + 1e-6) of the clip (float32 [128, 80]),
computed in float64 from scipy's STFT and the closed-form filterbank below (NOT from oracle/logmel.py)."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
from scipy.signal import get_window, stft

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
from voice_activity_detection_amd.features import load_wav_mono16k  # noqa: E402  (stdlib wave reader)


def slaney_hz(m):
    """Slaney / Auditory-Toolbox mel scale in closed form: linear 200/3 Hz per mel below 1 kHz (15 mel), then
    geometric with ratio 6.4^(1/27) per mel."""
    m = np.asarray(m, dtype=np.float64)
    return np.where(m < 15.0, 200.0 / 3.0 * m, 1000.0 * 6.4 ** ((m - 15.0) / 27.0))


def slaney_mel(f):
    f = np.asarray(f, dtype=np.float64)
    return np.where(f < 1000.0, 3.0 * f / 200.0, 15.0 + 27.0 * np.log(np.maximum(f, 1e-30) / 1000.0) / np.log(6.4))


def triangle_filterbank(sr=16000, n_fft=512, n_mels=80):
    """Area-normalised triangles between consecutive mel-spaced edge frequencies, written from the definition."""
    edges = slaney_hz(np.linspace(0.0, slaney_mel(sr / 2.0), n_mels + 2))
    bins = np.arange(n_fft // 2 + 1) * sr / n_fft
    fb = np.zeros((n_mels, len(bins)))
    for i in range(n_mels):
        lo, c, hi = edges[i], edges[i + 1], edges[i + 2]
        up = (bins - lo) / (c - lo)
        down = (hi - bins) / (hi - c)
        fb[i] = np.clip(np.minimum(up, down), 0.0, None) * 2.0 / (hi - lo)
    return fb


def main():
    wav = REPO / "tests" / "golden" / "data" / "WhenTheWeatherIsFine" / "When_the_Weather_Is_Fine_12_4.wav"
    y = load_wav_mono16k(wav).astype(np.float64)
    x = np.pad(y, 200, mode="reflect")
    _, _, Z = stft(x, fs=16000, window="hann", nperseg=400, noverlap=240, nfft=512, boundary=None, padded=False)
    power = (np.abs(Z) * get_window("hann", 400).sum()) ** 2  # undo scipy's 1/sum(window) scaling -> [257, N]
    assert power.shape == (257, 1 + len(y) // 160)
    logmel = np.log(triangle_filterbank() @ power + 1e-6).T  # [N, 80]
    out = REPO / "tests" / "golden" / "golden_logmel.npz"
    np.savez_compressed(out, frames=np.arange(0, logmel.shape[0], 8), logmel=logmel[::8].astype(np.float32),
                        power_frame100=power[:, 100].astype(np.float64))
    print("wrote", out, logmel[::8].shape)


if __name__ == "__main__":
    main()
