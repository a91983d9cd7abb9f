"""CPU oracle of the reference's log-mel front-end -- TEST INFRASTRUCTURE ONLY.

PARITY WITH LIBROSA UNPINNED (its STFT is pinned against scipy.signal.stft and its filterbank against Slaney's
closed form: tests/golden/make_golden_logmel.py, tests/test_oracle_golden.py): the algorithm lives in a third-party
dependency that is absent from
/root/reference and from this image -- librosa (pinned ``librosa==0.8.0`` in the reference's
``requirements.txt:3``) -- and no reference test holds feature values.  This
file restates librosa 0.8.0's published defaults for the reference's only call site,

    librosa.feature.melspectrogram(y, sr=16000, n_mels=80, n_fft=512, hop_length=160, win_length=400)
    feature = np.log(feature + 1e-6)                 vad/acoustics/transforms/log_mel_spectrogram.py:19-32
    features = np.swapaxes(features, 0, 1)           vad/acoustics/feature_extractor.py:71-80   -> [N, 80]

i.e. librosa.stft(center=True, pad_mode="reflect", window="hann" (periodic, scipy get_window), zero-
padded to n_fft around its centre, dtype complex64), power = 2.0, librosa.filters.mel(htk=False
(Slaney scale), norm="slaney", fmin=0, fmax=sr/2, float32), N = 1 + len(y) // hop.
"""
from __future__ import annotations

import numpy as np

SR, N_FFT, HOP, WIN, N_MELS = 16000, 512, 160, 400, 80


def hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sr=SR, n_fft=N_FFT, n_mels=N_MELS, fmin=0.0, fmax=None) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm='slaney', dtype=float32)."""
    fmax = sr / 2.0 if fmax is None else fmax
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis].astype(np.float32)
    return weights


def hann_periodic(n: int) -> np.ndarray:
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n))  # scipy.signal.get_window("hann", n, fftbins=True)


def frame_count(n_samples: int, hop: int = HOP) -> int:
    return 1 + n_samples // hop


def log_mel(y: np.ndarray) -> np.ndarray:
    """y: float32 mono @16 kHz -> float32 [N, 80]."""
    y = np.asarray(y, dtype=np.float32)
    win = np.zeros(N_FFT, dtype=np.float32)
    lpad = (N_FFT - WIN) // 2
    win[lpad:lpad + WIN] = hann_periodic(WIN).astype(np.float32)  # pad_center
    yp = np.pad(y, N_FFT // 2, mode="reflect")
    n = frame_count(len(y))
    idx = HOP * np.arange(n)[:, None] + np.arange(N_FFT)[None, :]
    frames = yp[idx] * win[None, :]  # float32
    D = np.fft.rfft(frames, axis=1).astype(np.complex64)
    S = np.abs(D) ** 2.0  # float32
    mel = S @ mel_filterbank().T  # np.dot(mel_basis, S) transposed
    return np.log(mel + 1e-6).astype(np.float32)
