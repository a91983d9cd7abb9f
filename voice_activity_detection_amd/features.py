"""Log-mel front-end on the GPU: mirror of the reference's ``FeatureExtractor.extract_with_postprocessing``
for its only shipped transform (log-mel, n_fft 512, hop 10 ms, window 25 ms, 80 mels @16 kHz:
``vad/acoustics/feature_extractor.py:71-80``, ``vad/acoustics/transforms/log_mel_spectrogram.py:19-32``)."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib

SAMPLE_RATE = 16000  # vad/data_models/audio_data.py:9


def resample_to_16k(audio: np.ndarray, sample_rate: int) -> np.ndarray:
    """Rational-ratio polyphase resampling to 16 kHz (scipy.signal.resample_poly, Kaiser-windowed FIR).  The
    reference calls librosa.resample(..., res_type="kaiser_fast") (vad/data_models/audio_data.py:27-30; resampy, a
    windowed-sinc interpolator); neither librosa nor resampy exists here, so sample values are NOT pinned against
    it -- only length (ceil(n * 16000 / rate), as resampy) and spectral content are tested."""
    from math import gcd

    from scipy.signal import resample_poly

    if sample_rate == SAMPLE_RATE:
        return np.asarray(audio, dtype=np.float32)
    g = gcd(SAMPLE_RATE, int(sample_rate))
    out = resample_poly(np.asarray(audio, dtype=np.float64), SAMPLE_RATE // g, int(sample_rate) // g)
    n = int(np.ceil(len(audio) * SAMPLE_RATE / sample_rate))
    return out[:n].astype(np.float32)


def load_wav_mono16k(path) -> np.ndarray:
    """Audio file -> float32 mono @16 kHz in [-1, 1): the reference's AudioData.load (vad/data_models/audio_data.py:
    18-34) with the stdlib instead of soundfile: ``.pcm`` = headerless 16-bit mono @16 kHz (:21-24); otherwise a PCM
    WAV of 8 / 16 / 24 / 32 bits, any channel count (averaged, :26) and any rate (resampled, :27-30)."""
    import wave
    from pathlib import Path

    path = Path(path)
    if path.suffix == ".pcm":
        return (np.fromfile(path, dtype=np.int16).astype(np.float32) / 32768.0).astype(np.float32)
    with wave.open(str(path)) as w:
        rate, width, ch = w.getframerate(), w.getsampwidth(), w.getnchannels()
        raw = w.readframes(w.getnframes())
    if width == 2:
        pcm = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 1:  # unsigned
        pcm = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        pcm = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif width == 4:
        pcm = (np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported PCM sample width {width}")
    if ch > 1:
        pcm = pcm.reshape(-1, ch).mean(axis=1).astype(np.float32)
    return resample_to_16k(pcm, rate)


_HOP, _N_FFT = 160, 512  # savad_logmel.h: HOP, N_FFT (frames = 1 + n // hop; workspace = padded signal + slack)


def log_mel(audio, device="cuda") -> torch.Tensor:
    """audio: 1-D float32 samples @16 kHz (numpy or tensor) -> device tensor [N, 80] float32, N = 1 + len // 160.
    (Host side kept thin on purpose: for a 10 s clip the two kernels take ~25 us, the Python around them used to
    take longer.)"""
    lib = _lib.load()
    dev = device if isinstance(device, torch.device) else torch.device(device)
    y = audio if (isinstance(audio, torch.Tensor) and audio.dtype == torch.float32 and audio.device == dev and audio.is_contiguous()) \
        else torch.as_tensor(audio, dtype=torch.float32).to(dev).contiguous()
    if y.dim() != 1 or y.numel() < 1:
        raise ValueError("audio must be a non-empty 1-D array")
    if y.device.type != "cuda":
        raise _lib.SavadError("the log-mel front-end runs only on a HIP device (no CPU fallback)")
    n = y.numel()
    frames = 1 + n // _HOP
    ws_floats = n + _N_FFT + 64  # savad_logmel_workspace_bytes(n) / 4
    idx = y.device.index if y.device.index is not None else torch.cuda.current_device()
    if idx != torch.cuda.current_device():
        with torch.cuda.device(idx):
            return log_mel(y, y.device)
    buf = torch.empty(ws_floats, dtype=torch.float32, device=y.device)
    out = torch.empty((frames, 80), dtype=torch.float32, device=y.device)
    _lib.check(lib.savad_logmel(ctypes.c_void_p(y.data_ptr()), n, ctypes.c_void_p(buf.data_ptr()),
                                ctypes.c_void_p(out.data_ptr()),
                                ctypes.c_void_p(torch.cuda.current_stream(y.device).cuda_stream)))
    return out
