"""Log-mel front-end on the GPU: mirror of the reference's ``FeatureExtractor.extract_with_postprocessing``
for its only shipped transform (log-mel, n_fft 512, hop 10 ms, window 25 ms, 80 mels @16 kHz:
``vad/acoustics/feature_extractor.py:71-80``, ``vad/acoustics/transforms/log_mel_spectrogram.py:19-32``)."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib

SAMPLE_RATE = 16000  # vad/data_models/audio_data.py:9


def load_wav_mono16k(path) -> np.ndarray:
    """PCM WAV -> float32 mono in [-1, 1) (stdlib only).  The reference's AudioData.load
    (vad/data_models/audio_data.py:18-34) also resamples other rates with librosa; that is not restated:
    only 16 kHz input is accepted."""
    import wave

    with wave.open(str(path)) as w:
        if w.getframerate() != SAMPLE_RATE:
            raise ValueError(f"{path}: only {SAMPLE_RATE} Hz WAV is supported, got {w.getframerate()}")
        if w.getsampwidth() != 2:
            raise ValueError(f"{path}: only 16-bit PCM is supported")
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
        ch = w.getnchannels()
    return pcm.reshape(-1, ch).mean(axis=1).astype(np.float32) if ch > 1 else pcm


@torch.no_grad()
def log_mel(audio, device="cuda") -> torch.Tensor:
    """audio: 1-D float32 samples @16 kHz (numpy or tensor) -> device tensor [N, 80] float32, N = 1 + len // 160."""
    lib = _lib.load()
    dev = torch.device(device)
    y = torch.as_tensor(audio, dtype=torch.float32).to(dev).contiguous()
    if y.dim() != 1 or y.numel() < 1:
        raise ValueError("audio must be a non-empty 1-D array")
    n = y.numel()
    with torch.cuda.device(dev):
        frames = lib.savad_logmel_frames(n)
        ws = torch.empty(lib.savad_logmel_workspace_bytes(n), dtype=torch.uint8, device=dev)
        out = torch.empty((frames, 80), dtype=torch.float32, device=dev)
        _lib.check(lib.savad_logmel(ctypes.c_void_p(y.data_ptr()), n, ctypes.c_void_p(ws.data_ptr()),
                                    ctypes.c_void_p(out.data_ptr()),
                                    ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return out
