"""Log-mel front-end on the GPU: mirror of the reference's ``FeatureExtractor.extract_with_postprocessing``
for its only shipped transform (log-mel, n_fft 512, hop 10 ms, window 25 ms, 80 mels @16 kHz:
``vad/acoustics/feature_extractor.py:71-80``, ``vad/acoustics/transforms/log_mel_spectrogram.py:19-32``)."""
from __future__ import annotations

import ctypes
from pathlib import Path

import numpy as np
import torch

from . import _lib

SAMPLE_RATE = 16000  # vad/data_models/audio_data.py:9


_RS_ZEROS, _RS_BITS, _RS_ROLLOFF, _RS_BETA = 16, 9, 0.85, 8.555504641634386   # resampy's "kaiser_fast" table
_rs_table = None


def _kaiser_fast_table():
    """right wing of rolloff * sinc(rolloff * t), t in [0, 16], 512 samples per zero crossing, under the right half of a Kaiser
    window (beta above) + its first differences (for the linear interpolation between entries)"""
    global _rs_table
    if _rs_table is None:
        n = (1 << _RS_BITS) * _RS_ZEROS
        win = np.kaiser(2 * n + 1, _RS_BETA)[n:] * _RS_ROLLOFF * np.sinc(_RS_ROLLOFF * np.linspace(0, _RS_ZEROS, num=n + 1, endpoint=True))
        _rs_table = win
    return _rs_table


def resample_to_16k(audio: np.ndarray, sample_rate: int) -> np.ndarray:
    """Band-limited (windowed-sinc) interpolation to 16 kHz: the algorithm of the reference's
    ``librosa.resample(audio, sr, 16000, res_type="kaiser_fast")`` (vad/data_models/audio_data.py:27-30) = resampy's
    "kaiser_fast" filter -- 16 zero crossings, 512 table entries per crossing linearly interpolated, roll-off 0.85, Kaiser
    beta 8.5555; table scaled by the ratio and strided when downsampling -- then zero-padded to ceil(n * 16000 / rate) samples
    (librosa's fix_length).  numpy only, vectorised over blocks of output samples.  librosa / resampy are absent from this
    image, so parity with THEM is unpinned; this function is held to a loop-by-loop restatement of resampy's published code
    (the test-side CPU restatement, resample.py) on 8 kHz / 44.1 kHz / 48 kHz fixtures (tests/test_postprocessing.py)."""
    x = np.asarray(audio, dtype=np.float32)
    rate = int(sample_rate)
    if rate == SAMPLE_RATE or x.shape[0] == 0:
        return x
    ratio = float(SAMPLE_RATE) / rate
    n_in = x.shape[0]
    n_out = int(n_in * ratio)
    n_fix = int(np.ceil(n_in * ratio))
    win = _kaiser_fast_table()
    num_table = 1 << _RS_BITS
    if ratio < 1:
        win = win * ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    scale = min(1.0, ratio)
    step = int(scale * num_table)
    nwin = win.shape[0]
    taps = nwin // step + 1                                    # upper bound of a wing's length
    xp = x.astype(np.float64)
    y = np.zeros(n_fix, dtype=np.float32)
    # the time register is accumulated as resampy does it (repeated addition, not t * increment)
    times = np.concatenate([[0.0], np.cumsum(np.full(max(n_out - 1, 0), 1.0 / ratio))]) if n_out else np.zeros(0)
    j = np.arange(taps)
    for t0 in range(0, n_out, 32768):
        tr = times[t0:t0 + 32768]
        n = tr.astype(np.int64)
        frac = scale * (tr - n)
        acc = np.zeros(tr.shape[0], dtype=np.float64)
        for wing in (0, 1):
            f = frac if wing == 0 else scale - frac
            idx_f = f * num_table
            off = idx_f.astype(np.int64)
            eta = idx_f - off
            widx = off[:, None] + j[None, :] * step            # table index of tap j
            src = (n[:, None] - j[None, :]) if wing == 0 else (n[:, None] + j[None, :] + 1)
            limit = np.minimum(n + 1, (nwin - off) // step) if wing == 0 else np.minimum(n_in - n - 1, (nwin - off) // step)
            ok = j[None, :] < limit[:, None]
            widx = np.where(ok, widx, 0)
            src = np.where(ok, src, 0)
            w = win[widx] + eta[:, None] * delta[widx]
            acc += np.where(ok, w * xp[src], 0.0).sum(axis=1)
        y[t0:t0 + tr.shape[0]] = acc.astype(np.float32)
    return y


def _riff_wave(path):
    """(rate, channels, width, is_float, raw little-endian sample bytes) of a RIFF/WAVE file: PCM (format 1), IEEE float (3) and
    WAVE_FORMAT_EXTENSIBLE (0xFFFE) wrapping either -- the stdlib `wave` module refuses the last two, soundfile (the reference's reader,
    vad/data_models/audio_data.py:32) reads them"""
    import struct

    data = Path(path).read_bytes()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, raw = 12, None, None
    while pos + 8 <= len(data):
        tag, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            fmt = body
        elif tag == b"data":
            raw = body
            break
        pos += 8 + size + (size & 1)
    if fmt is None or raw is None or len(fmt) < 16:
        raise ValueError(f"{path}: no fmt / data chunk")
    code, ch, rate, _, _, bits = struct.unpack("<HHIIHH", fmt[:16])
    if code == 0xFFFE and len(fmt) >= 26:   # extensible: the sub-format GUID starts with the plain format code
        code = struct.unpack("<H", fmt[24:26])[0]
    if code not in (1, 3):
        raise ValueError(f"{path}: unsupported WAVE format code {code} (PCM and IEEE float are read)")
    return rate, ch, bits // 8, code == 3, raw


def _pcm_to_float(raw: bytes, width: int, big_endian: bool = False, unsigned8: bool = True) -> np.ndarray:
    """interleaved integer PCM bytes -> float32 in [-1, 1) (divide by 2^(bits - 1), as soundfile does)"""
    e = ">" if big_endian else "<"
    if width == 2:
        return np.frombuffer(raw, dtype=e + "i2").astype(np.float32) / 32768.0
    if width == 1:
        if unsigned8:   # WAV
            return (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        return np.frombuffer(raw, dtype=np.int8).astype(np.float32) / 128.0   # AIFF / AU
    if width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        if big_endian:
            b = b[:, ::-1]
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        return (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    if width == 4:
        return (np.frombuffer(raw, dtype=e + "i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    raise ValueError(f"unsupported PCM sample width {width}")


def load_wav_mono16k(path) -> np.ndarray:
    """Audio file -> float32 mono @16 kHz in [-1, 1): the reference's AudioData.load (vad/data_models/audio_data.py:
    18-34) with the stdlib instead of soundfile: ``.pcm`` = headerless 16-bit mono @16 kHz (:21-24); WAV with integer PCM of
    8 / 16 / 24 / 32 bits or IEEE float 32 / 64 (plain or WAVE_FORMAT_EXTENSIBLE); AIFF / AIFF-C (uncompressed) and Sun AU (linear PCM)
    through the stdlib readers; any channel count (averaged, :26) and any rate (resampled, :27-30).  Compressed containers (FLAC,
    OGG, MP3 ...) need a decoder this image does not have: convert them first."""
    path = Path(path)
    suffix = path.suffix.lower()
    if suffix == ".pcm":
        return (np.fromfile(path, dtype=np.int16).astype(np.float32) / 32768.0).astype(np.float32)
    if suffix in (".aiff", ".aif", ".aifc", ".au", ".snd"):
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore", DeprecationWarning)   # (both readers leave the stdlib in Python 3.13)
            mod = __import__("aifc" if suffix.startswith(".aif") else "sunau")
        with mod.open(str(path), "rb") as r:
            if r.getcomptype() not in (b"NONE", "NONE"):
                raise ValueError(f"{path}: compressed {suffix} audio ({r.getcomptype()!r}) is not read")
            rate, width, ch = r.getframerate(), r.getsampwidth(), r.getnchannels()
            pcm = _pcm_to_float(r.readframes(r.getnframes()), width, big_endian=True, unsigned8=False)
    else:
        rate, ch, width, is_float, raw = _riff_wave(path)
        if is_float:
            if width not in (4, 8):
                raise ValueError(f"{path}: IEEE float samples of {8 * width} bits")
            pcm = np.frombuffer(raw[:len(raw) // width * width], dtype="<f4" if width == 4 else "<f8").astype(np.float32)
        else:
            pcm = _pcm_to_float(raw[:len(raw) // width * width], width)
    if ch > 1:
        pcm = pcm[:pcm.size // ch * ch].reshape(-1, ch).mean(axis=1).astype(np.float32)
    return resample_to_16k(pcm, rate)


_HOP, _N_FFT = 160, 512  # savad_logmel.h: HOP, N_FFT (frames = 1 + n // hop; workspace = padded signal + slack)


def pcm16_to_f32(pcm: torch.Tensor) -> torch.Tensor:
    """int16 PCM samples on a HIP device -> float32 in [-1, 1) (sample / 32768, savad_pcm16_to_f32): the conversion soundfile does for
    the reference (vad/data_models/audio_data.py:21-24,32), on the device -- so that an upload moves 2 bytes per sample"""
    if not (isinstance(pcm, torch.Tensor) and pcm.dtype == torch.int16 and pcm.dim() == 1 and pcm.device.type == "cuda" and pcm.is_contiguous()):
        raise ValueError("pcm must be a contiguous 1-D int16 tensor on a HIP device")
    with torch.cuda.device(pcm.device):
        out = torch.empty(pcm.numel(), dtype=torch.float32, device=pcm.device)
        _lib.check(_lib.load().savad_pcm16_to_f32(ctypes.c_void_p(pcm.data_ptr()), pcm.numel(), ctypes.c_void_p(out.data_ptr()),
                                                 ctypes.c_void_p(torch.cuda.current_stream(pcm.device).cuda_stream)))
    return out


def log_mel(audio, device="cuda") -> torch.Tensor:
    """audio: 1-D float32 samples @16 kHz (numpy or tensor; int16 PCM is uploaded as it is and converted on the device) -> device
    tensor [N, 80] float32, N = 1 + len // 160.
    (Host side kept thin on purpose: for a 10 s clip the two kernels take ~25 us, the Python around them used to
    take longer.)"""
    lib = _lib.load()
    dev = device if isinstance(device, torch.device) else torch.device(device)
    if getattr(audio, "dtype", None) in (torch.int16, np.dtype("int16")):
        audio = pcm16_to_f32(torch.as_tensor(audio).to(dev).contiguous())
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


def log_mel_span(audio, audio_first: int, n_samples: int, frame_first: int, frame_count: int) -> torch.Tensor:
    """Frames [frame_first, frame_first + frame_count) of the log-mel matrix of an n_samples-long signal, from a device
    slice `audio` that starts at sample `audio_first` (savad_logmel_span; `span_samples` names the slice a frame span
    needs).  What one rank of a sharded run computes: the same bits as the rows of log_mel(whole signal)."""
    lib = _lib.load()
    if not (isinstance(audio, torch.Tensor) and audio.dtype == torch.float32 and audio.dim() == 1 and audio.device.type == "cuda"
            and audio.is_contiguous()):
        raise ValueError("audio must be a contiguous 1-D float32 tensor on a HIP device")
    with torch.cuda.device(audio.device):
        buf = torch.empty(lib.savad_logmel_span_workspace_bytes(int(frame_count)) // 4, dtype=torch.float32, device=audio.device)
        out = torch.empty((int(frame_count), 80), dtype=torch.float32, device=audio.device)
        _lib.check(lib.savad_logmel_span(ctypes.c_void_p(audio.data_ptr()), int(audio_first), audio.numel(), int(n_samples),
                                         int(frame_first), int(frame_count), ctypes.c_void_p(buf.data_ptr()),
                                         ctypes.c_void_p(out.data_ptr()),
                                         ctypes.c_void_p(torch.cuda.current_stream(audio.device).cuda_stream)))
    return out


def span_samples(n_samples: int, frame_first: int, frame_count: int):
    """(first, count): the samples frames [frame_first, +frame_count) of an n_samples-long signal read (first % 4 == 0)."""
    lib = _lib.load()
    first, count = ctypes.c_long(), ctypes.c_long()
    _lib.check(lib.savad_logmel_span_samples(int(n_samples), int(frame_first), int(frame_count), ctypes.byref(first), ctypes.byref(count)))
    return first.value, count.value
