"""CPU restatement of the reference's resampler -- TEST INFRASTRUCTURE (only tests/ may import it).

The reference resamples with ``librosa.resample(audio, sr, 16000, res_type="kaiser_fast")``
(/root/reference/vad/data_models/audio_data.py:27-30; librosa==0.8.0, requirements.txt:3), which is
``resampy.resample(..., filter="kaiser_fast")`` followed by ``librosa.util.fix_length`` to ``ceil(n * ratio)`` samples
(``fix=True``, ``scale=False`` defaults).  Neither librosa nor resampy exists in this image, so -- like the log-mel
front-end -- PARITY WITH THEM IS UNPINNED; this file restates resampy's published algorithm (J. O. Smith's bandlimited
interpolation, resampy 0.2.x ``core.resample`` / ``interpn.resample_f`` / ``filters.sinc_window``):

* filter "kaiser_fast": the right wing of  rolloff * sinc(rolloff * t)  for t in [0, num_zeros], sampled 2^precision
  times per zero crossing, tapered by the right half of a Kaiser window: num_zeros = 16, precision = 9 (512 samples per
  zero crossing), rolloff = 0.85, beta = 8.555504641634386 (the constants resampy documents for its precomputed table);
* when downsampling the table is scaled by the sample ratio and traversed with a step of int(ratio * 512);
* output sample t sits at input time t / ratio: left wing over x[n], x[n-1], ..., right wing over x[n+1], x[n+2], ...,
  weights linearly interpolated between table entries; accumulation in the output's dtype (float32), in that order;
* int(n * ratio) output samples, zero-padded to ceil(n * ratio) by librosa's fix_length.

Plain loops, written to be read against resampy; use it on fixtures of a few thousand samples.
"""
from __future__ import annotations

import numpy as np

NUM_ZEROS, PRECISION_BITS, ROLLOFF, KAISER_BETA = 16, 9, 0.85, 8.555504641634386


def kaiser_fast_window():
    """resampy.filters.sinc_window(num_zeros=16, precision=9, window=kaiser(beta), rolloff=0.85) -> (half window, 512)"""
    num_bits = 2 ** PRECISION_BITS
    n = num_bits * NUM_ZEROS
    sinc_win = ROLLOFF * np.sinc(ROLLOFF * np.linspace(0, NUM_ZEROS, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, KAISER_BETA)[n:]
    return taper * sinc_win, num_bits


def resample(x: np.ndarray, sr_orig: int, sr_new: int = 16000) -> np.ndarray:
    """float32 mono signal -> librosa.resample(x, sr_orig, sr_new, res_type="kaiser_fast") as restated above"""
    x = np.asarray(x, dtype=np.float32)
    if sr_orig == sr_new:
        return x
    ratio = float(sr_new) / sr_orig
    n_out = int(x.shape[0] * ratio)
    y = np.zeros(n_out, dtype=np.float32)
    interp_win, num_table = kaiser_fast_window()
    if ratio < 1:
        interp_win = interp_win * ratio
    interp_delta = np.zeros_like(interp_win)
    interp_delta[:-1] = np.diff(interp_win)
    scale = min(1.0, ratio)
    time_increment = 1.0 / ratio
    index_step = int(scale * num_table)
    time_register = 0.0
    nwin, n_orig = interp_win.shape[0], x.shape[0]
    for t in range(n_out):
        n = int(time_register)
        frac = scale * (time_register - n)
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        acc = np.float32(0.0)
        i_max = min(n + 1, (nwin - offset) // index_step)
        for i in range(i_max):          # left wing
            weight = interp_win[offset + i * index_step] + eta * interp_delta[offset + i * index_step]
            acc = np.float32(acc + weight * x[n - i])
        frac = scale - frac
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        k_max = min(n_orig - n - 1, (nwin - offset) // index_step)
        for k in range(k_max):          # right wing
            weight = interp_win[offset + k * index_step] + eta * interp_delta[offset + k * index_step]
            acc = np.float32(acc + weight * x[n + k + 1])
        y[t] = acc
        time_register += time_increment
    n_fix = int(np.ceil(x.shape[0] * ratio))     # librosa.util.fix_length
    if n_fix > n_out:
        y = np.concatenate([y, np.zeros(n_fix - n_out, dtype=np.float32)])
    return y[:n_fix]
