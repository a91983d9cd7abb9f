"""Device-side mirror of the hot half of ``VADFromScratchPredictor.predict_probabilities``
(reference ``vad/predictor.py:159-262``): sliding-window gather (:180-220), model forward
(:221-224) and the boosted-prediction scatter / softmax / mean (:238-258, :95).

The feature matrix ``[N, F]`` (log-mel frames; the librosa front-end that produces it is
outside this path) is uploaded ONCE; windows are gathered, evaluated and boosted on the GPU;
only ``probs [N, W]`` (and the per-frame mean) come back.  The reference instead builds every
window in a Python loop, pushes ``[B, 7, 80]`` (7x read amplification) per chunk and scatters
with numpy on the host.
"""
from __future__ import annotations

import ctypes
import math
import os
from collections import OrderedDict
from dataclasses import dataclass
from datetime import timedelta
from typing import Optional

import numpy as np
import torch

from . import _lib
from .data_models import Activity, VoiceActivity, merge_voice_activities
from .model import SelfAttentiveVAD
from .postprocessing import (convert_frames_to_samples, convert_samples_to_segments, optimal_split_voice_activity,
                             trim_voice_activity)


def window_offsets(half: int, jump: int) -> np.ndarray:
    """arange(-half, 0, jump) ++ [0] ++ arange(1, half+1, jump)  (vad/predictor.py:186-212)."""
    buf = (ctypes.c_int32 * 64)()
    w = _lib.load().savad_window_offsets(half, jump, buf)
    if w < 0:
        _lib.check(w)
    return np.array(buf[:w], dtype=np.int64)


@dataclass
class ContextResolution:
    """config.context_resolution of the reference (vad/configs/dataset_config.py:7-10)."""
    context_window_half_frames: int = 19
    context_window_jump_frames: int = 9


@dataclass
class VADPredictParameters:
    """vad/predictor.py:27-38 (same field order as the reference's positional construction in vad/predict.py:32-43)."""
    split_max_seconds: Optional[float] = None
    threshold: float = 0.5
    min_vally_ms: int = 0
    min_hill_ms: int = 0
    hang_before_ms: int = 0
    hang_over_ms: int = 0
    activity_max_seconds: Optional[int] = None
    return_probs: bool = False
    probs_sample_rate: Optional[int] = None
    show_progress_bar: bool = False


class VADFromScratchPredictor:
    """The reference class of the same name (vad/predictor.py:41-262) for the self-attention model:
    predict_probabilities on the GPU (log-mel, window gather, forward, boost), predict() with the
    reference's chunking and post-processing (native host code)."""

    hop_ms, window_ms = 10, 25  # the reference's only transform config (tests/configs/vad/train_config.yaml:21-22)

    def __init__(self, model: SelfAttentiveVAD, device: torch.device, context: ContextResolution = ContextResolution(),
                 chunk_size: int = 16384, graph: bool = False, graph_max_seconds: float = 120.0, graph_cache: int = 8):
        """`graph=True` (not in the reference's signature): clip-sized inputs -- up to `graph_max_seconds` of audio -- run as a
        replayed HIP graph of the whole chain log-mel -> window gather -> forward -> boost, captured on first use per (length, model
        knobs) and re-captured when the weights change; at most `graph_cache` graphs are kept (least recently used goes).  For a 10 s
        clip the four launches take less time than the Python and the library calls around them: the replay halves the time per clip,
        same bits (predict_audio_device)."""
        self.graph, self.graph_max_seconds, self.graph_cache = bool(graph), float(graph_max_seconds), int(graph_cache)
        self._graphs: "OrderedDict[tuple, dict]" = OrderedDict()
        self.graph_stats = {"captures": 0, "replays": 0, "eager": 0}
        self.model = model
        self.device = torch.device(device)
        self.context_window_half_frames = context.context_window_half_frames
        self.context_window_jump_frames = context.context_window_jump_frames
        # vad/predictor.py:57-59
        self.context_window_frames = 2 * (self.context_window_half_frames - 1) // self.context_window_jump_frames + 3
        # windows per forward.  The reference uses 1000 (vad/predictor.py:180); windows are independent, so any value gives
        # the same probabilities up to fp32 summation order (<= 1e-6: a larger batch may pick another launch schedule), and
        # 10 min of audio take 9.4 ms with 1000 against 5.1 ms with 16384
        self.chunk_size = int(chunk_size)

    @staticmethod
    def _load_checkpoint(checkpoint_path, trust: bool):
        """torch.load restricted to tensors, containers and the numpy scalar / dtype reconstructors a reference
        checkpoint's `metrics` dict holds (weights_only=True inside safe_globals); arbitrary pickles (e.g. a checkpoint
        whose `config` is a pickled OmegaConf object) load only on explicit opt-in: trust=True or SAVAD_TRUST_CHECKPOINT=1."""
        import numpy as np
        safe = [np.dtype, np.ndarray, type(np.dtype("float64")), type(np.dtype("float32")), type(np.dtype("int64")),
                type(np.dtype("int32")), type(np.dtype("bool"))]
        for mod in ("numpy._core.multiarray", "numpy.core.multiarray"):
            try:
                m = __import__(mod, fromlist=["scalar", "_reconstruct"])
                safe += [m.scalar, m._reconstruct]
                break
            except (ImportError, AttributeError):
                continue
        import pickle

        by_env = not trust and os.environ.get("SAVAD_TRUST_CHECKPOINT") == "1"
        trusted = trust or by_env

        def full_unpickle():
            if by_env:   # a process-wide switch is easy to forget: say which file it just applied to
                import warnings
                warnings.warn(f"SAVAD_TRUST_CHECKPOINT=1: unpickling {checkpoint_path} in full (code embedded in the file can run); "
                              "prefer trust_checkpoint=True on the calls that need it", RuntimeWarning, stacklevel=3)
            try:
                return torch.load(checkpoint_path, map_location="cpu", weights_only=False)
            except TypeError:   # torch < 1.13: no weights_only keyword
                return torch.load(checkpoint_path, map_location="cpu")

        if not hasattr(torch.serialization, "safe_globals"):
            if trusted:   # an older torch: the explicit opt-in still loads (what the reference's plain torch.load does)
                return full_unpickle()
            raise RuntimeError("this torch build has no torch.serialization.safe_globals: checkpoints cannot be loaded with weights_only=True "
                               "(pass trust_checkpoint=True / set SAVAD_TRUST_CHECKPOINT=1 if you trust the file)")
        try:
            with torch.serialization.safe_globals(safe):
                return torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError as exc:  # an unlisted global: only this falls through; I/O and format errors propagate as they are
            if not trusted:
                raise RuntimeError(
                    f"{checkpoint_path}: not loadable with weights_only=True ({str(exc).splitlines()[0]}). If you trust the "
                    "file, pass trust_checkpoint=True / set SAVAD_TRUST_CHECKPOINT=1 to unpickle it in full "
                    "(this can execute code embedded in the file).") from exc
        return full_unpickle()

    @classmethod
    def from_checkpoint(cls, checkpoint_path, device, trust_checkpoint: bool = False):
        """vad/predictor.py:264-280: a training checkpoint holds {"config": ..., "state_dict": ...} plus what
        ModelCheckpointer adds (epoch, global_step, monitor_metric, a `metrics` dict of numpy scalars, optimizer /
        scheduler / grad-scaler state: vad/training/checkpointers/model_checkpointer.py:97-110); the model size, the
        feature transform and the window geometry come from the config, the weights load strictly.  Unlike the
        reference's plain torch.load, the file is first read with weights_only=True (see _load_checkpoint)."""
        ckpt = cls._load_checkpoint(checkpoint_path, trust_checkpoint)
        cfg = ckpt["config"]

        def get(c, *names, default=None):
            for n in names:
                if isinstance(c, dict):
                    if n not in c:
                        return default
                    c = c[n]
                elif hasattr(c, n):
                    c = getattr(c, n)
                else:
                    return default
            return c

        if get(cfg, "model", "name") != "self-attention":
            raise NotImplementedError("only the self-attention model is built for MI355X")
        sa = get(cfg, "model", "self_attention")
        # The device front-end (savad_logmel) implements the reference's one shipped transform configuration
        # (tests/configs/vad/train_config.yaml:18-28).  Anything else would load cleanly and then produce wrong features
        # or timings, so it is refused here rather than failing late with a shape error.
        fe = get(cfg, "feature_extractor")
        tr = get(fe, "transform")
        want = {"name": "log-mel", "n_fft": 512, "hop_ms": 10, "window_ms": 25, "n_mels": 80}
        got = {k: get(tr, k) for k in want}
        if got != want:
            raise NotImplementedError(f"unsupported transform {got}: the MI355X front-end implements {want} "
                                      "(vad/acoustics/transforms/transform_factory.py:30-59 has further transforms)")
        if get(fe, "temporal_differences", default=False) or get(fe, "stack_differences", default=False):
            raise NotImplementedError("unsupported feature_extractor: temporal_differences / stack_differences "
                                      "(vad/acoustics/feature_extractor.py:135-147) are not built")
        if get(fe, "silence_remover", default=None):
            raise NotImplementedError("unsupported feature_extractor: silence_remover (vad/acoustics/feature_extractor.py:115-116) is not built")
        model = SelfAttentiveVAD(got["n_mels"], get(sa, "num_layers"), get(sa, "d_model"), get(sa, "dropout"))
        model.load_state_dict(ckpt["state_dict"])
        ctx = ContextResolution(get(cfg, "context_resolution", "context_window_half_frames"),
                                get(cfg, "context_resolution", "context_window_jump_frames"))
        predictor = cls(model.to(device).eval(), device, ctx)
        predictor.hop_ms, predictor.window_ms = got["hop_ms"], got["window_ms"]  # vad/predictor.py:103-104
        return predictor

    def predict_from_path(self, audio_path, parameters: VADPredictParameters) -> VoiceActivity:
        """vad/predictor.py:71-75 (16 kHz PCM WAV only; the reference also resamples via librosa)."""
        from .features import load_wav_mono16k

        return self.predict(load_wav_mono16k(audio_path), parameters)

    def predict(self, audio: np.ndarray, parameters: VADPredictParameters, features_fn=None) -> VoiceActivity:
        """vad/predictor.py:77-157.  audio: float32 mono @16 kHz.  features_fn(chunk_audio) -> [N, F] overrides
        the GPU log-mel front-end (used by the parity tests to feed the reference's feature matrix)."""
        from .features import SAMPLE_RATE, log_mel

        audio = np.asarray(audio, dtype=np.float32)
        duration_s = len(audio) / SAMPLE_RATE
        num_chunks = math.ceil(duration_s / parameters.split_max_seconds) if parameters.split_max_seconds is not None else 1
        adjusted = duration_s / num_chunks
        chunks = []
        for ci in range(num_chunks):
            start_sample = int(ci * adjusted * SAMPLE_RATE)
            end_sample = int((ci + 1) * adjusted * SAMPLE_RATE)
            chunk = audio[start_sample:end_sample]
            if features_fn is not None:
                probs_dev, mean_dev = self.predict_probabilities_device(features_fn(chunk))
            else:
                probs_dev, mean_dev = self.predict_audio_device(chunk)   # (a replayed HIP graph for clip-sized chunks when graph=True)
            # float64 mean of the float32 [N, 7] matrix, like numpy's probs.mean(axis=1) on the host (:95)
            boosted = probs_dev.cpu().numpy().mean(axis=1)
            predictions = boosted > parameters.threshold
            hop_ms, window_ms = self.hop_ms, self.window_ms
            trimmed = trim_voice_activity(predictions, min_vally=round(parameters.min_vally_ms / hop_ms),
                                          min_hill=round(parameters.min_hill_ms / hop_ms),
                                          hang_before=round(parameters.hang_before_ms / hop_ms),
                                          hang_over=round(parameters.hang_over_ms / hop_ms))
            sample_predictions = convert_frames_to_samples(trimmed, sample_rate=16000, hop_ms=hop_ms, window_ms=window_ms)
            if parameters.activity_max_seconds is not None and parameters.activity_max_seconds > 0:
                sample_full_probs = convert_frames_to_samples(boosted, sample_rate=16000, hop_ms=hop_ms, window_ms=window_ms)
                sample_predictions = optimal_split_voice_activity(sample_predictions, sample_full_probs,
                                                                  max_length_seconds=parameters.activity_max_seconds,
                                                                  sample_rate=16000)
            activities = [Activity(start=a, end=b) for a, b in convert_samples_to_segments(sample_predictions, 16000)]
            probs = None
            if parameters.return_probs:
                probs = convert_frames_to_samples(boosted, sample_rate=parameters.probs_sample_rate, hop_ms=hop_ms,
                                                  window_ms=window_ms).tolist()
            chunks.append(VoiceActivity(duration=timedelta(seconds=adjusted), activities=activities,
                                        probs_sample_rate=parameters.probs_sample_rate if parameters.return_probs else None,
                                        probs=probs))
        return merge_voice_activities(chunks)

    def predict_probabilities(self, feature) -> np.ndarray:
        """feature [N, F] (numpy or tensor) -> positive-class probabilities [N, W] (float32 numpy),
        exactly the array the reference returns from predict_probabilities (:257-260)."""
        probs, _ = self.predict_probabilities_device(feature)
        return probs.cpu().numpy()

    def predict_boosted(self, feature) -> np.ndarray:
        """Per-frame boosted probability = probs.mean(axis=1) (vad/predictor.py:95)."""
        _, mean = self.predict_probabilities_device(feature)
        return mean.cpu().numpy()

    @torch.no_grad()
    def predict_probabilities_device(self, feature):
        """feature [N, F] -> (probs [N, W], mean [N]) on the device: window gather (a13), forward, boosted prediction (a14)
        in one library call (savad_predict_probabilities)."""
        if self.device.type != "cuda":
            raise _lib.SavadError("the MI355X predictor needs a HIP device (no CPU fallback)")
        feat = torch.as_tensor(feature, dtype=torch.float32).to(self.device)
        if feat.dim() != 2:
            raise ValueError("feature must be [N, F]")
        self.model.eval()
        return self.model.predict_windows(feat, self.context_window_half_frames, self.context_window_jump_frames, self.chunk_size)

    @torch.no_grad()
    def predict_audio_device(self, audio):
        """audio: 1-D float32 mono @16 kHz (numpy or tensor) -> (probs [N, W], mean [N]) on the device, N = 1 + len // 160: the GPU
        log-mel front-end (vad/acoustics/transforms/log_mel_spectrogram.py:19-32) followed by predict_probabilities_device
        (vad/predictor.py:159-262) -- what `predict` runs per chunk.  With `graph=True` a clip-sized input replays a captured HIP
        graph (same kernels, same bits); the returned tensors then belong to the graph and are overwritten by the next call with
        the same length and knobs: copy them if they must outlive it."""
        from .features import SAMPLE_RATE, log_mel

        if self.device.type != "cuda":
            raise _lib.SavadError("the MI355X predictor needs a HIP device (no CPU fallback)")
        n = int(audio.shape[0]) if hasattr(audio, "shape") and len(audio.shape) == 1 else -1
        if not self.graph or n < 1 or n > self.graph_max_seconds * SAMPLE_RATE or self.model.training:
            self.graph_stats["eager"] += 1
            return self.predict_probabilities_device(log_mel(audio, self.device))
        model = self.model
        dev = self.device if self.device.index is not None else torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(dev):
            model._prepare_call(dev)   # weights pushed, knobs set: the walk over the parameters' versions costs ~8 us
            key = (n, dev.index, model._pushed_knobs, self.context_window_half_frames, self.context_window_jump_frames, self.chunk_size)
            entry = self._graphs.get(key)
            if entry is not None and entry["weights"] is not model._synced_versions:
                # the weights were pushed again since the capture: the library re-folds / re-packs them inside its NEXT call, which a
                # replay never makes -- every graph of this predictor is stale
                self._graphs.clear()
                entry = None
            if entry is None:
                entry = self._capture(key, n, dev)
            else:
                self._graphs.move_to_end(key)
            src = audio if isinstance(audio, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32))
            entry["audio"].copy_(src, non_blocking=True)
            entry["graph"].replay()
            self.graph_stats["replays"] += 1
        return entry["probs"], entry["mean"]

    @staticmethod
    def host_chunk_plan(N: int, half: int, W: int, frames_per_chunk: int):
        """[(f0, f1, g0, g1)]: output frames [f0, f1) of an N-frame recording are computed from feature frames [g0, g1).  A frame's W
        probabilities come from the windows centred within `half` of it (vad/predictor.py:238-258), a window reads the frames within
        `half` of its centre (:186-212): 2 x half frames of halo per side, clipped to the recording -- where the slice is clipped, the
        windows that are missing are exactly the ones the whole recording lacks there (the 0.5 placeholders).  g0 is a multiple of
        32 // W, so that every window keeps its slot in a packed 32-row block (the single-launch kernels sum a slot's keys in slot order);
        a tail shorter than half a chunk rides with the chunk before it (a tiny last chunk would run another kernel variant).
        Host-side arithmetic only."""
        G = max(32 // W, 1)
        per = max(int(frames_per_chunk), 4 * half)
        starts = list(range(0, N, per))
        if len(starts) > 1 and N - starts[-1] < per // 2:
            starts.pop()
        plan = []
        for k, f0 in enumerate(starts):
            f1 = starts[k + 1] if k + 1 < len(starts) else N
            plan.append((f0, f1, max(0, f0 - 2 * half) // G * G, min(N, f1 + 2 * half)))
        return plan

    @torch.no_grad()
    def predict_audio_host(self, audio, frames_per_chunk: int = 65536):
        """The reference's mode END TO END from host memory: `audio` = the whole recording on the host, mono 16 kHz, 16-bit PCM
        (uploaded as it is, converted on the device) or float32, numpy array or CPU tensor (pinned: asynchronous uploads) ->
        (probs [N, W], mean [N]) on the device.  Output frames are produced in chunks of `frames_per_chunk`; chunk c needs the feature
        frames within 2 x half of its own (every window that reaches one of its frames, vad/predictor.py:186-258), whose samples are
        uploaded on a copy stream while chunk c - 1 runs.  A chunk's first window keeps its place modulo the packed block (a multiple
        of 32 // W windows), so the results are predict_audio_device's bits."""
        from .features import log_mel_span, pcm16_to_f32, span_samples

        if self.device.type != "cuda":
            raise _lib.SavadError("the MI355X predictor needs a HIP device (no CPU fallback)")
        src = StreamingPredictor._host_source(audio)
        n = int(src.shape[0])
        N = 1 + n // 160
        half, jump, Wn = self.context_window_half_frames, self.context_window_jump_frames, self.context_window_frames
        plan = [(f0, f1, g0, g1) + tuple(span_samples(n, g0, g1 - g0))
                for f0, f1, g0, g1 in self.host_chunk_plan(N, half, Wn, frames_per_chunk)]
        self.model.eval()
        dev = self.device if self.device.index is not None else torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            if getattr(self, "_copy_stream", None) is None or self._copy_stream.device != dev:
                self._copy_stream = torch.cuda.Stream(dev)
            cs = self._copy_stream
            dev_audio = torch.empty(n, dtype=src.dtype, device=dev)
            dev_audio.record_stream(cs)
            cs.wait_stream(cur)
            probs = torch.empty((N, Wn), dtype=torch.float32, device=dev)
            mean = torch.empty((N,), dtype=torch.float32, device=dev)
            uploaded = 0

            def upload(c):
                nonlocal uploaded
                end = plan[c][4] + plan[c][5]
                ev = torch.cuda.Event()
                with torch.cuda.stream(cs):
                    if end > uploaded:
                        dev_audio[uploaded:end].copy_(src[uploaded:end], non_blocking=True)
                        uploaded = end
                    ev.record(cs)
                return ev

            ev = upload(0)
            for c, (f0, f1, g0, g1, first, count) in enumerate(plan):
                nxt = upload(c + 1) if c + 1 < len(plan) else None
                cur.wait_event(ev)
                sl = dev_audio[first:first + count]
                if sl.dtype == torch.int16:
                    sl = pcm16_to_f32(sl)
                feat = log_mel_span(sl, first, n, g0, g1 - g0)
                p, mu = self.model.predict_windows(feat, half, jump, self.chunk_size)
                probs[f0:f1].copy_(p[f0 - g0:f1 - g0])
                mean[f0:f1].copy_(mu[f0 - g0:f1 - g0])
                ev = nxt
        return probs, mean

    def _capture(self, key, n: int, dev: torch.device) -> dict:
        from .features import log_mel

        static_audio = torch.zeros(n, dtype=torch.float32, device=dev)

        def chain():
            return self.model.predict_windows(log_mel(static_audio, dev), self.context_window_half_frames,
                                              self.context_window_jump_frames, self.chunk_size)

        # eager runs first, on a side stream like the capture itself: positional-encoding table, packed weights and the cached
        # workspace reach their final sizes, so that the captured call launches nothing but its own kernels
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                chain()
        torch.cuda.current_stream(dev).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            probs, mean = chain()
        entry = {"graph": g, "audio": static_audio, "probs": probs, "mean": mean, "weights": self.model._synced_versions,
                 "workspace": self.model._workspace}   # (the graph holds the workspace's address: keep the block alive)
        self._graphs[key] = entry
        while len(self._graphs) > max(self.graph_cache, 1):
            self._graphs.popitem(last=False)
        self.graph_stats["captures"] += 1
        return entry

    @torch.no_grad()
    def predict_probabilities_device_stepwise(self, feature):
        """The same through the three separate entry points (savad_gather_windows, savad_forward, savad_boost): what
        savad_predict_probabilities must reproduce bit for bit (tests)."""
        lib = _lib.load()
        feat = torch.as_tensor(feature, dtype=torch.float32).to(self.device).contiguous()
        N, F = feat.shape
        half, jump, W = self.context_window_half_frames, self.context_window_jump_frames, self.context_window_frames
        data_length = N - 2 * half  # vad/predictor.py:169
        self.model.eval()
        with torch.cuda.device(self.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            n_items = max(data_length, 0)
            logp = torch.empty((n_items, W, 2), dtype=torch.float32, device=self.device)
            pos = torch.empty((n_items, W), dtype=torch.int64, device=self.device)
            for first in range(0, n_items, self.chunk_size):
                count = min(self.chunk_size, n_items - first)
                win = torch.empty((count, W, F), dtype=torch.float32, device=self.device)
                _lib.check(lib.savad_gather_windows(ctypes.c_void_p(feat.data_ptr()), N, F, half, jump, first, count,
                                                    ctypes.c_void_p(win.data_ptr()),
                                                    ctypes.c_void_p(pos[first:first + count].data_ptr()), stream))
                self.model(features=win, out=logp[first:first + count])  # straight into the boost's input
            probs = torch.empty((N, W), dtype=torch.float32, device=self.device)
            mean = torch.empty((N,), dtype=torch.float32, device=self.device)
            if N > 0:
                boosted = torch.empty((N, W, 2), dtype=torch.float32, device=self.device)
                _lib.check(lib.savad_boost(ctypes.c_void_p(logp.data_ptr()), ctypes.c_void_p(pos.data_ptr()), n_items, N,
                                           W, ctypes.c_void_p(boosted.data_ptr()), ctypes.c_void_p(probs.data_ptr()),
                                           ctypes.c_void_p(mean.data_ptr()), stream))
        return probs, mean


class StreamingPredictor:
    """Long-form mode of BASELINE.json configs[4]: sliding windows of T frames every `hop` frames over
    a long feature matrix [N, F]; each window is one sequence of the self-attentive model; per-frame
    speech probability = mean over the (<= T/hop) windows covering the frame.  Not a mode of the
    reference (its predictor only cuts 7-frame windows); semantics defined in include/savad.h.
    With torch.distributed initialised, windows are sharded contiguously over the ranks and the
    log-probs are exchanged with ONE all_gather (voice_activity_detection_amd.distributed)."""

    def __init__(self, model: SelfAttentiveVAD, device, T: int = 800, hop: int = 400, max_batch: int = 256, in_flight: int = 2):
        self.model, self.device, self.T, self.hop, self.max_batch = model, torch.device(device), int(T), int(hop), int(max_batch)
        # the window batches are independent: `in_flight` of them run concurrently (PipelinedVAD: own stream / handle / workspace
        # each, same bits); 1 = one after the other on the caller's stream.
        # bf16 operands: a full batch of max_batch windows and a smaller remainder (or a rank's smaller shard) may run different attention
        # kernels, whose results differ in the bf16 rounding of a few rows' context (<= 3e-3 in the log-probs); set
        # model.batch_invariant = True where the same window must give the same bits in every batching (+4 % of a large forward)
        self.in_flight = int(in_flight)
        self._pipe = None

    @torch.no_grad()
    def predict_device(self, feature):
        from .distributed import sharded_rows

        lib = _lib.load()
        feat = torch.as_tensor(feature, dtype=torch.float32).to(self.device).contiguous()
        N, F = feat.shape
        T, hop = self.T, self.hop
        W = lib.savad_stream_window_count(N, T, hop)
        if W < 0:
            _lib.check(W)
        self.model.eval()
        with torch.cuda.device(self.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

            if self._pipe is None or self._pipe.model is not self.model or self._pipe.depth != max(self.in_flight, 1):
                from .pipeline import PipelinedVAD
                self._pipe = PipelinedVAD(self.model, depth=max(self.in_flight, 1))

            def windows_logp(lo, hi):  # this rank's contiguous span of windows -> [hi - lo, T, 2] log-probs
                return self._windows_logp(feat, 0, N, lo, hi, stream)

            logp = sharded_rows(W, windows_logp, (T, 2), torch.float32, self.device).contiguous()
            probs = torch.empty((N,), dtype=torch.float32, device=self.device)
            _lib.check(lib.savad_overlap_merge(ctypes.c_void_p(logp.data_ptr()), W, N, T, hop,
                                               ctypes.c_void_p(probs.data_ptr()), stream))
        return probs

    def _windows_logp(self, feat, frame0, n_total, lo, hi, stream, out=None, join=True):
        """log-probs [hi - lo, T, 2] of windows [lo, hi) of an n_total-frame recording, from `feat` = its frames [frame0,
        frame0 + len(feat)): windows that lie inside the recording are read IN PLACE (model.forward_windows, sequences hop * F
        elements apart: no copies); the last window of a recording that does not end on a window boundary is zero-padded
        through savad_gather_strided (one window's copy)."""
        lib = _lib.load()
        T, hop, F = self.T, self.hop, feat.shape[1]
        n_local = feat.shape[0]
        local = out if out is not None else torch.empty((hi - lo, T, 2), dtype=torch.float32, device=self.device)
        full_end = min(hi, (n_total - T) // hop + 1 if n_total >= T else 0)   # windows [lo, full_end) end inside the recording
        for first in range(lo, full_end, self.max_batch):
            count = min(self.max_batch, full_end - first)
            self._pipe.submit_windows(feat, T, hop, first - frame0 // hop, count, out=local[first - lo:first - lo + count])
        for w in range(max(lo, full_end), hi):   # at most one: the padded tail
            if F % 4:   # (savad_gather_strided moves 16-byte pieces)
                win = torch.zeros((1, T, F), dtype=torch.float32, device=self.device)
                f_lo = hop * (w - frame0 // hop)
                win[0, :max(min(T, n_local - f_lo), 0)] = feat[f_lo:f_lo + T]
            else:
                win = torch.empty((1, T, F), dtype=torch.float32, device=self.device)
                _lib.check(lib.savad_gather_strided(ctypes.c_void_p(feat.data_ptr()), n_local, F, T, hop, w - frame0 // hop, 1,
                                                    ctypes.c_void_p(win.data_ptr()), stream))
            self._pipe.submit(win, out=local[w - lo:w - lo + 1])
        if join:
            self._pipe.join()
        return local

    @staticmethod
    def audio_shard_plan(n_samples: int, T: int, hop: int, rank: int, world: int):
        """What rank `rank` of `world` needs of an n_samples-long recording: (W, lo, hi, f0, f1, first, count) -- its windows [lo, hi)
        of the W in all (contiguous split, voice_activity_detection_amd.distributed.shard_bounds), the feature frames [f0, f1)
        they cover, and the samples [first, first + count) those frames read (savad_logmel_span_samples: 208 samples of halo
        per side, the mirrored stretch at an end of the recording; first % 4 == 0).  Host-side arithmetic only."""
        from .distributed import shard_bounds
        from .features import span_samples

        lib = _lib.load()
        N = 1 + n_samples // 160
        W = lib.savad_stream_window_count(N, T, hop)
        if W < 0:
            _lib.check(W)
        lo, hi = shard_bounds(W, rank, world)
        if hi <= lo:
            return W, lo, hi, 0, 0, 0, 0
        f0, f1 = hop * lo, min(N, hop * (hi - 1) + T)
        first, count = span_samples(n_samples, f0, f1 - f0)
        return W, lo, hi, f0, f1, first, count

    @torch.no_grad()
    def audio_span_logp(self, audio, rank: int, world: int):
        """rank-local half of predict_audio_device: log-probs [hi - lo, T, 2] of this rank's windows, from the audio -- only its
        slice of the samples is uploaded (when `audio` is a host array), only its frames' log-mel is computed"""
        from .features import log_mel_span

        n = int(audio.shape[0])
        W, lo, hi, f0, f1, first, count = self.audio_shard_plan(n, self.T, self.hop, rank, world)
        if hi <= lo:
            return torch.zeros((0, self.T, 2), dtype=torch.float32, device=self.device)
        self.model.eval()
        with torch.cuda.device(self.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            if self._pipe is None or self._pipe.model is not self.model or self._pipe.depth != max(self.in_flight, 1):
                from .pipeline import PipelinedVAD
                self._pipe = PipelinedVAD(self.model, depth=max(self.in_flight, 1))
            sl = audio[first:first + count]
            if not isinstance(sl, torch.Tensor):
                sl = torch.from_numpy(np.ascontiguousarray(sl) if sl.dtype == np.int16 else np.ascontiguousarray(sl, dtype=np.float32))
            if sl.dtype == torch.int16:   # 16-bit PCM goes up as it is (half the bytes) and is converted on the device
                from .features import pcm16_to_f32
                sl = pcm16_to_f32(sl.to(self.device).contiguous())
            else:
                sl = sl.to(self.device, torch.float32).contiguous()
            feat = log_mel_span(sl, first, n, f0, f1 - f0)
            return self._windows_logp(feat, f0, 1 + n // 160, lo, hi, stream)

    @torch.no_grad()
    def predict_audio_device(self, audio):
        """configs[4] from the AUDIO: `audio` = the whole recording, mono 16 kHz, float32 or 16-bit PCM (uploaded as it is, converted on the
        device), on the HOST (numpy) or the device.  With
        torch.distributed initialised every rank computes the log-mel features of ITS window span only (audio_shard_plan), runs
        its windows in place on them, and ONE all_gather of the log-probs + the overlap merge give every rank the per-frame
        probabilities.  Same bits as predict_device(log_mel(audio)) on one GPU (tests/test_gpu_parity.py)."""
        import torch.distributed as dist

        from .distributed import all_gather_rows

        lib = _lib.load()
        n = int(audio.shape[0])
        N = 1 + n // 160
        world, rank = (dist.get_world_size(), dist.get_rank()) if dist.is_initialized() else (1, 0)
        local = self.audio_span_logp(audio, rank, world)
        W = lib.savad_stream_window_count(N, self.T, self.hop)
        with torch.cuda.device(self.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            logp = (all_gather_rows(local, W) if world > 1 else local).contiguous()
            probs = torch.empty((N,), dtype=torch.float32, device=self.device)
            _lib.check(lib.savad_overlap_merge(ctypes.c_void_p(logp.data_ptr()), W, N, self.T, self.hop,
                                               ctypes.c_void_p(probs.data_ptr()), stream))
        return probs

    @staticmethod
    def _host_source(audio):
        """host audio (numpy array or CPU tensor, int16 PCM or float32) as a CPU tensor without a copy"""
        src = audio if isinstance(audio, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(audio))
        if src.device.type != "cpu" or src.dim() != 1 or src.numel() < 1 or src.dtype not in (torch.int16, torch.float32) or not src.is_contiguous():
            raise ValueError("audio must be a non-empty contiguous 1-D int16 or float32 array on the host")
        return src

    @torch.no_grad()
    def predict_audio_host(self, audio, windows_per_chunk: Optional[int] = None, ramp: bool = False):
        """configs[4] END TO END from host memory on one GPU: `audio` = the whole recording on the host, mono 16 kHz, as 16-bit PCM
        (AudioData's source format, vad/data_models/audio_data.py:21-24: uploaded as it is -- half the bytes of the float signal --
        and converted on the device) or float32; a numpy array or a CPU tensor (pinned memory makes the uploads asynchronous).
        The recording is cut into spans of `windows_per_chunk` windows (default max_batch: the batches predict_device runs, so the
        results are ITS bits); the next span is uploaded on a copy stream while a span's log-mel frames and forwards run.  The SHORT
        span (the recording's last W mod windows_per_chunk windows) goes first: the upload nothing can hide is the smallest one.
        `ramp=True` starts with spans of 32, 64, 128 ... windows instead: the exposed upload shrinks to an eighth (1 h fp32s: 10.20 ->
        9.96 ms against 9.75 device-resident; bf16: no gain, its small batches run the slower kernels -- scripts/ubench/
        host_pipeline_parts.py), but the batches are no longer predict_device's: the same bits only where a window's result does
        not depend on its batch (fp32 / fp32s: measured equal; bf16 needs model.batch_invariant).  Returns the per-frame
        probabilities [N] on the device."""
        from .features import log_mel_span, pcm16_to_f32, span_samples

        lib = _lib.load()
        src = self._host_source(audio)
        n = int(src.shape[0])
        N = 1 + n // 160
        T, hop = self.T, self.hop
        W = lib.savad_stream_window_count(N, T, hop)
        if W < 0:
            _lib.check(W)
        per = int(windows_per_chunk or self.max_batch)
        bounds, lo = [], 0
        if ramp:   # spans of 32, 64, 128 ... windows up to `per`: the first upload -- the one nothing hides -- is an eighth of a full span's
            step = 32
            while lo < W and step < per:
                bounds.append((lo, min(W, lo + step)))
                lo, step = lo + step, 2 * step
        bounds += [(b, min(W, b + per)) for b in range(lo, W, per)]
        plan = []
        for lo, hi in bounds:
            f0, f1 = hop * lo, min(N, hop * (hi - 1) + T)
            first, count = span_samples(n, f0, f1 - f0)
            plan.append((lo, hi, f0, f1, first, count))
        if not ramp and len(plan) > 1 and plan[-1][1] - plan[-1][0] < per:
            plan.insert(0, plan.pop())
        self.model.eval()
        dev = self.device if self.device.index is not None else torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            stream = ctypes.c_void_p(cur.cuda_stream)
            if self._pipe is None or self._pipe.model is not self.model or self._pipe.depth != max(self.in_flight, 1):
                from .pipeline import PipelinedVAD
                self._pipe = PipelinedVAD(self.model, depth=max(self.in_flight, 1))
            if getattr(self, "_copy_stream", None) is None or self._copy_stream.device != dev:
                self._copy_stream = torch.cuda.Stream(dev)
            cs = self._copy_stream
            dev_audio = torch.empty(n, dtype=src.dtype, device=dev)
            dev_audio.record_stream(cs)
            cs.wait_stream(cur)
            logp = torch.empty((W, T, 2), dtype=torch.float32, device=dev)
            uploaded, tail_from = 0, n   # on the device so far: samples [0, uploaded) and [tail_from, n)

            def upload(c):   # everything span c reads that is not on the device yet
                nonlocal uploaded, tail_from
                a, b = plan[c][4], plan[c][4] + plan[c][5]
                ev = torch.cuda.Event()
                with torch.cuda.stream(cs):
                    if uploaded == 0 and a > 0 and c == 0:   # the short last span, taken first
                        dev_audio[a:b].copy_(src[a:b], non_blocking=True)
                        tail_from = a
                    else:
                        a, b = max(a, uploaded), min(b, tail_from)
                        if b > a:
                            dev_audio[a:b].copy_(src[a:b], non_blocking=True)
                        uploaded = max(uploaded, b)
                    ev.record(cs)
                return ev

            ev = upload(0)
            for c, (lo, hi, f0, f1, first, count) in enumerate(plan):
                nxt = upload(c + 1) if c + 1 < len(plan) else None   # (pageable memory: this call stages synchronously -- while span c - 1 still runs)
                cur.wait_event(ev)
                sl = dev_audio[first:first + count]
                if sl.dtype == torch.int16:
                    sl = pcm16_to_f32(sl)
                feat = log_mel_span(sl, first, n, f0, f1 - f0)
                self._windows_logp(feat, f0, N, lo, hi, stream, out=logp[lo:hi], join=False)   # (spans overlap on the pipeline's streams)
                ev = nxt
            self._pipe.join()
            probs = torch.empty((N,), dtype=torch.float32, device=dev)
            _lib.check(lib.savad_overlap_merge(ctypes.c_void_p(logp.data_ptr()), W, N, T, hop, ctypes.c_void_p(probs.data_ptr()), stream))
        return probs

    def predict(self, feature) -> np.ndarray:
        return self.predict_device(feature).cpu().numpy()
