"""`SelfAttentiveVAD` -- the reference's nn.Module surface over the MI355X forward pass.

Drop-in for ``vad.models.self_attention.SelfAttentiveVAD`` (reference
``vad/models/self_attention.py:6-28``): same constructor signature
``(feature_size, num_layers, d_model, dropout)`` (``vad/models/model_factory.py:42-48``), same
``state_dict`` keys (so ``load_state_dict(checkpoint["state_dict"])`` of
``vad/predictor.py:278`` is strict-clean), same call forms ``model(features=x)``
(``vad/predictor.py:224``) and ``model(x)`` (``vad/model_runner.py:32``), same result: a
contiguous fp32 ``[B, T, 2]`` tensor of log-probabilities on the input's device.

The submodules below are PARAMETER CONTAINERS only (they give the state_dict its reference
key names); ``forward`` never calls them.  It hands raw device pointers to ``savad_forward``
in ``libsavad.so`` (C ABI ``include/savad.h``), which runs hand-written gfx950 kernels on the
caller's current HIP stream.  There is no CPU / eager fallback: a non-GPU input raises.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch
from torch import Tensor, nn

from . import _lib


class _Holder(nn.Module):
    """Namespace module: only holds children so that state_dict keys match the reference."""


def _attention_params(d_model: int) -> nn.Module:
    m = _Holder()  # reference: MultiHeadAttention, vad/modeling/transformer.py:241-252
    m.query_projection = nn.Linear(d_model, d_model)
    m.key_projection = nn.Linear(d_model, d_model)
    m.value_projection = nn.Linear(d_model, d_model)
    m.final_projection = nn.Linear(d_model, d_model)
    return m


def _sublayer_params(d_model: int) -> nn.Module:
    m = _Holder()  # reference: Sublayer, vad/modeling/transformer.py:227-232
    m.layer_norm = nn.LayerNorm(d_model)
    return m


def _ffn_params(d_model: int, d_ff: int) -> nn.Module:
    m = _Holder()  # reference: PositionwiseFeedForwardNetwork, vad/modeling/transformer.py:366-375
    # indices 0 and 3 carry parameters (1 = ReLU, 2 = Dropout in the reference Sequential)
    m.feed_forward = nn.ModuleDict({"0": nn.Linear(d_model, d_ff), "3": nn.Linear(d_ff, d_model)})
    return m


def _layer_params(d_model: int, d_ff: int) -> nn.Module:
    m = _Holder()  # reference: TransformerEncoderLayer, vad/modeling/transformer.py:37-47
    m.self_attention = _attention_params(d_model)
    m.self_attention_sublayer = _sublayer_params(d_model)
    m.feed_forward = _ffn_params(d_model, d_ff)
    m.feed_forward_sublayer = _sublayer_params(d_model)
    return m


PRECISIONS = {"fp32": 0, "bf16": 1, "fp32s": 2}   # savad_set_precision codes (include/savad.h)


class SelfAttentiveVAD(nn.Module):
    def __init__(self, feature_size: int, num_layers: int, d_model: int, dropout: float = 0.0):
        super().__init__()
        self.feature_size = int(feature_size)
        self.num_layers = int(num_layers)
        self.d_model = int(d_model)
        self.dropout_p = float(dropout)  # inference path: Dropout is the identity in .eval()
        d_ff = 4 * d_model  # vad/models/self_attention.py:10

        self.input_layer = nn.ModuleDict({"0": nn.Linear(feature_size, d_model)})  # key "input_layer.0.*"
        encoder = _Holder()
        encoder.layers = nn.ModuleList([_layer_params(d_model, d_ff) for _ in range(num_layers)])
        encoder.layer_norm = nn.LayerNorm(d_model)
        self.encoder = encoder
        self.classifier = nn.Linear(d_model, 2)

        self._handle: Optional[ctypes.c_void_p] = None
        self._handle_device: Optional[torch.device] = None
        self._synced_versions = None
        self._workspace: Optional[Tensor] = None
        self._param_dicts = None      # the `_parameters` dicts of the leaf modules, collected once (the module tree is fixed)
        self._pushed_knobs = None     # (attention_splits, row_mode, precision, batch_invariant) the handle was last told
        self._ws_bytes = {}           # (B, T, knobs) -> savad_workspace_bytes
        # bumped whenever the caller declares the weights changed behind autograd's back (sync_weights(force=True), a mode switch):
        # PipelinedVAD's replicas share the parameters but keep their own handles, and re-push when they see a new generation
        self._weights_generation = 0
        self._seen_generation = 0
        self._pdev = None             # device of the parameters (cached: nn.Module attribute lookups are slow; _apply resets it)
        self.attention_splits = 0  # 0 = automatic
        self.row_mode = 0  # 0 = automatic, 1 = N-split 32-row tiles, 2 / 3 = M-split 128-row tiles, 4 = T <= 32 in one launch (include/savad.h)
        # "fp32": exact-fp32 MFMA (default, log-probs within 1e-4 of the reference).
        # "bf16": bf16 MFMA operands, fp32 accumulation / statistics, fp16-stored residual stream (BASELINE configs[2..3]).
        # "fp32s": fp32 parity on the bf16 matrix pipe -- every GEMM operand as three bf16 pieces, six MFMA products per
        #          K-step into an fp32 accumulator (csrc/savad_kernels_f32s.h): the same 1e-4 bar as "fp32", 6/16 of its matrix time.
        self.precision = "fp32"
        # bf16 only: the same bits for a sequence whatever batch it is evaluated in (chunk sizes, shard sizes, remainders), at 3 - 4 % of a
        # large-batch forward: the persistent attention kernel then runs without its key-split tail items (include/savad.h)
        self.batch_invariant = False

    # ---- library handle / weights -----------------------------------------------------------
    def _ensure_handle(self, device: torch.device):
        if self._handle is not None and self._handle_device == device:
            return
        self._release()
        lib = _lib.load()
        cfg = _lib.savad_config(self.feature_size, self.num_layers, self.d_model)
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.savad_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._handle, self._handle_device = h, device
        self._synced_versions = None
        self._pushed_knobs = None

    def _release(self):
        if self._handle is not None:
            _lib.load().savad_destroy(self._handle)
            self._handle = None
            self._synced_versions = None
            self._pushed_knobs = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # The library handle, its device and the cached workspace are per-process runtime state: copies and pickles of
    # the module (copy.copy, copy.deepcopy, torch.save(model)) drop them and recreate them lazily -- a copy never
    # shares (and so never double-frees) the original's native handle.
    _RUNTIME_ATTRS = ("_handle", "_handle_device", "_synced_versions", "_workspace", "_param_dicts", "_pushed_knobs")

    def __copy__(self):
        # explicit, so that a shallow copy never depends on which pickling protocol copy.copy() happens to use
        new = self.__class__.__new__(self.__class__)
        new.__dict__.update(self.__dict__)
        for name in self._RUNTIME_ATTRS:
            new.__dict__[name] = None
        new.__dict__["_ws_bytes"] = {}
        return new

    def _replicate_for_data_parallel(self):
        # nn.DataParallel replicas are shallow __dict__ copies WITHOUT parameters (vad/training/trainer.py:115-116 is the
        # reference's only use, in training -- out of scope): a replica would share this module's native handle, destroy
        # it when it moves to its own device, and find no state_dict to push.  Multi-GPU inference here is one process
        # per GPU (voice_activity_detection_amd.distributed.forward_sharded), as north_star asks.
        raise _lib.SavadError(
            "nn.DataParallel is not supported by the MI355X SelfAttentiveVAD: run one process per GPU and use "
            "voice_activity_detection_amd.distributed.forward_sharded (one RCCL all_gather at the end)")

    def __getstate__(self):
        state = dict(self.__dict__)
        for name in self._RUNTIME_ATTRS:
            state[name] = None
        state["_ws_bytes"] = {}
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        for name in self._RUNTIME_ATTRS:
            self.__dict__.setdefault(name, None)
        self.__dict__.setdefault("_ws_bytes", {})
        self.__dict__.setdefault("_weights_generation", 0)
        self.__dict__.setdefault("_seen_generation", 0)
        self.__dict__["_pdev"] = None

    def train(self, mode: bool = True):
        # SWITCHING modes is where parameters were most likely edited behind autograd's back (p.data.copy_ in EMA /
        # init code leaves data_ptr and _version unchanged): re-push the weights at the next forward.  A call that
        # changes nothing (the predictor's model.eval() before every batch) must cost nothing: the re-push is 54 copies
        # plus the fold / pack kernels, ~0.4 ms against a 0.09 ms forward.
        if bool(mode) != self.training:
            self._synced_versions = None
            self._weights_generation += 1
        return super().train(mode)  # always: children must follow the parent's mode even when it did not change

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float() ...: storage moves without a version bump -- re-push (here and in a pipeline's replicas)
        self._synced_versions = None
        self._weights_generation += 1
        self._pdev = None
        return super()._apply(fn, *args, **kwargs)

    def _param_versions(self):
        """(identity, _version) of every parameter, read from the leaf modules' own `_parameters` dicts: a replaced Parameter
        object is seen as well as an in-place update (storage moves are announced by _apply), and the walk costs ~8 us where
        (data_ptr, _version) over `self.parameters()` cost ~130 (scripts/ubench/host_overhead.py) -- this runs on every forward."""
        dicts = self._param_dicts
        if dicts is None:
            dicts = self._param_dicts = [m._parameters for m in self.modules() if m._parameters]
        ps = [p for d in dicts for p in d.values()]
        # data_ptr: `p.data = other_tensor` keeps identity and version but moves the storage (cheap once the dicts are cached: ~5 us)
        return tuple(map(id, ps)), [p._version for p in ps], [p.data_ptr() for p in ps]

    def __setattr__(self, name, value):
        # a replaced SUBMODULE (model.classifier = nn.Linear(...)) after the first forward: the cached walk above would keep serving
        # the old module's parameters
        # (the generation bump carries the change to a PipelinedVAD's replicas: shallow copies that share _modules but hold their own
        # cached walk; an assignment into a nested container -- model.layers[i] = ... -- bypasses this hook: sync_weights(force=True))
        if isinstance(value, nn.Module) and "_param_dicts" in self.__dict__:
            self.__dict__["_param_dicts"] = None
            self.__dict__["_synced_versions"] = None
            self.__dict__["_weights_generation"] = self.__dict__.get("_weights_generation", 0) + 1
        super().__setattr__(name, value)

    def sync_weights(self, force: bool = False):
        """Push the module's parameters into the library's packed weight store when they changed.  Changes are detected
        through object identity + _version (+ the module's own _apply): load_state_dict, .to(), optimizer steps, in-place
        tensor ops and replaced Parameter objects are all caught.
        NOT caught: writes through `.data` / `.detach()` views (p.data.mul_(...), p.data.copy_(...)), which bump neither --
        call `model.sync_weights(force=True)` after such edits (a `.train()` / `.eval()` that switches the mode also forces a re-push).
        One module instance = one library handle + one cached workspace: use it from one stream at a time."""
        if force:
            self._weights_generation += 1
        if self._seen_generation != self._weights_generation:   # (a replica of a PipelinedVAD sees the base module's counter)
            self._seen_generation = self._weights_generation
            self._synced_versions = None
            self._param_dicts = None   # (a replaced submodule on the base module: this replica's cached walk is of the old one)
        if self._handle is None:   # a forced re-push on a module that has not run yet (its replicas may have): nothing to push to yet
            if force:
                pdev = self.classifier.weight.device
                if pdev.type != "cuda":
                    return
                self._ensure_handle(pdev)
            else:
                return
        versions = self._param_versions()
        if not force and versions == self._synced_versions:
            return
        lib = _lib.load()
        stream = torch.cuda.current_stream(self._handle_device).cuda_stream
        for key, p in self.state_dict(keep_vars=True).items():
            t = p.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.to(torch.float32).contiguous()
            _lib.check(lib.savad_set_param(self._handle, key.encode(), ctypes.c_void_p(t.data_ptr()), t.numel(),
                                           ctypes.c_void_p(stream)))
            if t.device.type != "cuda":
                torch.cuda.current_stream(self._handle_device).synchronize()  # host source must outlive the copy
        self._synced_versions = versions

    # ---- forward ------------------------------------------------------------------------------
    def _prepare_call(self, device: torch.device):
        """Shared entry checks of forward / predict_windows; call inside `with torch.cuda.device(device)`."""
        if device.type != "cuda":
            raise _lib.SavadError(
                "SelfAttentiveVAD (MI355X build) runs only on a HIP device tensor; there is no CPU fallback. "
                "Move the model and the features to the GPU (model.to('cuda'), features.to('cuda')).")
        if self.training and self.dropout_p > 0:
            raise _lib.SavadError("training-mode dropout is outside this build's scope: call model.eval()")
        if self.precision not in PRECISIONS:
            raise ValueError(f"precision must be 'fp32', 'fp32s' or 'bf16', got {self.precision!r}")
        pdev = self._pdev
        if pdev is None:
            pdev = self._pdev = self.classifier.weight.device
        if pdev != device:
            raise _lib.SavadError(f"model parameters are on {pdev}, features on {device}")
        lib = _lib.load()
        self._ensure_handle(device)
        self.sync_weights()
        knobs = (int(self.attention_splits), int(self.row_mode), self.precision, bool(getattr(self, "batch_invariant", False)))
        if knobs != self._pushed_knobs:   # four library calls only when a knob moved, none on the steady path
            _lib.check(lib.savad_set_attention_splits(self._handle, knobs[0]))
            _lib.check(lib.savad_set_row_mode(self._handle, knobs[1]))
            _lib.check(lib.savad_set_precision(self._handle, PRECISIONS[knobs[2]]))
            _lib.check(lib.savad_set_batch_invariant(self._handle, int(knobs[3])))
            self._pushed_knobs = knobs
        return lib

    def _workspace_bytes(self, lib, B: int, T: int) -> int:
        key = (B, T, self._pushed_knobs)
        n = self._ws_bytes.get(key)
        if n is None:
            nbytes = ctypes.c_size_t()
            _lib.check(lib.savad_workspace_bytes(self._handle, B, T, ctypes.byref(nbytes)))
            if len(self._ws_bytes) > 256:
                self._ws_bytes.clear()
            n = self._ws_bytes[key] = nbytes.value
        return n

    def _workspace_for(self, nbytes: int, device: torch.device) -> Tensor:
        ws = self._workspace
        if ws is None or ws.device != device or ws.numel() < nbytes:
            if ws is not None and ws.device == device:
                # forwards enqueued on this stream may still read the old block: the allocator must not hand it to another stream first
                ws.record_stream(torch.cuda.current_stream(device))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)  # caching allocator owns it
            self._workspace = ws
        return ws

    def forward(self, features: Tensor, out: Optional[Tensor] = None) -> Tensor:
        """features [B, T, F] -> log-probabilities [B, T, 2] (fp32, on features.device).  `out` (optional, not in the
        reference's signature) is a contiguous fp32 [B, T, 2] tensor to write into instead of allocating one."""
        if features.dim() != 3 or features.size(2) != self.feature_size:
            raise ValueError(f"features must be [B, T, {self.feature_size}], got {tuple(features.shape)}")
        device = features.device
        if device.type != "cuda" or self.precision not in PRECISIONS:
            self._prepare_call(device)  # raises the matching error
        x = features.detach() if features.requires_grad else features
        x_dtype = 0
        if self.precision == "bf16" and x.dtype == torch.bfloat16:
            x_dtype = 1  # bf16 features are consumed as they are
        elif x.dtype != torch.float32:
            x = x.float()
        if not x.is_contiguous():
            x = x.contiguous()
        B, T, _ = x.shape
        if out is None:
            out = torch.empty((B, T, 2), dtype=torch.float32, device=device)
        elif (tuple(out.shape) != (B, T, 2) or out.dtype != torch.float32 or out.device != device or not out.is_contiguous()):
            raise ValueError(f"out must be a contiguous float32 [{B}, {T}, 2] tensor on {device}")
        if B == 0 or T == 0:
            return out
        if torch.cuda.current_device() == device.index:   # (the context manager costs ~10 us; it is only needed to switch devices)
            self._launch_forward(x, x_dtype, B, T, out, device)
        else:
            with torch.cuda.device(device):
                self._launch_forward(x, x_dtype, B, T, out, device)
        return out

    def _launch_forward(self, x: Tensor, x_dtype: int, B: int, T: int, out: Tensor, device: torch.device) -> None:
        lib = self._prepare_call(device)
        ws = self._workspace_for(self._workspace_bytes(lib, B, T), device)
        stream = torch.cuda.current_stream(device).cuda_stream
        _lib.check(lib.savad_forward_ex(self._handle, ctypes.c_void_p(x.data_ptr()), x_dtype, B, T,
                                        ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                                        ws.numel(), ctypes.c_void_p(stream)))

    @torch.no_grad()
    def forward_windows(self, feature: Tensor, T: int, hop: int, first: int, count: int, out: Optional[Tensor] = None) -> Tensor:
        """Log-probs [count, T, 2] of the sliding windows first .. first + count - 1 of a device feature matrix [N, F] -- window w
        = frames [hop w, hop w + T) -- read IN PLACE (savad_forward_strided: sequences hop * F elements apart), no window copies.
        Every window must lie inside the matrix (hop (first + count - 1) + T <= N); the streaming mode's zero-padded last window
        goes through a copy (StreamingPredictor)."""
        device = feature.device
        if device.type != "cuda":
            self._prepare_call(device)  # raises: no CPU fallback
        if feature.dim() != 2 or feature.shape[1] != self.feature_size or feature.dtype != torch.float32 or not feature.is_contiguous():
            raise ValueError(f"feature must be a contiguous float32 [N, {self.feature_size}] tensor, got {tuple(feature.shape)} {feature.dtype}")
        N, F = feature.shape
        if count < 1 or first < 0 or hop < 1 or T < 1 or hop * (first + count - 1) + T > N:
            raise ValueError(f"windows [{first}, {first + count}) of {T} frames every {hop} do not lie inside {N} frames")
        if out is None:
            out = torch.empty((count, T, 2), dtype=torch.float32, device=device)
        elif tuple(out.shape) != (count, T, 2) or out.dtype != torch.float32 or out.device != device or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous float32 [{count}, {T}, 2] tensor on {device}")
        if T <= 32 or F % 16 or self.d_model != 128:
            # savad_forward_strided reads windows in place only where a kernel takes the sequence stride (T > 32, no feature padding,
            # the d_model = 128 kernels); everything else goes through one gathered copy of the windows and the plain forward
            if F % 4:   # (savad_gather_strided moves 16-byte pieces: odd feature sizes take a strided view's copy)
                win = feature.as_strided((count, T, F), (hop * F, F, 1), first * hop * F).contiguous()
            else:
                win = torch.empty((count, T, F), dtype=torch.float32, device=device)
                with torch.cuda.device(device):
                    lib = self._prepare_call(device)
                    stream = torch.cuda.current_stream(device).cuda_stream
                    _lib.check(lib.savad_gather_strided(ctypes.c_void_p(feature.data_ptr()), N, F, T, hop, first, count,
                                                        ctypes.c_void_p(win.data_ptr()), ctypes.c_void_p(stream)))
            if out.data_ptr() % 16 == 0:
                return self(features=win, out=out)
            out.copy_(self(features=win))   # (a slice of a larger result that starts off a 16-byte boundary: odd T x odd window count)
            return out
        with torch.cuda.device(device):
            lib = self._prepare_call(device)
            ws = self._workspace_for(self._workspace_bytes(lib, count, T), device)
            stream = torch.cuda.current_stream(device).cuda_stream
            x = feature.data_ptr() + 4 * first * hop * F
            _lib.check(lib.savad_forward_strided(self._handle, ctypes.c_void_p(x), 0, count, T, hop * F, ctypes.c_void_p(out.data_ptr()),
                                                 ctypes.c_void_p(ws.data_ptr()), ws.numel(), ctypes.c_void_p(stream)))
        return out

    @torch.no_grad()
    def predict_windows(self, feature: Tensor, half: int, jump: int, chunk: int):
        """The whole of VADFromScratchPredictor.predict_probabilities (vad/predictor.py:159-262) in ONE library call
        (savad_predict_probabilities): feature [N, F] fp32 on the device -> (probs [N, W], mean [N]); the windows are
        read straight out of the feature matrix when the single-launch forward applies."""
        device = feature.device
        if device.type != "cuda":
            self._prepare_call(device)  # raises: no CPU fallback
        feat = feature.detach().float().contiguous()
        if feat.dim() != 2 or feat.shape[1] != self.feature_size:
            raise ValueError(f"feature must be [N, {self.feature_size}], got {tuple(feat.shape)}")
        N = feat.shape[0]
        lib = _lib.load()
        W = lib.savad_window_offsets(int(half), int(jump), None)
        probs = torch.empty((N, W), dtype=torch.float32, device=device)
        mean = torch.empty((N,), dtype=torch.float32, device=device)
        if N == 0:
            return probs, mean
        with torch.cuda.device(device):
            self._prepare_call(device)
            nbytes = ctypes.c_size_t()
            _lib.check(lib.savad_predict_workspace_bytes(self._handle, N, int(half), int(jump), int(chunk), ctypes.byref(nbytes)))
            ws = self._workspace_for(nbytes.value, device)
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(lib.savad_predict_probabilities(self._handle, ctypes.c_void_p(feat.data_ptr()), N, int(half), int(jump),
                                                       int(chunk), ctypes.c_void_p(probs.data_ptr()),
                                                       ctypes.c_void_p(mean.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                                                       ws.numel(), ctypes.c_void_p(stream)))
        return probs, mean

    def reserve(self, max_frames: int, device=None, max_batch: int = 0):
        """Everything a later forward would otherwise do once: pushes the parameters into the library, folds / packs
        them for the selected precision, sizes the positional-encoding table for sequences of up to `max_frames` frames
        (savad_reserve) and -- with `max_batch` -- the cached workspace for a [max_batch, max_frames, F] batch.  Forwards
        with T <= max_frames (and B <= max_batch) then neither allocate nor synchronise nor launch anything but their own
        kernels, so even the FIRST forward can be captured into a HIP graph."""
        device = torch.device(device) if device is not None else self.classifier.weight.device
        if device.type == "cuda" and device.index is None:  # "cuda" -> the indexed device tensors report
            device = torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(device):
            lib = self._prepare_call(device)
            stream = torch.cuda.current_stream(device).cuda_stream
            _lib.check(lib.savad_reserve(self._handle, int(max_frames), ctypes.c_void_p(stream)))
            if max_batch > 0 and max_frames > 0:
                nbytes = ctypes.c_size_t()
                _lib.check(lib.savad_workspace_bytes(self._handle, int(max_batch), int(max_frames), ctypes.byref(nbytes)))
                self._workspace_for(nbytes.value, device)

    def residual_saturations(self) -> int:
        """precision "bf16" stores the residual stream as fp16 between kernels (+-65504): number of elements that had to
        be clamped since the last call (0 on every workload the parity tests know; > 0 means: use precision "fp32").
        Synchronises the current stream."""
        if self._handle is None:
            return 0
        n = ctypes.c_ulonglong()
        with torch.cuda.device(self._handle_device):
            stream = torch.cuda.current_stream(self._handle_device).cuda_stream
            _lib.check(_lib.load().savad_residual_saturations(self._handle, ctypes.byref(n), ctypes.c_void_p(stream)))
        return int(n.value)

    # ---- profiling hooks used by bench.py -------------------------------------------------------
    def set_profiling(self, capacity: int, skip: int = 0):
        """record the kernels of the next `capacity` forwards, after `skip` un-recorded ones (clocks settle)"""
        with torch.cuda.device(self._handle_device):
            _lib.check(_lib.load().savad_set_profiling(self._handle, int(capacity)))
            if skip:
                _lib.check(_lib.load().savad_profiling_skip(self._handle, int(skip)))

    def kernel_times(self):
        """[(kernel name, average ms)] per launch position since set_profiling; sync the stream first."""
        names = (ctypes.c_char_p * 64)()
        ms = (ctypes.c_float * 64)()
        n = _lib.load().savad_last_kernel_times(self._handle, names, ms, 64)
        if n < 0:
            _lib.check(n)
        return [(names[i].decode(), float(ms[i])) for i in range(n)]
