"""Several independent forwards in flight on one GPU.

One forward at a time leaves matrix-core time unused that no kernel change recovers: at BASELINE configs[1]
(``[32, 800, 80]`` fp32) the fused launches put 800 wave-sized work items on 1 024 SIMDs, the input stage 200 workgroups on
256 CUs, and every bf16 launch ends in a partial round.  Batches are independent (``vad/predictor.py:180-224`` walks them in a
Python loop; north_star shards them), so the idle slots can run the NEXT batch: :class:`PipelinedVAD` keeps ``depth`` forwards
in flight, each on its own HIP stream with its own library handle and workspace (the parameters are shared with the module
it was built from; a handle's folded device copy of them is 2.4 MB).  Results are what the module itself returns, bit for bit;
only the throughput changes (measured, one MI355X, forwards per second: ``[32,800,80]`` fp32 +25 % at depth 3, ``[256,800,80]``
bf16 +5 % at depth 2; ``scripts/ubench/pipelined_streams.py``).  Latency of a single batch grows accordingly -- this is for
loops over many batches (window chunks of a long recording, streaming windows, a serving queue).
"""
from __future__ import annotations

import copy
from typing import Iterable, List, Optional

import torch
from torch import Tensor

from .model import SelfAttentiveVAD

_KNOBS = ("precision", "row_mode", "attention_splits", "batch_invariant", "training")


class PipelinedVAD:
    """``depth`` replicas of ``model`` (shared parameters, private native handle / workspace / HIP stream each).

    ``submit(x)`` enqueues one forward on the next replica's stream and returns its output tensor immediately; the tensor's
    contents are valid once ``join()`` (or a synchronisation of the device) has run.  Inputs are ordered after whatever the
    CURRENT stream did before ``submit``; ``join()`` orders the current stream after every submitted forward."""

    def __init__(self, model: SelfAttentiveVAD, depth: Optional[int] = None):
        if depth is None:
            depth = 2 if model.precision == "bf16" else 3
        if depth < 1:
            raise ValueError(f"depth must be >= 1, got {depth}")
        self.model = model
        self.depth = int(depth)
        # the base module is never used by the pipeline itself (depth > 1): calling it directly stays safe while forwards are in flight
        self._replicas: List[SelfAttentiveVAD] = [model] if self.depth == 1 else [copy.copy(model) for _ in range(self.depth)]
        self._streams: Optional[List[torch.cuda.Stream]] = None
        self._next = 0
        self._busy: List[bool] = [False] * self.depth
        self.active = self.depth   # forwards kept in flight (<= depth): set_active() lets a caller tune it for its shape
        self.last_replica = 0      # index of the replica / stream the latest submit() used (0 when it ran on the caller's stream)

    def set_active(self, n: int) -> None:
        """use only the first n replicas from now on (1 <= n <= depth); call it with nothing in flight"""
        if not 1 <= n <= self.depth:
            raise ValueError(f"active must be in [1, {self.depth}], got {n}")
        self.join()
        self.active, self._next = int(n), 0

    def _ensure_streams(self, device: torch.device) -> List[torch.cuda.Stream]:
        if device.type == "cuda" and device.index is None:   # "cuda" -> the indexed device the streams report
            device = torch.device("cuda", torch.cuda.current_device())
        if self._streams is None or self._streams[0].device != device:
            self.join()   # forwards still in flight belong to the streams about to be replaced
            self._streams = [torch.cuda.Stream(device) for _ in range(self.depth)]
        return self._streams

    @torch.no_grad()
    def submit(self, features: Tensor, out: Optional[Tensor] = None) -> Tensor:
        """enqueue ``model(features=features, out=out)`` on the next replica; returns the (not yet valid) output tensor"""
        device = features.device
        if device.type != "cuda":
            return self.model(features=features, out=out)  # raises the module's own "no CPU fallback" error
        if self.active == 1:   # nothing to overlap: the plain module on the caller's stream (after whatever is still in flight)
            self.join()
            self.last_replica = 0
            return self.model(features=features, out=out)
        streams = self._ensure_streams(device)
        k = self._next
        self._next = (k + 1) % self.active
        self.last_replica = k
        rep, s = self._replicas[k], streams[k]
        self._follow(rep)
        s.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(s):
            y = rep(features=features, out=out)
        # the caching allocator must not hand these blocks to another stream while this forward still uses them
        features.record_stream(s)
        y.record_stream(s)
        self._busy[k] = True
        return y

    @torch.no_grad()
    def submit_windows(self, feature: Tensor, T: int, hop: int, first: int, count: int, out: Optional[Tensor] = None) -> Tensor:
        """``submit`` for ``model.forward_windows``: windows read in place out of a feature matrix"""
        device = feature.device
        if device.type != "cuda" or self.active == 1:
            if device.type == "cuda":
                self.join()
            self.last_replica = 0
            return self.model.forward_windows(feature, T, hop, first, count, out=out)
        streams = self._ensure_streams(device)
        k = self._next
        self._next = (k + 1) % self.active
        self.last_replica = k
        rep, s = self._replicas[k], streams[k]
        self._follow(rep)
        s.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(s):
            y = rep.forward_windows(feature, T, hop, first, count, out=out)
        feature.record_stream(s)
        y.record_stream(s)
        self._busy[k] = True
        return y

    def _follow(self, rep: SelfAttentiveVAD) -> None:
        """knobs set on the base module after construction apply to every replica, and so does a declared weight change
        (model.sync_weights(force=True), a .train() / .eval() switch, .to()): the replica re-pushes at its next forward"""
        base = self.model
        if rep is base:
            return
        for name in _KNOBS:
            if getattr(rep, name) != getattr(base, name):
                setattr(rep, name, getattr(base, name))
        if rep._weights_generation != base._weights_generation:
            rep._pdev = None   # (the base may have moved: the replica re-reads its parameters' device at its next forward)
        rep._weights_generation = base._weights_generation

    def wait_for_replica(self, k: int) -> None:
        """the current stream waits for everything submitted to replica k so far"""
        if self._streams is not None and self.active > 1:
            torch.cuda.current_stream(self._streams[k].device).wait_stream(self._streams[k])

    def join(self) -> None:
        """the current stream waits for every forward submitted so far"""
        if self._streams is None:
            return
        cur = torch.cuda.current_stream(self._streams[0].device)
        for k, s in enumerate(self._streams):
            if self._busy[k]:
                cur.wait_stream(s)
                self._busy[k] = False

    @torch.no_grad()
    def forward_many(self, batches: Iterable[Tensor]) -> List[Tensor]:
        """all batches through the pipeline; the returned outputs are ordered like the inputs and valid on the current stream"""
        outs = [self.submit(x) for x in batches]
        self.join()
        return outs

    def reserve(self, max_frames: int, device=None, max_batch: int = 0) -> None:
        """``SelfAttentiveVAD.reserve`` on every replica (handles, folded weights, positional-encoding tables, workspaces)"""
        dev = torch.device(device) if device is not None else self.model.classifier.weight.device
        streams = self._ensure_streams(dev) if (dev.type == "cuda" and self.depth > 1) else None
        for k, rep in enumerate(self._replicas):
            self._follow(rep)
            if streams is None:
                rep.reserve(max_frames, device=device, max_batch=max_batch)
                continue
            # on the replica's OWN stream: the workspace it caches here is the one its forwards use there (a block allocated on the
            # caller's stream could be handed out again while a side-stream forward still reads it)
            streams[k].wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(streams[k]):
                rep.reserve(max_frames, device=device, max_batch=max_batch)
            torch.cuda.current_stream(dev).wait_stream(streams[k])
