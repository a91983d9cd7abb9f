"""Batch sharding across the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards embarrassingly: every sequence / window is independent (no cross-batch op:
LayerNorm is per row, softmax per sequence), weights (2.4 MB) are
replicated, and the only exchange is ONE all_gather of the fp32 log-probabilities
``[B/N, T, 2]`` at the end (205 KB per rank at B=32, T=800).  The reference does nothing here
for inference (single device, ``vad/predict.py:26-29``); this is new design, not a port.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

# how many collectives this process has issued through this module, by kind: every rank of a job must show the same
# numbers (tests/test_dist_gloo.py compares them across the ranks of a world-size-2 bench.py run)
_COUNTS: Dict[str, int] = {"all_gather": 0, "broadcast": 0, "all_reduce": 0, "barrier": 0}


def collective_counts() -> Dict[str, int]:
    return dict(_COUNTS)


def _all_gather(out: torch.Tensor, local: torch.Tensor, group=None) -> None:
    _COUNTS["all_gather"] += 1
    dist.all_gather_into_tensor(out, local, group=group)


def barrier(group=None) -> None:
    """dist.barrier when a process group exists (counted), nothing otherwise"""
    if dist.is_initialized():
        _COUNTS["barrier"] += 1
        dist.barrier(group=group)


def agree(value: float, device, group=None) -> float:
    """rank 0's `value` on every rank (decisions that steer code paths containing collectives -- how many timed blocks to
    run, how many forwards to keep in flight -- must be taken once per job); the value itself without a process group"""
    if not dist.is_initialized():
        return value
    _COUNTS["broadcast"] += 1
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.broadcast(t, 0, group=group)
    return float(t.item())


def max_over_ranks(values: Sequence[float], device, group=None) -> List[float]:
    """element-wise maximum over the ranks (a job is as slow as its slowest rank)"""
    if not dist.is_initialized():
        return [float(v) for v in values]
    _COUNTS["all_reduce"] += 1
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return [float(v) for v in t.tolist()]


def shard_bounds(batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of `batch` sequences: rank r owns [lo, hi); sizes differ by at most 1."""
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def forward_sharded(forward: Callable[[torch.Tensor], torch.Tensor], features: torch.Tensor,
                    group=None) -> torch.Tensor:
    """`features` is the GLOBAL batch [B, T, F] (same on every rank, or at least this rank's
    slice valid); every rank evaluates its shard with `forward` and receives the full
    [B, T, 2] log-probabilities.  One collective per call (also with a single rank whenever a process
    group exists, so that the code path on an N-GPU node is the one a 1-GPU run exercises)."""
    B, T = features.shape[0], features.shape[1]
    return sharded_rows(B, lambda lo, hi: forward(features[lo:hi]), (T, 2), torch.float32, features.device, group)


def sharded_rows(total_rows: int, compute: Callable[[int, int], torch.Tensor], tail: Tuple[int, ...], dtype, device,
                 group=None) -> torch.Tensor:
    """The whole multi-GPU pattern of this path: rows [0, total_rows) (sequences of a batch, windows of a long
    recording) are split contiguously over the ranks (shard_bounds), rank r evaluates `compute(lo, hi)` ->
    [hi - lo, *tail] for its own span only, and ONE all_gather hands every rank all rows.  Without an initialised
    process group it is just compute(0, total_rows)."""
    if not dist.is_initialized():
        return compute(0, total_rows)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(total_rows, rank, world)
    local = compute(lo, hi) if hi > lo else torch.zeros((0,) + tuple(tail), dtype=dtype, device=device)
    return all_gather_rows(local, total_rows, group)


def all_gather_rows(local: torch.Tensor, total_rows: int, group=None) -> torch.Tensor:
    """all_gather of contiguous row shards of possibly unequal size (sizes follow shard_bounds):
    shards are padded to the largest one so that a single all_gather_into_tensor suffices."""
    world = dist.get_world_size(group)
    per = -(-total_rows // world)  # ceil
    tail = tuple(local.shape[1:])
    if local.shape[0] == per and local.is_contiguous():
        padded = local
    else:
        padded = local.new_zeros((per,) + tail)
        padded[: local.shape[0]] = local
    gathered = local.new_empty((world * per,) + tail)
    _all_gather(gathered, padded, group)
    if total_rows % world == 0:
        return gathered
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(total_rows, r, world)
        parts.append(gathered[r * per: r * per + (hi - lo)])
    return torch.cat(parts, 0)


class ShardedPipeline:
    """The multi-batch form of :func:`forward_sharded`: every rank pushes its OWN shards of a stream of batches through a
    :class:`~voice_activity_detection_amd.pipeline.PipelinedVAD` (several forwards in flight on the GPU) and the
    log-probabilities travel over RCCL either

    * ``gather="step"`` -- one ``all_gather_into_tensor`` of ``[B, T, 2]`` per forward, issued on the caller's stream as
      soon as the forward ``depth - 1`` submissions back has finished, so the newer forwards keep running underneath it; or
    * ``gather="final"`` -- every forward writes straight into its slot of a ``[K, B, T, 2]`` send buffer and ONE
      ``all_gather_into_tensor`` at :meth:`join` moves all of them (north_star: "a single RCCL gather over xGMI at the end").

    ``submit(x)`` takes this rank's shard ``[B, T, F]`` (same shape on every rank and for every batch between two joins; use
    :func:`forward_sharded` for ragged splits) and returns nothing useful before ``join()``, which returns one
    ``[world, B, T, 2]`` tensor per submitted batch (rank-major: ``out[r]`` is rank r's shard), valid on the caller's
    stream.  Every rank must submit the same number of batches between joins -- collectives are matched by order.
    Without an initialised process group it is the plain pipeline (``world = 1``, no collective).

    ``forward``: a stand-in ``forward(x, out)`` used INSTEAD of the model (CPU dry runs of the control flow under gloo:
    ``bench.py --backend gloo --stub-forward``, tests/test_dist_gloo.py); the product passes a model.
    """

    def __init__(self, model=None, slots: int = 8, depth: Optional[int] = None, gather: str = "step", group=None,
                 forward: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None):
        if gather not in ("step", "final"):
            raise ValueError(f"gather must be 'step' or 'final', got {gather!r}")
        if (model is None) == (forward is None):
            raise ValueError("pass either a model or a stand-in forward")
        if slots < 1:
            raise ValueError("slots must be >= 1")
        self.gather, self.group, self.slots = gather, group, int(slots)
        self._forward = forward
        self.pipe = None
        self._stand_in_depth = self._stand_in_active = max(int(depth or 1), 1)   # a stand-in forward runs synchronously: the
        if model is not None:                                                      # depth only sets how far the gathers lag
            from .pipeline import PipelinedVAD

            self.pipe = PipelinedVAD(model, depth)
        self.distributed = dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self._send: Optional[torch.Tensor] = None
        self._recv: Optional[torch.Tensor] = None
        self._n = 0          # batches submitted since the last join
        self._gathered = 0   # gather="step": how many of them have been gathered already
        self._replica_of: List[int] = []   # which pipeline replica ran batch k

    @property
    def pending(self) -> int:
        """batches submitted since the last join() (at most `slots`)"""
        return self._n

    @property
    def depth(self) -> int:
        return self.pipe.depth if self.pipe is not None else self._stand_in_depth

    @property
    def in_flight(self) -> int:
        return self.pipe.active if self.pipe is not None else self._stand_in_active

    def set_in_flight(self, n: int) -> None:
        """forwards kept in flight from now on (1 .. depth); call it between joins, with the same n on every rank"""
        if self._n:
            raise RuntimeError("set_in_flight between submit and join")
        if self.pipe is not None:
            self.pipe.set_active(n)
        elif not 1 <= n <= self._stand_in_depth:
            raise ValueError(f"in_flight must be in [1, {self._stand_in_depth}], got {n}")
        else:
            self._stand_in_active = int(n)

    def set_gather(self, gather: str) -> None:
        if gather not in ("step", "final"):
            raise ValueError(f"gather must be 'step' or 'final', got {gather!r}")
        if self._n:
            raise RuntimeError("set_gather between submit and join")
        self.gather = gather

    def _buffers(self, x: torch.Tensor) -> None:
        shape = (self.slots, x.shape[0], x.shape[1], 2)
        if self._send is None or tuple(self._send.shape) != shape or self._send.device != x.device:
            if self._n:
                raise ValueError("every batch between two joins must have the same shape")
            self._send = torch.empty(shape, dtype=torch.float32, device=x.device)
            self._recv = torch.empty((self.world * self.slots,) + shape[1:], dtype=torch.float32, device=x.device) if self.distributed else None

    def _gather_one(self, k: int) -> None:
        # gather="step": log-probs of batch k -> rows [k * world, (k + 1) * world) of the receive buffer
        out = self._recv[k * self.world:(k + 1) * self.world]            # [world, B, T, 2]
        _all_gather(out.view((-1,) + tuple(out.shape[2:])), self._send[k], self.group)   # (the concatenating form: gloo knows no other)

    @torch.no_grad()
    def submit(self, features: torch.Tensor) -> None:
        if features.dim() != 3:
            raise ValueError(f"features must be [B, T, F], got {tuple(features.shape)}")
        self._buffers(features)
        if self._n >= self.slots:
            raise RuntimeError(f"{self.slots} batches are waiting for join(): build the pipeline with more slots")
        out = self._send[self._n]
        if self.pipe is not None:
            self.pipe.submit(features, out=out)
            self._replica_of.append(self.pipe.last_replica)
        else:
            self._forward(features, out)
            self._replica_of.append(0)
        self._n += 1
        if self.distributed and self.gather == "step":
            lag = self.in_flight - 1   # the forward that many submissions back has had the others' time to finish
            while self._n - self._gathered > lag:
                self._ready(self._gathered)
                self._gather_one(self._gathered)
                self._gathered += 1

    def _ready(self, k: int) -> None:
        """order the caller's stream after forward k.  Waiting for its replica's stream is waiting for exactly that forward:
        replicas are used round-robin and at most `in_flight - 1` newer batches have been submitted when this runs"""
        if self.pipe is not None:
            self.pipe.wait_for_replica(self._replica_of[k])

    def join(self) -> List[torch.Tensor]:
        """-> [world, B, T, 2] per batch submitted since the last join, in submission order.  The tensors are views of the
        pipeline's own buffers: consume (or clone) them before the next submit()."""
        n = self._n
        if n == 0:
            return []
        if self.pipe is not None:
            self.pipe.join()
        if not self.distributed:
            outs = [self._send[k].unsqueeze(0) for k in range(n)]
        elif self.gather == "step":
            while self._gathered < n:
                self._gather_one(self._gathered)
                self._gathered += 1
            outs = [self._recv[k * self.world:(k + 1) * self.world] for k in range(n)]
        else:
            flat = self._recv[: self.world * n]
            _all_gather(flat, self._send[:n], self.group)          # rank-major: [world][n][B][T][2]
            whole = flat.view((self.world, n) + tuple(self._send.shape[1:]))
            outs = [whole[:, k] for k in range(n)]
        self._n = self._gathered = 0
        self._replica_of = []
        return outs


def forward_sharded_many(pipeline: ShardedPipeline, batches: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """GLOBAL batches ``[B, T, F]`` (all of one shape, the same on every rank) -> their full ``[B, T, 2]`` log-probabilities on
    every rank, through `pipeline`: each rank evaluates rows ``shard_bounds(B, rank, world)`` of every batch (shards are
    padded to the largest one, so any B works), ``pipeline.slots`` batches per join."""
    if not batches:
        return []
    world, rank = pipeline.world, pipeline.rank
    B = batches[0].shape[0]
    per = -(-B // world)
    lo, hi = shard_bounds(B, rank, world)
    results: List[torch.Tensor] = []
    for i0 in range(0, len(batches), pipeline.slots):
        chunk = batches[i0:i0 + pipeline.slots]
        for x in chunk:
            if tuple(x.shape) != tuple(batches[0].shape):
                raise ValueError("forward_sharded_many wants batches of one shape")
            shard = x[lo:hi]
            if hi - lo < per:   # pad with copies of the first row: any finite input will do, the rows are dropped below
                pad = x[:1].expand(per - (hi - lo), -1, -1)
                shard = torch.cat([shard, pad], 0)
            pipeline.submit(shard.contiguous())
        for g in pipeline.join():        # [world, per, T, 2]
            if B % world == 0:
                results.append(g.reshape(B, g.shape[2], 2).clone())   # (join() hands out views of buffers the next chunk reuses)
            else:
                parts = []
                for r in range(world):
                    rlo, rhi = shard_bounds(B, r, world)
                    parts.append(g[r, : rhi - rlo])
                results.append(torch.cat(parts, 0))
    return results
