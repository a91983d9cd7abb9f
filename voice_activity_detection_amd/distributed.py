"""Batch sharding across the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards embarrassingly: every sequence / window is independent (no cross-batch op:
LayerNorm is per row, softmax per sequence), weights (2.4 MB) are
replicated, and the only exchange is ONE all_gather of the fp32 log-probabilities
``[B/N, T, 2]`` at the end (205 KB per rank at B=32, T=800).  The reference does nothing here
for inference (single device, ``vad/predict.py:26-29``); this is new design, not a port.
"""
from __future__ import annotations

from typing import Callable, Tuple

import torch
import torch.distributed as dist


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
    dist.all_gather_into_tensor(gathered, padded, group=group)
    if total_rows % world == 0:
        return gathered
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(total_rows, r, world)
        parts.append(gathered[r * per: r * per + (hi - lo)])
    return torch.cat(parts, 0)
