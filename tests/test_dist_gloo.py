"""N > 1 path on CPU: world_size-2 gloo processes run the batch-shard + single all_gather logic of
voice_activity_detection_amd.distributed around a CPU stand-in forward (the stock-PyTorch port from
oracle/, test infrastructure) and must reproduce the unsharded result exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from voice_activity_detection_amd.distributed import shard_bounds


def test_shard_bounds_cover_batch():
    for B in (0, 1, 2, 7, 32, 33, 2048):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, batch, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import torch_port
    from voice_activity_detection_amd.distributed import forward_sharded
    from voice_activity_detection_amd.seeded import seeded_features, seeded_state_dict

    state = {k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()}
    x = torch.from_numpy(seeded_features(3, (batch, 9, 80)))
    calls = []

    def fwd(t):
        calls.append(t.shape[0])
        return torch_port.forward(state, t)

    y = forward_sharded(fwd, x)
    np.save(os.path.join(out_dir, f"y{rank}.npy"), y.numpy())
    np.save(os.path.join(out_dir, f"n{rank}.npy"), np.array(calls))
    dist.destroy_process_group()


@pytest.mark.parametrize("batch", [6, 5, 1])
def test_two_rank_gloo_matches_unsharded(tmp_path, batch):
    from oracle import torch_port
    from voice_activity_detection_amd.seeded import seeded_features, seeded_state_dict

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), batch, str(tmp_path)), nprocs=world, join=True)
    state = {k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()}
    ref = torch_port.forward(state, torch.from_numpy(seeded_features(3, (batch, 9, 80)))).numpy()
    for r in range(world):
        y = np.load(tmp_path / f"y{r}.npy")
        assert y.shape == ref.shape
        assert np.abs(y - ref).max() < 1e-6  # same ATen ops on a sub-batch
    sizes = [int(np.load(tmp_path / f"n{r}.npy").sum()) for r in range(world)]
    assert sum(sizes) == batch and max(sizes) - min(sizes) <= 1
