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


def _worker(rank, world, rendezvous, batch, out_dir):
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"file://{rendezvous}", rank=rank, world_size=world)
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
    mp.spawn(_worker, args=(world, str(tmp_path / "rendezvous"), batch, str(tmp_path)), nprocs=world, join=True)
    state = {k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()}
    ref = torch_port.forward(state, torch.from_numpy(seeded_features(3, (batch, 9, 80)))).numpy()
    for r in range(world):
        y = np.load(tmp_path / f"y{r}.npy")
        assert y.shape == ref.shape
        assert np.abs(y - ref).max() < 1e-6  # same ATen ops on a sub-batch
    sizes = [int(np.load(tmp_path / f"n{r}.npy").sum()) for r in range(world)]
    assert sum(sizes) == batch and max(sizes) - min(sizes) <= 1


def _stream_worker(rank, world, rendezvous, n_frames, out_dir):
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"file://{rendezvous}", rank=rank, world_size=world)
    from oracle import oracle, torch_port
    from voice_activity_detection_amd.distributed import sharded_rows
    from voice_activity_detection_amd.seeded import seeded_features, seeded_state_dict

    T, hop = 24, 12
    state_np = seeded_state_dict(1234)
    state = {k: torch.from_numpy(v) for k, v in state_np.items()}
    feat = seeded_features(5, (n_frames, 80))
    W = oracle.lib().savad_oracle_stream_window_count(n_frames, T, hop)
    spans = []

    def windows_logp(lo, hi):  # the CPU stand-in of StreamingPredictor.predict_device's closure: same contract
        spans.append((lo, hi))
        win = np.zeros((hi - lo, T, 80), np.float32)
        for w in range(lo, hi):
            seg = feat[hop * w: hop * w + T]
            win[w - lo, : len(seg)] = seg
        return torch_port.forward(state, torch.from_numpy(win))

    logp = sharded_rows(W, windows_logp, (T, 2), torch.float32, torch.device("cpu"))
    np.save(os.path.join(out_dir, f"logp{rank}.npy"), logp.numpy())
    np.save(os.path.join(out_dir, f"span{rank}.npy"), np.array(spans))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [24 + 12 * 4, 24 + 12 * 3 + 5, 10])
def test_two_rank_streaming_shard_and_merge(tmp_path, n_frames):
    """The sharded branch of StreamingPredictor (configs[4]): windows split contiguously over 2 ranks, ONE all_gather of the
    log-probs (uneven split and the single-window case included), then the overlap merge -- against the unsharded oracle."""
    import ctypes

    from oracle import oracle
    from voice_activity_detection_amd.seeded import seeded_features, seeded_state_dict

    world, T, hop = 2, 24, 12
    mp.spawn(_stream_worker, args=(world, str(tmp_path / "rendezvous"), n_frames, str(tmp_path)), nprocs=world, join=True)
    ref_probs, ref_logp = oracle.predict_streaming(seeded_state_dict(1234), seeded_features(5, (n_frames, 80)), T=T, hop=hop)
    W = ref_logp.shape[0]
    covered = []
    for r in range(world):
        logp = np.load(tmp_path / f"logp{r}.npy")
        assert logp.shape == ref_logp.shape and np.abs(logp - ref_logp).max() < 2e-5
        probs = np.empty((n_frames,), np.float32)
        fp = ctypes.POINTER(ctypes.c_float)
        oracle.lib().savad_oracle_overlap_merge(np.ascontiguousarray(logp).ctypes.data_as(fp), W, n_frames, T, hop, probs.ctypes.data_as(fp))
        assert np.abs(probs - ref_probs).max() < 2e-5
        covered += [tuple(s) for s in np.load(tmp_path / f"span{r}.npy").reshape(-1, 2)]
    covered.sort()
    assert covered[0][0] == 0 and covered[-1][1] == W and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))


def _audio_worker(rank, world, rendezvous, n_samples, out_dir):
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"file://{rendezvous}", rank=rank, world_size=world)
    from oracle import logmel, torch_port
    from voice_activity_detection_amd import StreamingPredictor
    from voice_activity_detection_amd.distributed import all_gather_rows
    from voice_activity_detection_amd.seeded import seeded_state_dict

    T, hop = 24, 12
    state = {k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()}
    audio = (np.random.default_rng(4).standard_normal(n_samples) * 0.1).astype(np.float32)
    W, lo, hi, f0, f1, first, count = StreamingPredictor.audio_shard_plan(n_samples, T, hop, dist.get_rank(), world)
    # CPU stand-ins with the contracts of log_mel_span (frames [f0, f1) of the recording's log-mel: the device kernel computes them
    # from the slice alone, GPU test) and of _windows_logp (forwards on LOCAL frame indices)
    if hi > lo:
        feat = logmel.log_mel(audio)[f0:f1]
        assert first <= max(0, 160 * f0 - 208) and first + count >= min(n_samples, 160 * (f1 - 1) + 208)
        win = np.zeros((hi - lo, T, 80), np.float32)
        for w in range(lo, hi):
            seg = feat[hop * (w - lo): hop * (w - lo) + T]
            win[w - lo, : len(seg)] = seg
        local = torch_port.forward(state, torch.from_numpy(win))
    else:
        local = torch.zeros((0, T, 2))
    logp = all_gather_rows(local, W)
    np.save(os.path.join(out_dir, f"alogp{rank}.npy"), logp.numpy())
    np.save(os.path.join(out_dir, f"aplan{rank}.npy"), np.array([W, lo, hi, f0, f1, first, count]))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_samples", [160 * (24 + 12 * 6), 160 * (24 + 12 * 5) + 1234, 160 * 30])
def test_two_rank_audio_level_sharding(tmp_path, n_samples):
    """configs[4] sharded from the AUDIO (StreamingPredictor.predict_audio_device): each of two gloo ranks takes the plan of
    audio_shard_plan -- its windows, their frames, the samples those frames read -- computes features for ITS frames only, forwards
    its windows on local frame indices, and one all_gather gives both ranks what the unsharded oracle computes.  (The device halves
    of the contract -- savad_logmel_span == rows of the whole log-mel, windows in place == window copies -- are GPU tests.)"""
    from oracle import logmel, oracle
    from voice_activity_detection_amd.seeded import seeded_state_dict

    world, T, hop = 2, 24, 12
    mp.spawn(_audio_worker, args=(world, str(tmp_path / "rendezvous"), n_samples, str(tmp_path)), nprocs=world, join=True)
    audio = (np.random.default_rng(4).standard_normal(n_samples) * 0.1).astype(np.float32)
    feat = logmel.log_mel(audio)
    _, ref_logp = oracle.predict_streaming(seeded_state_dict(1234), feat, T=T, hop=hop)
    plans = [np.load(tmp_path / f"aplan{r}.npy") for r in range(world)]
    assert plans[0][1] == 0 and plans[0][2] == plans[1][1] and plans[1][2] == plans[0][0] == ref_logp.shape[0]
    for r in range(world):
        logp = np.load(tmp_path / f"alogp{r}.npy")
        assert logp.shape == ref_logp.shape and np.abs(logp - ref_logp).max() < 2e-5
        W, lo, hi, f0, f1, first, count = plans[r]
        if hi > lo and world > 1 and len(feat) > 2 * T:
            assert count < n_samples   # a rank's share of the samples, not the recording


# ---- the multi-batch form: ShardedPipeline / forward_sharded_many ------------------------------------------------------------
def _pipeline_worker(rank, world, rendezvous, batch, gather, depth, out_dir):
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"file://{rendezvous}", rank=rank, world_size=world)
    from oracle import torch_port
    from voice_activity_detection_amd.distributed import ShardedPipeline, collective_counts, forward_sharded_many
    from voice_activity_detection_amd.seeded import seeded_features, seeded_state_dict

    state = {k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()}

    def fwd(x, out):   # the CPU stand-in of model(features=x, out=out)
        out.copy_(torch_port.forward(state, x))
        return out

    sp = ShardedPipeline(forward=fwd, slots=3, depth=depth, gather=gather)
    batches = [torch.from_numpy(seeded_features(20 + i, (batch, 9, 80))) for i in range(5)]   # 5 batches through 3 slots: two joins
    before = collective_counts()["all_gather"]
    ys = forward_sharded_many(sp, batches)
    n_gathers = collective_counts()["all_gather"] - before
    np.save(os.path.join(out_dir, f"y{rank}.npy"), torch.stack(ys).numpy())
    np.save(os.path.join(out_dir, f"g{rank}.npy"), np.array([n_gathers]))
    dist.destroy_process_group()


@pytest.mark.parametrize("batch,gather,depth", [(6, "step", 1), (6, "final", 1), (5, "step", 3), (5, "final", 3), (1, "step", 2)])
def test_two_rank_sharded_pipeline_matches_unsharded(tmp_path, batch, gather, depth):
    """K batches per join through ShardedPipeline on two gloo ranks (even, ragged and single-row global batches; gathers lagging
    behind the forwards as with several forwards in flight): every rank ends up with the unsharded results, after one all_gather
    per forward ('step') or one per join ('final')."""
    from oracle import torch_port
    from voice_activity_detection_amd.seeded import seeded_features, seeded_state_dict

    world = 2
    mp.spawn(_pipeline_worker, args=(world, str(tmp_path / "rendezvous"), batch, gather, depth, str(tmp_path)), nprocs=world, join=True)
    state = {k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()}
    ref = np.stack([torch_port.forward(state, torch.from_numpy(seeded_features(20 + i, (batch, 9, 80)))).numpy() for i in range(5)])
    for r in range(world):
        y = np.load(tmp_path / f"y{r}.npy")
        assert y.shape == ref.shape and np.abs(y - ref).max() < 1e-6
        assert int(np.load(tmp_path / f"g{r}.npy")[0]) == (5 if gather == "step" else 2)


def test_bench_runs_with_two_gloo_ranks(tmp_path):
    """bench.py's N > 1 control flow end to end on the CPU: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2
    --backend gloo --stub-forward` -- in-flight tuning agreed across the ranks, K-step blocks, both gather modes, the config3 leg,
    ONE JSON line from rank 0, the same number of collectives on both ranks, a clean exit.  (The stand-in forward replaces the
    library; everything between it and the JSON line is the code the driver's 8-GPU run executes.)"""
    import json
    import subprocess
    import sys
    from pathlib import Path

    repo = Path(__file__).resolve().parent.parent
    for attempt in range(3):   # the driver's own command line (a fixed --master-port): a port picked here can be gone by the time it is bound
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), str(repo / "bench.py"), "--gpus", "2", "--backend", "gloo", "--stub-forward",
               "--steps", "3", "--warmup", "1", "--min-seconds", "0.01", "--batch", "2", "--frames", "40", "--config3-shape", "3,24"]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
        if res.returncode == 0 or "EADDRINUSE" not in res.stderr and "address already in use" not in res.stderr.lower():
            break
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["stub_forward"] is True and d["finite"] is True
    assert d["steps"] == 3 and d["config"]["global_batch"] == 4
    assert "gather_step_ms" in d and "gather_final_ms" in d and "value_one_forward" in d
    assert set(d["in_flight_tuning_ms"]) == {"1", "2", "3"}          # the tuning ran (and was agreed: both ranks went on)
    assert d["config3"]["finite"] is True and d["config3"]["global_batch"] == 6
    c0, c1 = d["collective_counts"]
    assert c0 == c1 and c0["all_gather"] > 0 and c0["barrier"] > 0


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 ...` invoked PLAINLY (no launcher, no WORLD_SIZE): the file re-executes itself under
    torch.distributed.run with one rank per GPU -- the form the driver used for its 1-GPU line; with N > 1 it used to exit with a
    usage message.  The headline of an N > 1 run is the single gather that closes the block (north_star), and the line says so."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    repo = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, str(repo / "bench.py"), "--gpus", "2", "--backend", "gloo", "--stub-forward", "--steps", "3", "--warmup", "1",
           "--min-seconds", "0.01", "--batch", "2", "--frames", "40", "--config3-shape", "3,24"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(tmp_path), env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["finite"] is True and d["config"]["global_batch"] == 4
    assert "per block" in d["config"]["parallelism"] and d["gather_final_ms"] == d["ms_per_step"]   # value = the single-gather mode
    assert "gather_step_ms" in d and "`value` is quoted on gather_final" in d["note"]


def test_sharded_pipeline_without_a_process_group():
    """no process group: the plain pipeline (world 1, no collective), with its argument checks"""
    from voice_activity_detection_amd.distributed import ShardedPipeline, collective_counts, forward_sharded_many

    calls = []

    def fwd(x, out):
        calls.append(tuple(x.shape))
        out.copy_(torch.log_softmax(x[..., :2], dim=-1))
        return out

    with pytest.raises(ValueError):
        ShardedPipeline(forward=fwd, gather="sometimes")
    with pytest.raises(ValueError):
        ShardedPipeline()
    sp = ShardedPipeline(forward=fwd, slots=2, depth=3, gather="step")
    assert (sp.world, sp.rank, sp.distributed, sp.depth, sp.in_flight) == (1, 0, False, 3, 3)
    before = collective_counts()
    x = torch.randn(3, 5, 80)
    sp.submit(x)
    sp.submit(x + 1)
    assert sp.pending == 2
    with pytest.raises(RuntimeError):
        sp.submit(x)                      # more batches than slots between two joins
    with pytest.raises(RuntimeError):
        sp.set_in_flight(1)               # ... and no retuning in between either
    outs = sp.join()
    assert len(outs) == 2 and tuple(outs[0].shape) == (1, 3, 5, 2) and sp.pending == 0 and sp.join() == []
    assert torch.equal(outs[1][0], torch.log_softmax((x + 1)[..., :2], dim=-1))
    sp.set_in_flight(1)
    sp.set_gather("final")
    ys = forward_sharded_many(sp, [x, x + 1, x + 2])      # three batches through two slots
    assert len(ys) == 3 and torch.equal(ys[2], torch.log_softmax((x + 2)[..., :2], dim=-1))
    sp.submit(torch.randn(4, 7, 80))       # another shape after a join: buffers follow
    assert tuple(sp.join()[0].shape) == (1, 4, 7, 2)
    assert collective_counts() == before


def _bench_line(res):
    import json

    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:] + res.stderr[-2000:]
    return json.loads(lines[0])


def test_bench_self_checks_fail_the_run(tmp_path):
    """Every leg's self-check (`finite`, `*equals*`, `within_*`) is folded into the top-level `finite`, a compact `summary` is the LAST key
    of the line (what a truncated log still shows), and a check that reads false ends the run with a non-zero exit code: planted here
    (round 5's configs[2] leg printed finite = false and nothing noticed)."""
    import subprocess
    import sys
    from pathlib import Path

    repo = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, str(repo / "bench.py"), "--gpus", "1", "--backend", "gloo", "--stub-forward", "--steps", "3", "--warmup", "1",
           "--min-seconds", "0.01", "--batch", "2", "--frames", "40"]
    ok = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=str(tmp_path), env=env)
    assert ok.returncode == 0, ok.stderr[-2000:]
    d = _bench_line(ok)
    assert list(d)[-1] == "summary" and d["finite"] is True
    assert d["summary"]["all_flags_true"] is True and d["summary"]["false_flags"] == [] and d["summary"]["flags_checked"] >= 1
    assert list(d)[:4] == ["metric", "value", "value_one_forward", "ms_one_forward"] and "section 8d" in d["config"]["workload"]
    assert {"headline", "headline_one_forward"} <= set(d["summary"])
    bad = subprocess.run(cmd + ["--plant-false-flag"], capture_output=True, text=True, timeout=300, cwd=str(tmp_path), env=env)
    assert bad.returncode == 3, (bad.returncode, bad.stderr[-2000:])
    d = _bench_line(bad)
    assert d["finite"] is False and d["summary"]["all_flags_true"] is False and d["summary"]["false_flags"] == ["secondary.planted.finite"]


def test_bench_dry_run_at_the_real_rank_count(tmp_path):
    """`python bench.py --gpus 8 --backend gloo --stub-forward`, plain (the file launches its own eight ranks): the control flow of the
    driver's 8-GPU scaling run at the rank count it will use -- tuning agreed by eight ranks, both gather modes, the config3 leg with a
    batch that does not divide by eight, ONE line, the same collective counts on all eight ranks, a clean exit."""
    import subprocess
    import sys
    from pathlib import Path

    repo = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, str(repo / "bench.py"), "--gpus", "8", "--backend", "gloo", "--stub-forward", "--steps", "2", "--warmup", "1",
           "--min-seconds", "0.01", "--batch", "2", "--frames", "24", "--config3-shape", "3,16"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(tmp_path), env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _bench_line(res)
    assert d["n_gpus"] == 8 and d["finite"] is True and d["config"]["global_batch"] == 16 and d["config3"]["global_batch"] == 24
    counts = d["collective_counts"]
    assert len(counts) == 8 and all(c == counts[0] for c in counts) and counts[0]["all_gather"] > 0
    assert list(d)[-1] == "summary" and d["summary"]["all_flags_true"] is True


def _hour_worker(rank, world, rendezvous, out_dir):
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"file://{rendezvous}", rank=rank, world_size=world)
    from voice_activity_detection_amd import distributed as vdist
    from voice_activity_detection_amd.distributed import shard_bounds, sharded_rows

    n_samples, T, hop = 16000 * 3600, 800, 400
    N = 1 + n_samples // 160
    W = 1 if N <= T else (N - T + hop - 1) // hop + 1          # savad_stream_window_count (csrc/savad.hip), restated: no library on the CPU
    lo, hi = shard_bounds(W, rank, world)
    f0, f1 = hop * lo, min(N, hop * (hi - 1) + T)              # StreamingPredictor.audio_shard_plan's frame span
    spans = []

    def windows_logp(a, b):   # stand-in for the rank's forwards: window w -> its index in channel 0, the last frame it really covers in channel 1
        spans.append((a, b))
        w = torch.arange(a, b, dtype=torch.float32)
        out = torch.empty((b - a, T, 2), dtype=torch.float32)
        out[..., 0] = w[:, None]
        out[..., 1] = torch.minimum(hop * w + T, torch.tensor(float(N)))[:, None]
        return out

    logp = sharded_rows(W, windows_logp, (T, 2), torch.float32, torch.device("cpu"))
    ok = logp.shape == (W, T, 2) and bool((logp[:, 0, 0] == torch.arange(W, dtype=torch.float32)).all()) and float(logp[-1, 0, 1]) == float(N)
    np.save(os.path.join(out_dir, f"hour{rank}.npy"), np.array([W, lo, hi, f0, f1, int(ok), spans[0][0], spans[0][1], vdist.collective_counts()["all_gather"]]))
    dist.destroy_process_group()


def test_eight_rank_hour_plan_and_ragged_gather(tmp_path):
    """configs[4] at the rank count of a node: an hour of audio = 360 001 frames = 900 windows of 800 frames every 400 -> 112 or 113
    windows per rank (contiguous), the last rank's last window zero-padded past the recording, ONE ragged all_gather of [n_r, 800, 2]
    per rank (padded to 113 rows) that every rank reassembles to the same [900, 800, 2]; equal collective counts on all eight."""
    world = 8
    mp.spawn(_hour_worker, args=(world, str(tmp_path / "rendezvous"), str(tmp_path)), nprocs=world, join=True)
    rows = [np.load(tmp_path / f"hour{r}.npy") for r in range(world)]
    assert all(int(r[0]) == 900 for r in rows)
    sizes = [int(r[2] - r[1]) for r in rows]
    assert sum(sizes) == 900 and set(sizes) == {112, 113} and all(int(rows[i][2]) == int(rows[i + 1][1]) for i in range(world - 1))
    assert all(int(r[5]) == 1 for r in rows)                                      # every rank holds the whole, correctly ordered result
    assert all((int(r[6]), int(r[7])) == (int(r[1]), int(r[2])) for r in rows)    # each computed exactly its own span
    assert int(rows[-1][4]) == 360001 and 400 * 899 + 800 > 360001                  # the tail window reaches past the recording: zero-padded
    assert len({int(r[8]) for r in rows}) == 1 and int(rows[0][8]) == 1           # one collective each, the same on all ranks
