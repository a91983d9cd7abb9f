"""The multi-GPU path on the REAL backend: `torch.distributed` with backend "nccl" (= RCCL on ROCm) initialised in
this process with world_size 1 -- the largest world a 1-GPU box offers -- and the product's own collective code
(voice_activity_detection_amd.distributed: forward_sharded, all_gather_rows, sharded_rows; the sharded branch of
StreamingPredictor.predict_device) pushed through it, against the unsharded results.  The world-size-2 logic
(uneven splits, empty shards) is covered on CPU by tests/test_dist_gloo.py; the 8-GPU run itself belongs to the
driver's scaling bench.  SURVEY.md section 8e: weights replicated, sequences split contiguously, ONE all_gather."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (torch.cuda.is_available() is False)")
    return torch


@pytest.fixture(scope="module")
def model(torch_cuda, state1234):
    from voice_activity_detection_amd import SelfAttentiveVAD

    m = SelfAttentiveVAD(80, 3, 128, 0.5)
    m.load_state_dict({k: torch_cuda.from_numpy(v) for k, v in state1234.items()}, strict=True)
    return m.to("cuda").eval()


@pytest.fixture()
def rccl(torch_cuda, tmp_path):
    """A world-size-1 RCCL process group for the duration of ONE test (other tests must see no process group).  The rendezvous
    goes through a file store: a TCP store on a port picked by bind(0) / close / listen again lost the port to another socket in
    2 of 22 runs of this file on the GPU box (EADDRINUSE)."""
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"file://{tmp_path / 'rendezvous'}", rank=0, world_size=1,
                            device_id=torch_cuda.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    try:
        yield dist
    finally:
        torch_cuda.cuda.synchronize()
        dist.destroy_process_group()


def feats(seed, shape):
    from voice_activity_detection_amd.seeded import seeded_features

    return seeded_features(seed, shape)


def test_forward_sharded_through_rccl(torch_cuda, model, state1234, rccl):
    from oracle import oracle
    from voice_activity_detection_amd.distributed import forward_sharded

    torch = torch_cuda
    x = feats(21, (5, 96, 80))
    xd = torch.from_numpy(x).cuda()
    calls = []

    def fwd(t):
        calls.append(tuple(t.shape))
        with torch.no_grad():
            return model(features=t)

    y = forward_sharded(fwd, xd)
    torch.cuda.synchronize()
    assert calls == [(5, 96, 80)] and y.shape == (5, 96, 2) and y.is_cuda
    with torch.no_grad():
        direct = model(features=xd)
    assert torch.equal(y, direct)  # the gather moved the bits, nothing else
    assert np.abs(y.cpu().numpy() - oracle.forward(state1234, x)).max() < 3e-5


def test_all_gather_rows_and_barrier_through_rccl(torch_cuda, rccl):
    from voice_activity_detection_amd.distributed import all_gather_rows, sharded_rows

    torch = torch_cuda
    local = torch.arange(7 * 6, dtype=torch.float32, device="cuda").reshape(7, 3, 2)
    got = all_gather_rows(local, 7)
    rccl.barrier()
    assert torch.equal(got, local) and got.data_ptr() != local.data_ptr()
    nonc = torch.arange(14 * 6, dtype=torch.float32, device="cuda").reshape(14, 3, 2)[::2]  # a strided shard is padded/copied
    assert torch.equal(all_gather_rows(nonc, 7), nonc)
    got2 = sharded_rows(7, lambda lo, hi: local[lo:hi] * 2, (3, 2), torch.float32, local.device)
    assert torch.equal(got2, local * 2)
    t = torch.ones(4, device="cuda")
    rccl.all_reduce(t)  # the bench's max-over-ranks reduction uses the same group
    assert float(t.sum()) == 4.0


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_streaming_predictor_sharded_branch_through_rccl(torch_cuda, model, state1234, precision, tmp_path):
    """configs[4]: the long-form predictor's window sharding + single all_gather + overlap merge with a live RCCL
    group must reproduce the run without a process group bit for bit, and the oracle within tolerance."""
    import torch.distributed as dist

    from oracle import oracle
    from voice_activity_detection_amd.predictor import StreamingPredictor

    torch = torch_cuda
    feat = feats(33, (96 * 3 + 17, 80))
    model.precision = precision
    try:
        sp = StreamingPredictor(model, "cuda", T=96, hop=48, max_batch=4)
        assert not dist.is_initialized()
        plain = sp.predict_device(feat).clone()
        # (fixture used by hand so that the un-initialised run above comes first in the same test)
        dist.init_process_group("nccl", init_method=f"file://{tmp_path / 'rendezvous'}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        try:
            sharded = sp.predict_device(feat)
            torch.cuda.synchronize()
        finally:
            dist.destroy_process_group()
    finally:
        model.precision = "fp32"
    assert torch.equal(plain, sharded)
    ref, _ = oracle.predict_streaming(state1234, feat, T=96, hop=48)
    assert np.abs(sharded.cpu().numpy() - ref).max() < (1e-4 if precision == "fp32" else 2e-2)


def test_config3_global_batch_on_one_gpu(torch_cuda, model, state1234, rccl):
    """BASELINE configs[3] at its STATED size: the global [2048, 800, 80] bf16 batch on one MI355X, pushed through
    distributed.forward_sharded with a live RCCL group in the eight 256-sequence shards the eight ranks of a node would
    see (shard_bounds).  Every shard must equal the unsharded forward of the same rows bit for bit, the global batch
    as ONE forward must equal the concatenated shards bit for bit, and sampled sequences must match the oracle."""
    from oracle import oracle
    from voice_activity_detection_amd.distributed import forward_sharded, shard_bounds

    torch = torch_cuda
    Bg, T, world = 2048, 800, 8
    x = feats(2048, (Bg, T, 80))
    xd = torch.from_numpy(x).cuda().to(torch.bfloat16)
    model.precision = "bf16"
    try:
        def fwd(t):
            with torch.no_grad():
                return model(features=t)

        y = torch.empty((Bg, T, 2), dtype=torch.float32, device="cuda")
        for r in range(world):
            lo, hi = shard_bounds(Bg, r, world)
            assert hi - lo == 256
            got = forward_sharded(fwd, xd[lo:hi])          # rank r's call: its shard -> forward -> one all_gather
            assert got.shape == (256, T, 2) and torch.equal(got, fwd(xd[lo:hi]))
            y[lo:hi] = got
        whole = fwd(xd)                                     # 1.64 M rows, ~2.1 GB workspace: one MI355X holds the global batch
        torch.cuda.synchronize()
        assert torch.equal(whole, y)
        assert model.residual_saturations() == 0
    finally:
        model.precision = "fp32"
    yh = y.cpu().numpy()
    assert np.isfinite(yh).all() and np.abs(np.logaddexp(yh[..., 0], yh[..., 1])).max() < 1e-5
    pick = [0, 255, 256, 1000, 1791, 2047]                  # first / last sequence of shards 0, 1, 3, 6, 7
    xin = xd[pick].float().cpu().numpy()                    # the oracle sees the same bf16-rounded features
    assert np.abs(yh[pick] - oracle.forward(state1234, xin)).max() < 1.2e-2  # BF16_TOL of tests/test_gpu_parity.py


@pytest.mark.parametrize("gather", ["step", "final"])
@pytest.mark.parametrize("precision,shape", [("fp32", (8, 200, 80)), ("bf16", (40, 264, 80)), ("fp32", (500, 7, 80))])
def test_sharded_pipeline_through_rccl(torch_cuda, model, rccl, gather, precision, shape):
    """ShardedPipeline (the multi-batch form of forward_sharded: K rank-local forwards in flight on their own streams, the RCCL
    gather of their log-probs per forward -- lagging behind the newer forwards -- or once at join): the module's own bits for
    every batch, in order, for several joins, with inputs produced right before submit and outputs read right after join."""
    from voice_activity_detection_amd.distributed import ShardedPipeline, collective_counts

    torch = torch_cuda
    model.precision = precision
    try:
        sp = ShardedPipeline(model, slots=4, depth=3, gather=gather)
        assert sp.world == 1 and sp.distributed
        before = collective_counts()["all_gather"]
        n = 0
        for rnd in range(3):
            xs = [torch.from_numpy(feats(100 * rnd + i, shape)).cuda() for i in range(4 if rnd < 2 else 2)]
            for x in xs:
                sp.submit(x * 1.0)             # a temporary: the pipeline must keep it alive
            outs = [o.clone() for o in sp.join()]
            n += len(xs)
            with torch.no_grad():
                for i, (x, o) in enumerate(zip(xs, outs)):
                    assert tuple(o.shape) == (1, shape[0], shape[1], 2)
                    want = model(features=x)
                    if not torch.equal(o[0], want):   # say where: a race shows as whole 32-row tiles
                        bad = torch.nonzero((o[0] - want).abs().amax(dim=2).reshape(-1)).reshape(-1)
                        raise AssertionError(f"join {rnd} batch {i}: {bad.numel()} rows differ from the module's own forward, max "
                                             f"{float((o[0] - want).abs().max()):.3g}; 32-row tiles {sorted(set((bad // 32).tolist()))}")
        got = collective_counts()["all_gather"] - before
        assert got == (n if gather == "step" else 3)
    finally:
        model.precision = "fp32"
