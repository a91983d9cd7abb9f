"""Forwards while another stream evicts the L2s / the Infinity Cache, bit for bit against the same forwards on a quiet GPU.

The race of DESIGN.md section 4i (registers reused, or copied, while a hand-issued weight load was still in flight) needed the load
to MISS the L2: a steady loop of forwards keeps the 2.4 MB of weights cached and never provokes it.  Here a side stream copies 1 GiB
back and forth while three forwards are in flight on their own streams (and while the module runs alone), for one shape per kernel
family that issues loads by hand: fp32 N-split with key splits, the single-launch T <= 32 forward, bf16.  The static check
(tests/test_async_load_hazards.py) proves the absence of that bug class in the compiled code; this is its dynamic counterpart.

Written after round 4's GPU budget was spent: expected to pass (every ingredient is exercised by the tests that did run), but marked
xfail(strict=False) until the driver's round-end tier has shown it green on hardware once -- an XPASS there is the green; then the
marker goes (scripts/ubench/l2_pressure_stress.py is the longer, all-shapes form)."""
import pytest

pytestmark = pytest.mark.gpu

CASES = [("fp32", (8, 200, 80)), ("fp32", (500, 7, 80)), ("bf16", (40, 264, 80))]


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


@pytest.mark.xfail(strict=False, reason="first run on hardware is the driver's round-end tier (round 4 had no GPU minutes left); expected XPASS")
@pytest.mark.parametrize("precision,shape", CASES)
def test_forwards_under_cache_eviction_keep_their_bits(torch_cuda, model, precision, shape):
    from voice_activity_detection_amd import PipelinedVAD
    from voice_activity_detection_amd.seeded import seeded_features

    torch = torch_cuda
    model.precision = precision
    try:
        xs = [torch.from_numpy(seeded_features(7 * i + shape[1], shape)).cuda() for i in range(3)]
        with torch.no_grad():
            want = [model(features=x).clone() for x in xs]
        torch.cuda.synchronize()
        a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
        b = torch.empty_like(a)
        side = torch.cuda.Stream()
        pipe = PipelinedVAD(model, 3)
        differing = []
        for rnd in range(12):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(4):
                    b.copy_(a, non_blocking=True)
                    a.copy_(b, non_blocking=True)
            with torch.no_grad():
                if rnd % 2 == 0:
                    outs = [o.clone() for o in pipe.forward_many([x * 1.0 for x in xs])]
                else:
                    outs = [model(features=x).clone() for x in xs]
            for i, (o, w) in enumerate(zip(outs, want)):
                if not torch.equal(o, w):
                    rows = torch.nonzero((o - w).abs().amax(dim=2).reshape(-1)).reshape(-1)
                    differing.append((rnd, i, int(rows.numel()), float((o - w).abs().max()), sorted(set((rows // 32).tolist()))[:8]))
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert not differing, f"(round, batch, rows, max diff, first 32-row tiles) that differ from the quiet run: {differing[:6]}"
    finally:
        model.precision = "fp32"
