"""Forwards while another stream evicts the L2s / the Infinity Cache, bit for bit against the same forwards on a quiet GPU.

The race of DESIGN.md section 4i (registers reused, or copied, while a hand-issued weight load was still in flight) needed the load
to MISS the L2: a steady loop of forwards keeps the 2.4 MB of weights cached and never provokes it.  Here a side stream copies 1 GiB
back and forth while three forwards are in flight on their own streams (and while the module runs alone), for one shape per kernel
family that issues loads by hand -- fp32 N-split with key splits, the fp32 and the bf16 single-launch T <= 32 forwards, M-split /
fused fp32, the bf16 ring kernels with their hand-issued LDS reads, the log-mel kernel's DMA stage.  The static check
(tests/test_async_load_hazards.py) proves the absence of that bug class in the compiled code; this is its dynamic counterpart.

Round 5: seen green on hardware (it XPASSed 3/3 in the driver's round-4 tier), so the xfail marker is gone; the all-families sweep of
scripts/ubench/l2_pressure_stress.py is folded in; and a NEGATIVE test runs the same loop on a deliberately broken build
(tests/fault/, built by __graft_entry__.build) -- which the static checker must flag in any case (CPU suite).  Round 6: the planted
faults are deterministic (the load cannot have landed when its registers / its LDS block are read), the negative test is strict, and
it covers one register-level and one LDS-DMA publication fault; the fp32s kernels joined the pressure cases."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent

CASES = [("fp32s", (32, 800, 80)), ("fp32s", (500, 7, 80)), ("fp32s", (3000, 7, 80)), ("fp32s", (8, 200, 80)), ("fp32", (8, 200, 80)), ("fp32", (2, 800, 80)), ("fp32", (500, 7, 80)), ("fp32", (64, 20, 80)), ("fp32", (32, 800, 80)),
         ("fp32", (512, 50, 80)), ("bf16", (40, 264, 80)), ("bf16", (500, 7, 80)), ("bf16", (2000, 7, 80)), ("bf16", (9000, 7, 80)), ("bf16", (64, 800, 80))]
# (the three bf16 T = 7 shapes: one block per workgroup with weights hand-streamed from L2; four blocks + four mover waves; the 2-slot ring)


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


def pressure_rounds(torch, model, precision, shape, rounds, stop_at_first=False, side_streams=1):
    """-> list of (round, batch, rows that differ, max diff, first 32-row tiles) against the quiet run"""
    from voice_activity_detection_amd import PipelinedVAD
    from voice_activity_detection_amd.seeded import seeded_features

    model.precision = precision
    try:
        xs = [torch.from_numpy(seeded_features(7 * i + shape[1], shape)).cuda() for i in range(3)]
        with torch.no_grad():
            want = [model(features=x).clone() for x in xs]
        torch.cuda.synchronize()
        a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
        b = torch.empty_like(a)
        sides = [torch.cuda.Stream() for _ in range(side_streams)]
        n = a.numel() // side_streams
        pipe = PipelinedVAD(model, 3)
        differing = []
        for rnd in range(rounds):
            for k, side in enumerate(sides):  # every side stream thrashes its own slice of the two buffers
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(4):
                        b[k * n:(k + 1) * n].copy_(a[k * n:(k + 1) * n], non_blocking=True)
                        a[k * n:(k + 1) * n].copy_(b[k * n:(k + 1) * n], non_blocking=True)
            with torch.no_grad():
                if rnd % 2 == 0:
                    outs = [o.clone() for o in pipe.forward_many([x * 1.0 for x in xs])]
                else:
                    outs = [model(features=x).clone() for x in xs]
            for i, (o, w) in enumerate(zip(outs, want)):
                if not torch.equal(o, w):
                    rows = torch.nonzero((o - w).abs().amax(dim=2).reshape(-1)).reshape(-1)
                    differing.append((rnd, i, int(rows.numel()), float((o - w).abs().max()), sorted(set((rows // 32).tolist()))[:8]))
            for side in sides:
                torch.cuda.current_stream().wait_stream(side)
            if differing and stop_at_first:
                break
        torch.cuda.synchronize()
        return differing
    finally:
        model.precision = "fp32"


@pytest.mark.parametrize("precision,shape", CASES)
def test_forwards_under_cache_eviction_keep_their_bits(torch_cuda, model, precision, shape):
    differing = pressure_rounds(torch_cuda, model, precision, shape, rounds=8)
    assert not differing, f"(round, batch, rows, max diff, first 32-row tiles) that differ from the quiet run: {differing[:6]}"


def test_logmel_under_cache_eviction_keeps_its_bits(torch_cuda):
    """the factored log-mel kernel stages its samples by global -> LDS DMA one tile ahead: the same loop around it"""
    import numpy as np

    from voice_activity_detection_amd.features import log_mel

    torch = torch_cuda
    y = torch.from_numpy(np.random.default_rng(3).standard_normal(16000 * 600).astype(np.float32) * 0.1).cuda()
    want = log_mel(y).clone()
    a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    side = torch.cuda.Stream()
    bad = 0
    for _ in range(8):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                b.copy_(a, non_blocking=True)
                a.copy_(b, non_blocking=True)
        for _ in range(6):
            bad += not torch.equal(log_mel(y), want)
        torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert bad == 0


_CHILD = """
import json, sys
import numpy as np
sys.path.insert(0, {repo!r})
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features, _lib
assert str(_lib.LIB_PATH).endswith("libsavad_fault17.so")
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({{k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()}})
m = m.cuda().eval()
want = np.load({want!r})     # the PRODUCT library's bits for the same inputs
res = {{}}
for tag, prec, shape in (("fp32_packed", "fp32", (500, 7, 80)), ("bf16_row", "bf16", (40, 264, 80)), ("fp32_long", "fp32", (2, 800, 80))):
    m.precision, m.row_mode = prec, (4 if tag == "fp32_packed" else 0)   # (4: the single-launch forward, whatever the automatic choice at this size)
    x = torch.from_numpy(seeded_features(5, shape)).cuda()
    w = torch.from_numpy(want[tag]).cuda()
    bad = 0
    for rnd in range(5):
        with torch.no_grad():
            bad += not torch.equal(m(features=x), w)
    res[tag] = bad
print("RESULT " + json.dumps(res))
"""


def test_a_deliberately_broken_build_is_caught(torch_cuda, model, tmp_path):
    """NEGATIVE test, strict since round 6: libsavad built with SAVAD_FAULT_INJECT=17 run in a process of its own against the PRODUCT
    library's bits for the same inputs.  Bit 1: the single-launch fp32 forward re-targets the query block's registers with a new request
    while the query GEMM still reads them (registers touched while a load to them is in flight, round 4's bug class).
    Bit 16: the bf16 row chain hands ring block 2 over at its barrier without the counted wait, the DMA issued right in front of it (an
    LDS-DMA publication fault: every wave reads block 0's bytes).  Both are planted so that no cache state, clock or box decides (the load lands INSIDE the
    64-MFMA GEMM that reads its registers; the DMA is issued a few cycles before its block is read): every forward through a faulty kernel must differ, every forward through an untouched kernel (the fused
    fp32 long-sequence launches) must not.  The static checker flags the same build (tests/test_async_load_hazards.py)."""
    import numpy as np

    from voice_activity_detection_amd import build
    from voice_activity_detection_amd.seeded import seeded_features

    torch = torch_cuda
    lib = build.build_variant(build.FAULT_LIB, build.FAULT_DEFINES)   # (rebuilt only when missing or older than the sources)
    want = {}
    try:
        for tag, prec, shape in (("fp32_packed", "fp32", (500, 7, 80)), ("bf16_row", "bf16", (40, 264, 80)), ("fp32_long", "fp32", (2, 800, 80))):
            model.precision, model.row_mode = prec, (4 if tag == "fp32_packed" else 0)
            with torch.no_grad():
                want[tag] = model(features=torch.from_numpy(seeded_features(5, shape)).cuda()).cpu().numpy()
    finally:
        model.precision, model.row_mode = "fp32", 0
    np.savez(tmp_path / "want.npz", **want)
    env = dict(os.environ, SAVAD_LIB=str(lib))
    out = subprocess.run([sys.executable, "-c", _CHILD.format(repo=str(REPO), want=str(tmp_path / "want.npz"))], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    differing = json.loads(line[len("RESULT "):])
    assert differing == {"fp32_packed": 5, "bf16_row": 5, "fp32_long": 0}, differing
