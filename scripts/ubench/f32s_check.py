"""Dev check of precision "fp32s" (csrc/savad_kernels_f32s.h): errors against the reference goldens next to the exact-fp32 path's,
and event timings of one forward at the headline shapes.  python scripts/ubench/f32s_check.py [--time-only]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_features, seeded_state_dict  # noqa: E402

golden = np.load(REPO / "tests" / "golden" / "golden.npz")


def make(state, F=80, L=3):
    m = SelfAttentiveVAD(F, L, 128, 0.5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    return m.to("cuda").eval()


def run(m, x, prec):
    m.precision = prec
    with torch.no_grad():
        y = m(features=torch.from_numpy(x).to("cuda"))
    torch.cuda.synchronize()
    return y.cpu().numpy()


m = make(seeded_state_dict(1234))
if "--time-only" not in sys.argv:
    cases = [("g1_out", 101, (4, 7, 80), "logmel"), ("g2_out", 102, (2, 800, 80), "logmel"), ("g2n_out", 103, (3, 200, 80), "normal"),
             ("g6_out", 600, (2, 40, 80), "logmel"), ("g4_B1T7", 77, (1, 7, 80), "logmel")]
    cases += [(f"g4_T{T}", 400 + T, (3, T, 80), "logmel") for T in (1, 2, 5, 10, 11, 16, 17, 31, 32, 33, 63, 64, 65, 100, 799, 801)]
    worst = 0.0
    for tag, seed, shape, kind in cases:
        x = seeded_features(seed, shape, kind)
        e32 = np.abs(run(m, x, "fp32") - golden[tag]).max()
        y = run(m, x, "fp32s")
        es = np.abs(y - golden[tag]).max() if np.isfinite(y).all() else float("nan")
        worst = max(worst, es) if es == es else float("inf")
        print(f"{tag:12s} {str(shape):16s} fp32 {e32:.2e}   fp32s {es:.2e}", flush=True)
    mp = make(seeded_state_dict(4321, gain=4.0))
    for tag, seed, shape in (("g7_out", 700, (2, 96, 80)), ("g7_T800", 701, (1, 800, 80))):
        x = seeded_features(seed, shape)
        print(f"{tag:12s} peaked           fp32 {np.abs(run(mp, x, 'fp32') - golden[tag]).max():.2e}   fp32s {np.abs(run(mp, x, 'fp32s') - golden[tag]).max():.2e}")
    x = seeded_features(0, (32, 800, 80))
    y = run(m, x, "fp32s")
    print("g3 head/tail fp32s", np.abs(y[:2] - golden["g3_head"]).max(), np.abs(y[-2:] - golden["g3_tail"]).max(),
          "seqsum", np.abs(y.astype(np.float64).sum(axis=(1, 2)) - golden["g3_seqsum"]).max())
    print("worst fp32s error on the goldens:", worst)

for shape in ((32, 800, 80), (1000, 7, 80), (256, 800, 80), (8, 800, 80)):
    x = torch.from_numpy(seeded_features(5, shape)).to("cuda")
    for prec in ("fp32", "fp32s"):
        m.precision = prec
        with torch.no_grad():
            for _ in range(5):
                m(features=x)
            torch.cuda.synchronize()
            n = 30
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                m(features=x)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            m.set_profiling(8, 2)
            for _ in range(10):
                m(features=x)
            torch.cuda.synchronize()
            kt = m.kernel_times()
            m.set_profiling(0)
        fl = 2429440.0 * shape[0] * shape[1] if shape[1] == 800 else 1211392.0 * shape[0] * shape[1]
        print(f"{str(shape):16s} {prec:6s} {ms:.4f} ms/forward  {fl / ms / 1e9:.1f} TFLOP/s fp32-equivalent   kernels: " +
              ", ".join(f"{k} {v * 1e3:.1f}us" for k, v in kt), flush=True)
