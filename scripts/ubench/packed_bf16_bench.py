#!/usr/bin/env python3
"""T <= 32 forwards, ms per forward: fp32 single launch against bf16 single launch (4- and 8-wave workgroups) and the bf16
per-layer launches, and the reference-mode hour (359 963 windows of 7 frames through the predictor).
usage: python scripts/ubench/packed_bf16_bench.py"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from voice_activity_detection_amd import SelfAttentiveVAD, VADFromScratchPredictor, seeded_state_dict  # noqa: E402


def timed(fn, reps, blocks=7):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(blocks):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / reps)
    return round(float(np.median(out)), 4), round(float(min(out)), 4)


def main():
    m = SelfAttentiveVAD(80, 3, 128, 0.5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
    m = m.cuda().eval()
    res = {}
    shapes = ((1000, 7), (250, 7), (4000, 7), (16384, 7), (65536, 7), (400, 20), (1000, 32))
    if "--quick" in sys.argv:
        shapes = ((1000, 7), (2000, 7), (4000, 7), (8000, 7), (65536, 7))
    for B, T in shapes:
        x = torch.randn(B, T, 80, device="cuda")
        out = torch.empty(B, T, 2, device="cuda")
        reps = 50 if B <= 4000 else 5
        row = {}
        for name, prec, mode in (("fp32", "fp32", 0), ("bf16_auto", "bf16", 0), ("bf16_nw8", "bf16", 5), ("bf16_nw4_ring4", "bf16", 6), ("bf16_nw4_ring2", "bf16", 7), ("bf16_nsplit", "bf16", 8), ("bf16_per_layer", "bf16", 1)):
            m.precision = prec
            m.row_mode = mode
            with torch.no_grad():
                row[name] = timed(lambda: m(features=x, out=out), reps)
        m.precision, m.row_mode = "fp32", 0
        res[f"[{B},{T},80]"] = row
    if "--quick" in sys.argv:
        print(json.dumps(res))
        return
    # reference mode: an hour of audio = 360 001 feature frames -> 359 963 windows of 7 frames
    feat = torch.randn(360001, 80, device="cuda")
    for prec in ("fp32", "bf16"):
        m.precision = prec
        pred = VADFromScratchPredictor(m, "cuda")
        with torch.no_grad():
            res[f"1h_reference_mode_{prec}"] = timed(lambda: pred.predict_probabilities_device(feat), 2, blocks=5)
    m.precision = "fp32"
    print(json.dumps(res))


if __name__ == "__main__":
    main()
