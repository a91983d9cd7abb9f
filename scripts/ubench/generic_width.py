#!/usr/bin/env python3
"""forward time of the plain fp32 path (d_model != 128, csrc/savad_generic.h) beside the tuned d_model = 128 kernels"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_features, seeded_state_dict  # noqa: E402

for D in (64, 128, 256):
    m = SelfAttentiveVAD(80, 3, D, 0.5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1, d_model=D).items()})
    m = m.cuda().eval()
    for B, T in ((32, 800), (1000, 7), (4, 100)):
        x = torch.from_numpy(seeded_features(2, (B, T, 80))).cuda()
        with torch.no_grad():
            for _ in range(5):
                m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                m(x)
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        gflop = B * T * (2 * 80 * D + 3 * (2 * 4 * D * D + 2 * 8 * D * D + 4 * T * D)) / 1e9
        print(f"d_model={D:4d} [{B},{T},80]: {ms:8.3f} ms  {gflop / ms:8.1f} TFLOP/s")
