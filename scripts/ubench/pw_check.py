#!/usr/bin/env python3
"""persistent bf16 attention (row_mode 5) against the first-generation kernel (row_mode 1): bits + kernel times.
usage: pw_check.py B T [B T ...]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_features, seeded_state_dict  # noqa: E402

m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval()
m.precision = "bf16"
args = [int(a) for a in sys.argv[1:]]
for B, T in zip(args[0::2], args[1::2]):
    x = torch.from_numpy(seeded_features(B * 1000 + T, (B, T, 80))).cuda()
    out = {}
    for mode in (1, 5):
        m.row_mode = mode
        with torch.no_grad():
            y = m(features=x)
            torch.cuda.synchronize()
            m.set_profiling(5, skip=3)
            for _ in range(8):
                m(features=x)
            torch.cuda.synchronize()
            kt = m.kernel_times()
            m.set_profiling(0)
        out[mode] = (y.clone(), kt)
    y1, y5 = out[1][0], out[5][0]
    same = torch.equal(y1, y5)
    diff = float((y1 - y5).abs().max())
    att1 = [round(t * 1e3, 1) for n, t in out[1][1] if n == "attention_bf16"]
    att5 = [round(t * 1e3, 1) for n, t in out[5][1] if n == "attention_bf16"]
    print(f"B={B} T={T}: bit-equal={same} max|d|={diff:.3e} finite={bool(torch.isfinite(y5).all())} attention us old={att1} new={att5}", flush=True)
