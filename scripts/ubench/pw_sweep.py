#!/usr/bin/env python3
"""attention stage per layer, first-generation kernel (row_mode 1) against the persistent one (row_mode 5), over batch sizes and
lengths: where should automatic pick the persistent kernel (savad.hip, pw_pays)?   python scripts/ubench/pw_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.precision = "bf16"
shapes = [(96, 800), (128, 800), (160, 800), (192, 800), (224, 800), (256, 800), (320, 800), (512, 800), (256, 600), (344, 600), (512, 600),
          (256, 1000), (172, 1200), (128, 1600), (512, 400), (1024, 400), (64, 3200)]
for B, T in shapes:
    x = torch.from_numpy(seeded_features(1, (B, T, 80))).cuda().to(torch.bfloat16)
    res = {}
    for mode in (1, 5):
        m.row_mode = mode
        with torch.no_grad():
            for _ in range(20): m(x)
            torch.cuda.synchronize()
            m.set_profiling(10, skip=80)
            for _ in range(90): m(x)
            torch.cuda.synchronize()
        kt = m.kernel_times(); m.set_profiling(0)
        att = [t * 1e3 for n, t in kt if n == "attention_bf16"]
        res[mode] = (sum(att) / len(att), sum(t for _, t in kt) * 1e3)
    QB = (T + 31) // 32
    print(f"[{B:5d},{T:5d}] QB {QB:3d} full items per WG {B * (QB // 8) / 256:5.2f}: attention first-gen {res[1][0]:7.1f} us  persistent {res[5][0]:7.1f} us  "
          f"({100 * (res[5][0] / res[1][0] - 1):+5.1f} %)   forward {res[1][1]:7.1f} / {res[5][1]:7.1f} us", flush=True)
