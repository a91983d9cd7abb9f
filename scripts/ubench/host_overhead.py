#!/usr/bin/env python3
"""Host-side cost of one `model(features=x)` call (no synchronisation inside the timed loop: what the CPU spends before it can
issue the next call) on the shapes the reference runs: windows of 7 frames, vad/predictor.py:221-224.
    python scripts/ubench/host_overhead.py [B T ...]      (default: 1000 7  984 7  32 800)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features

args = [int(a) for a in sys.argv[1:]] or [1000, 7, 984, 7, 32, 800]
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval()
for B, T in zip(args[0::2], args[1::2]):
    x = torch.from_numpy(seeded_features(1, (B, T, 80))).cuda()
    out = torch.empty((B, T, 2), device="cuda")
    with torch.no_grad():
        for _ in range(50):
            m(features=x, out=out)
        torch.cuda.synchronize()
        res = {}
        for label, fn in (("model(features=x, out=out)", lambda: m(features=x, out=out)), ("model(features=x)", lambda: m(features=x)),
                          ("_param_versions()", m._param_versions)):
            # short bursts so that the launch queue never fills (a full queue would make the host wait for the GPU)
            ts = []
            for _ in range(40):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(8):
                    fn()
                ts.append((time.perf_counter() - t0) / 8 * 1e6)
            ts.sort()
            res[label] = (ts[len(ts) // 2], ts[0])
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            m(features=x, out=out)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / n * 1e6
    print(f"[{B},{T},80]: " + "; ".join(f"{k}: {v[0]:.1f} us median ({v[1]:.1f} min)" for k, v in res.items()) + f"; back-to-back forwards incl. GPU: {per:.1f} us each", flush=True)
