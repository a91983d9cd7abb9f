#!/usr/bin/env python3
"""same-run A/B of two library builds on one shape: ab_lib.py <libA.so> <libB.so> precision B T [rounds]"""
import os, subprocess, sys
if sys.argv[1] == "--one":
    sys.path.insert(0, os.getcwd())
    import torch
    from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features
    prec, B, T = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    m = SelfAttentiveVAD(80, 3, 128, 0.5)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
    m = m.cuda().eval(); m.precision = prec
    x = torch.from_numpy(seeded_features(1, (B, T, 80))).cuda()
    if prec == "bf16": x = x.to(torch.bfloat16)
    with torch.no_grad():
        for _ in range(300): m(x)
        torch.cuda.synchronize()
        best = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200): m(x)
            e1.record(); torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) / 200)
    print(f"{os.path.basename(os.environ['SAVAD_LIB']):28s} {prec} [{B},{T}]: median {sorted(best)[2]*1e3:7.1f} us  min {min(best)*1e3:7.1f} us", flush=True)
else:
    a, b, rest = sys.argv[1], sys.argv[2], sys.argv[3:6]
    for _ in range(int(sys.argv[6]) if len(sys.argv) > 6 else 2):
        for lib in (a, b):
            subprocess.run([sys.executable, __file__, "--one", *rest], env=dict(os.environ, SAVAD_LIB=os.path.abspath(lib)))
