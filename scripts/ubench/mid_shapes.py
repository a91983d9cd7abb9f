#!/usr/bin/env python3
"""per-kernel times of mid-size fp32 shapes under every launch schedule: mid_shapes.py B T [B T ...]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval()
a = [int(v) for v in sys.argv[1:]]
for B, T in zip(a[0::2], a[1::2]):
    x = torch.from_numpy(seeded_features(1, (B, T, 80))).cuda()
    flop = B * T * (2 * 80 * 128 + 3 * (2 * 4 * 128 * 128 + 2 * 8 * 128 * 128 + 4 * T * 128))
    for mode, splits in ((0, 0), (1, 0), (1, 1), (1, 2), (1, 4), (2, 0), (3, 0)):
        m.row_mode, m.attention_splits = mode, splits
        with torch.no_grad():
            for _ in range(30): m(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200): m(x)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 200
            m.set_profiling(10, skip=50)
            for _ in range(60): m(x)
            torch.cuda.synchronize()
        kt = m.kernel_times(); m.set_profiling(0)
        agg = {}
        for n, t in kt: agg.setdefault(n, []).append(t * 1e3)
        print(f"[{B},{T}] mode {mode} splits {splits}: {ms*1e3:7.1f} us  {flop/ms/1e9:6.1f} TF  ", {n: round(sum(v)/len(v), 1) for n, v in agg.items()}, flush=True)
