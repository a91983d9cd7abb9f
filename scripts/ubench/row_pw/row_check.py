#!/usr/bin/env python3
"""persistent 64-row bf16 row stage (experiment library, row_mode 7) against the 32-row one (row_mode 5 = same attention kernel when forced): bits + kernel times.
usage: row64_check.py B T [B T ...]"""
import os
import sys
from pathlib import Path

os.environ.setdefault("SAVAD_LIB", str(Path(__file__).resolve().parent / "libsavad_rowpw.so"))   # bash scripts/ubench/row_pw/build.sh

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_features, seeded_state_dict  # noqa: E402

m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval()
m.precision = "bf16"
MODE = 7
if sys.argv[1] == "--mode":
    MODE = int(sys.argv[2])
    del sys.argv[1:3]
args = [int(a) for a in sys.argv[1:]]
for B, T in zip(args[0::2], args[1::2]):
    x = torch.from_numpy(seeded_features(B * 1000 + T, (B, T, 80))).cuda()
    out = {}
    for mode in (1, MODE):
        m.row_mode = mode
        with torch.no_grad():
            y = m(features=x)
            torch.cuda.synchronize()
            big = B * T >= 100000
            m.set_profiling(10, skip=100 if big else 3)
            for _ in range(110 if big else 13):
                m(features=x)
            torch.cuda.synchronize()
            kt = m.kernel_times()
            m.set_profiling(0)
        out[mode] = (y.clone(), kt)
    ya, yb = out[1][0], out[MODE][0]
    def avg(kt, name):
        v = [t * 1e3 for n, t in kt if n == name]
        return sum(v) / max(len(v), 1)
    print(f"B={B} T={T}: bit-equal={torch.equal(ya, yb)} max|d|={float((ya - yb).abs().max()):.3e} "
          f"row us 32-row={avg(out[1][1], 'row_bf16'):.1f} 64-row={avg(out[MODE][1], 'row_bf16'):.1f}  "
          f"last {avg(out[1][1], 'row_last_bf16'):.1f} / {avg(out[MODE][1], 'row_last_bf16'):.1f}  "
          f"forward {sum(t for _, t in out[1][1]) * 1e3:.1f} / {sum(t for _, t in out[MODE][1]) * 1e3:.1f}", flush=True)
