"""fp32 (exact-fp32 MFMA kernels, automatic schedule) against fp32s (split-bf16 fused launches) over small and mid-size batches of long
sequences: where does fp32s stop paying?  python scripts/ubench/f32s_vs_f32_sweep.py"""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_features, seeded_state_dict  # noqa: E402

m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.to("cuda").eval()


def t(x, prec):
    m.precision = prec
    with torch.no_grad():
        for _ in range(5):
            m(features=x)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                m(features=x)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
    return best * 1e3


for T in (33, 64, 100, 200, 400, 800, 1600, 3200):
    for B in (1, 2, 4, 8, 12, 16, 24, 32, 64):
        if B * T > 64 * 1600:
            continue
        x = torch.from_numpy(seeded_features(5, (B, T, 80))).to("cuda")
        a, b = t(x, "fp32"), t(x, "fp32s")
        QB = (T + 31) // 32
        print(f"T={T:5d} B={B:3d} blocks={B * QB:5d} groups={B * ((QB + 3) // 4):5d}  fp32 {a:8.1f} us  fp32s {b:8.1f} us  ratio {a / b:5.2f}", flush=True)
