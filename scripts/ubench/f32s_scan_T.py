"""fp32s fused launch: time per workgroup as a function of the number of key tiles (B small: one round of workgroups) -> the cost of a
key tile and of the row chain.  python scripts/ubench/f32s_scan_T.py"""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_features, seeded_state_dict  # noqa: E402

m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.to("cuda").eval()
m.precision = sys.argv[1] if len(sys.argv) > 1 else "fp32s"
for T in (64, 128, 256, 512, 800, 1024, 1600, 3200):
    B = 8 if T >= 256 else 32
    x = torch.from_numpy(seeded_features(5, (B, T, 80))).to("cuda")
    with torch.no_grad():
        for _ in range(5):
            m(features=x)
        torch.cuda.synchronize()
        m.set_profiling(8, 4)
        for _ in range(12):
            m(features=x)
        torch.cuda.synchronize()
        kt = m.kernel_times()
        m.set_profiling(0)
    QB = (T + 31) // 32
    print(f"T={T:5d} B={B} QB={QB:4d} groups/seq={(QB + 3) // 4}: " + ", ".join(f"{k} {v * 1e3:.1f}us" for k, v in kt), flush=True)
