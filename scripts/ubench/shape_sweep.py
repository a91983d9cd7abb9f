"""fp32 forward over mid-size shapes: ms, share of the fp32 MFMA spec, per-launch times."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval()
shapes = [(1, 800), (4, 800), (8, 800), (16, 800), (20, 800), (24, 800), (64, 100), (200, 100), (500, 50), (2, 3000), (112, 800)]
for B, T in shapes:
    x = torch.randn(B, T, 80, device="cuda")
    with torch.no_grad():
        for _ in range(10): m(x)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): m(x)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        m.set_profiling(5, skip=50)
        for _ in range(55): m(x)
        torch.cuda.synchronize()
        kt = m.kernel_times(); m.set_profiling(0)
    fl = (2 * (80 * 128 + 3 * (4 * 128 * 128 + 2 * 128 * 512)) + 3 * 4 * T * 128 + 512) * B * T
    print(f"[{B:4d},{T:5d}] {best:8.4f} ms  {fl / best / 1e9 / 157.3 * 100:5.1f} % of spec   " + " ".join(f"{n}={v*1e3:.0f}" for n, v in kt))
