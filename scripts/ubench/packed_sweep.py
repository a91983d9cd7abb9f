"""T <= 32 schedules against each other: row_mode 4 (single launch, one workgroup per packed tile) vs 1 (per-layer
launches, N-split) vs 2 (M-split 128-row tiles + packed attention launch).  usage: packed_sweep.py [T]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict
T = int(sys.argv[1]) if len(sys.argv) > 1 else 7
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval()
def bench(B, mode):
    m.row_mode = mode
    x = torch.randn(B, T, 80, device="cuda")
    with torch.no_grad():
        for _ in range(5): m(x)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): m(x)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
    return best
G = 32 // T
print(f"T={T} ({G} sequences per tile); ms per forward")
print(f"{'B':>7} {'tiles':>6} {'mode4':>8} {'mode1':>8} {'mode2':>8} {'auto':>8}")
for B in [int(a) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else "256,1000,1024,1100,1500,2048,3000,4096,8192,16384".split(","))]:
    r = [bench(B, md) for md in (4, 1, 2, 0)]
    print(f"{B:7d} {(B + G - 1) // G:6d} " + " ".join(f"{v:8.4f}" for v in r))
