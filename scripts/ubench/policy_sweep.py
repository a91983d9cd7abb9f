"""Is the automatic schedule (row_mode 0, splits 0) the best forced one?  Times every row_mode per shape."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.precision = prec
def bench(x, n=30):
    with torch.no_grad():
        for _ in range(5): m(features=x)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): m(features=x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
shapes = []
for T in (7, 20, 40, 64, 100, 200, 400, 800, 1600):
    for rows in (2000, 8000, 14000, 20000, 25600, 102400):
        B = max(1, rows // T)
        shapes.append((B, T))
worst = 1.0
for B, T in shapes:
    x = torch.randn(B, T, 80, device="cuda")
    if prec == "bf16": x = x.to(torch.bfloat16)
    res = {}
    bench(x)  # first-touch effects (allocation, clocks) land here, not on the first mode timed
    for mode in (4, 3, 2, 1, 0):
        m.row_mode = mode
        res[mode] = bench(x)
    best = min(res.values())
    ratio = res[0] / best
    worst = max(worst, ratio)
    print(f"B={B:6d} T={T:5d}  auto {res[0]:.4f}  N {res[1]:.4f}  M {res[2]:.4f}  fused {res[3]:.4f}  one {res[4]:.4f}   auto/best = {ratio:.3f}" + ("  <<<" if ratio > 1.05 else ""), flush=True)
print("worst auto/best", round(worst, 3))
