"""Experiment: run the batch as C sequential chunks (working set per chunk fits the 256 MB Infinity Cache?)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict
B = int(sys.argv[1]); PREC = sys.argv[2]
sd = {k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()}
m = SelfAttentiveVAD(80, 3, 128, 0.5); m.load_state_dict(sd); m = m.cuda().eval(); m.precision = PREC
x = torch.randn(B, 800, 80, device="cuda")
def run(C):
    bounds = [B * i // C for i in range(C + 1)]
    return [m(x[bounds[i]:bounds[i + 1]]) for i in range(C)]
def bench(f, n=40):
    for _ in range(8): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for C in (1, 2, 3, 4, 8):
    print(B, PREC, "chunks =", C, round(bench(lambda: run(C)), 4))
