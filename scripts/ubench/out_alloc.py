import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.precision = sys.argv[1]
B, T = int(sys.argv[2]), int(sys.argv[3])
x = torch.from_numpy(seeded_features(1, (B, T, 80))).cuda()
if sys.argv[1] == "bf16": x = x.to(torch.bfloat16)
out = torch.empty((B, T, 2), device="cuda")
def t(f, n=400):
    for _ in range(300): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
with torch.no_grad():
    for rep in range(2):
        print("alloc per call", round(t(lambda: m(x)), 1), "us;  out= preallocated", round(t(lambda: m(x, out=out)), 1), "us", flush=True)
