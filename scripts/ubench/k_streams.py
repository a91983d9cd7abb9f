"""Experiment: batch split over K HIP streams (tail of one chunk's kernel overlaps the other chunks' kernels)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict

B = int(sys.argv[1]); PREC = sys.argv[2]
sd = {k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()}
def mk():
    m = SelfAttentiveVAD(80, 3, 128, 0.5); m.load_state_dict(sd); m = m.cuda().eval(); m.precision = PREC; return m
KMAX = 6
ms = [mk() for _ in range(KMAX)]
ss = [torch.cuda.Stream() for _ in range(KMAX)]
x = torch.randn(B, 800, 80, device="cuda")
def run(K):
    if K == 1:
        return ms[0](x)
    cur = torch.cuda.current_stream()
    outs = []
    bounds = [B * i // K for i in range(K + 1)]
    for i in range(K):
        ss[i].wait_stream(cur)
        with torch.cuda.stream(ss[i]):
            outs.append(ms[i](x[bounds[i]:bounds[i + 1]]))
    for i in range(K): cur.wait_stream(ss[i])
    return outs
def bench(f, n=40):
    for _ in range(8): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for K in (1, 2, 3, 4, 5, 6):
    print(B, PREC, "K =", K, round(bench(lambda: run(K)), 4))
