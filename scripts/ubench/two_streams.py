"""Experiment: does splitting the [32,800,80] batch over two HIP streams fill the idle SIMDs?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict

sd = {k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()}
def mk():
    m = SelfAttentiveVAD(80, 3, 128, 0.5)
    m.load_state_dict(sd)
    return m.cuda().eval()
m0, m1, m2 = mk(), mk(), mk()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
PREC = sys.argv[2] if len(sys.argv) > 2 else "fp32"
for mm in (m0, m1, m2): mm.precision = PREC
x = torch.randn(B, 800, 80, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run_single():
    return m0(x)
def run_split(k):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        a = m1(x[:k])
    with torch.cuda.stream(s2):
        b = m2(x[k:])
    cur.wait_stream(s1); cur.wait_stream(s2)
    return a, b
def bench(f, n=50):
    for _ in range(10): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("single", round(bench(run_single), 4))
for k in [B // 2, B // 2 + B // 8, B // 4]:
    print("split", k, B - k, round(bench(lambda: run_split(k)), 4))
