"""Experiment: consecutive INDEPENDENT forwards alternating over two HIP streams (two handles, two workspaces) against one
stream -- do the idle CU slots of one forward (32 idle CUs + the input stage's 56 at [32,800,80] fp32; the last partial round of
every bf16 launch) take the other forward's workgroups?   pipelined_streams.py B T precision"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features

B, T, PREC = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
MODE = int(sys.argv[4]) if len(sys.argv) > 4 else 0
sd = {k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()}
def mk():
    m = SelfAttentiveVAD(80, 3, 128, 0.5)
    m.load_state_dict(sd)
    m = m.cuda().eval(); m.precision = PREC; m.row_mode = MODE
    return m
NS = 6
ms = [mk() for _ in range(NS)]
x = torch.from_numpy(seeded_features(1, (B, T, 80))).cuda()
if PREC == "bf16": x = x.to(torch.bfloat16)
ss = [torch.cuda.Stream() for _ in range(NS)]
def run(nstreams, n):
    with torch.no_grad():
        for i in range(n):
            k = i % nstreams
            with torch.cuda.stream(ss[k]):
                ms[k](x)
    torch.cuda.synchronize()
for ns in (1, 2, 3, 4, 6):
    run(ns, 300)
    t = time.perf_counter(); run(ns, 400); dt = (time.perf_counter() - t) / 400
    flop = B * T * (2 * 80 * 128 + 3 * (2 * 4 * 128 * 128 + 2 * 8 * 128 * 128 + 4 * T * 128))
    peak = 2500.0 if PREC == "bf16" else 157.3
    print(f"[{B},{T}] {PREC} row_mode {MODE} {ns} stream(s): {dt*1e6:8.1f} us per forward  {flop/dt/1e12:6.1f} TF = {flop/dt/1e12/peak:.2f} of the peak", flush=True)
