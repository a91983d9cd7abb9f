#!/usr/bin/env python3
"""attention-stage time of one libsavad variant at [B, T] bf16: pw_time.py <lib.so> B T [row_mode (0 = automatic, default 5)] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SAVAD_LIB"] = os.path.abspath(sys.argv[1])
import torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features
B, T = int(sys.argv[2]), int(sys.argv[3])
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.precision = "bf16"; m.row_mode = int(sys.argv[4]) if len(sys.argv) > 4 else 5
N = int(sys.argv[5]) if len(sys.argv) > 5 else 0   # N > 0: only N plain forwards (counter passes serialise every dispatch)
x = torch.from_numpy(seeded_features(1, (B, T, 80))).cuda().to(torch.bfloat16)
if N:
    with torch.no_grad():
        for _ in range(N): m(x)
    torch.cuda.synchronize()
    sys.exit(0)
with torch.no_grad():
    for _ in range(30): m(x)
    torch.cuda.synchronize()
    m.set_profiling(20, skip=200)
    for _ in range(220): m(x)
    torch.cuda.synchronize()
kt = m.kernel_times()
att = [t * 1e3 for n, t in kt if n == "attention_bf16"]
print(f"{os.path.basename(sys.argv[1]):36s} B={B} T={T}: attention {sum(att)/len(att):7.1f} us   forward {sum(t for _, t in kt)*1e3:7.1f} us")
