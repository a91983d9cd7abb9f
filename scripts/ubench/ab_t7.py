"""same-box A/B of libraries at the 7-frame-window shapes (fp32s): python scripts/ubench/ab_t7.py lib1.so lib2.so ...  (each in its own process)"""
import os, subprocess, sys
CHILD = r'''
import sys, torch
sys.path.insert(0, ".")
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_features, seeded_state_dict
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.to("cuda").eval(); m.precision = "fp32s"
out = []
for B in (1000, 16384, 65536):
    x = torch.from_numpy(seeded_features(5, (B, 7, 80))).to("cuda")
    with torch.no_grad():
        for _ in range(5): m(features=x)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): m(features=x)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
    out.append(f"[{B},7] {best*1e3:.1f} us")
print(sys.argv[1].split("/")[-1], "  ".join(out))
'''
for rep in range(2):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", CHILD, lib], env=dict(os.environ, SAVAD_LIB=os.path.abspath(lib)), stderr=subprocess.DEVNULL)
