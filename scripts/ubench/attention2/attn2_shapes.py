"""bf16 row_mode 6 vs row_mode 1 on a list of shapes, each in its own process (a fault kills only that one)."""
import os, subprocess, sys
code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, seeded_features
B, T = int(sys.argv[1]), int(sys.argv[2])
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval(); m.precision = "bf16"
x = torch.from_numpy(seeded_features(B + T, (B, T, 80))).cuda()
with torch.no_grad():
    m.row_mode = 1; y1 = m(x).clone()
    m.row_mode = 6; y6 = m(x).clone()
torch.cuda.synchronize()
print(B, T, "max|d|", float((y1 - y6).abs().max()), "finite", bool(torch.isfinite(y6).all()))
'''
for shape in sys.argv[1:]:
    B, T = shape.split("x")
    r = subprocess.run([sys.executable, "-c", code, B, T], capture_output=True, text=True, timeout=300)
    out = [l for l in (r.stdout + r.stderr).splitlines() if "amdgpu.ids" not in l]
    print(shape, "rc", r.returncode, "|", " ".join(out[-2:])[:300], flush=True)
