"""same-box A/B of libraries on the 10 s clip through the product's graph mode (windowed single launch): python scripts/ubench/ab_clip.py lib1.so lib2.so"""
import os, subprocess, sys
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
from voice_activity_detection_amd import SelfAttentiveVAD, VADFromScratchPredictor, seeded_state_dict
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.to("cuda").eval()
audio = torch.from_numpy(np.random.default_rng(7).normal(0.0, 0.1, 160000).astype(np.float32)).cuda()
out = []
for prec in ("fp32s", "bf16"):
    m.precision = prec
    p = VADFromScratchPredictor(m, "cuda", graph=True)
    for _ in range(20): p.predict_audio_device(audio)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): p.predict_audio_device(audio)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100)
    out.append(f"{prec} {best*1e3:.1f} us")
print(sys.argv[1].split("/")[-1], "  ".join(out))
'''
for rep in range(2):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", CHILD, lib], env=dict(os.environ, SAVAD_LIB=os.path.abspath(lib)), stderr=subprocess.DEVNULL)
