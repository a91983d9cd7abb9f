import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from voice_activity_detection_amd.features import log_mel
for secs in (10.0, 3600.0):
    audio = torch.from_numpy(np.random.default_rng(0).normal(0, 0.1, int(16000 * secs)).astype(np.float32)).cuda()
    n = 200 if secs < 100 else 20
    for _ in range(10): log_mel(audio, "cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): log_mel(audio, "cuda")
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(os.path.basename(os.environ.get("SAVAD_LIB", "default")), f"{secs:6.0f} s: {1e6*(t2-t0)/n:8.1f} us per call")
