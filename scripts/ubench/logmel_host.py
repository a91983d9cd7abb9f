import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from voice_activity_detection_amd.features import log_mel
for secs in (10.0, 60.0):
    audio = torch.from_numpy(np.random.default_rng(0).normal(0, 0.1, int(16000 * secs)).astype(np.float32)).cuda()
    for _ in range(20): log_mel(audio, "cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): log_mel(audio, "cuda")
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{secs:5.0f} s: host issue {1e6*(t1-t0)/200:6.1f} us per call, incl. drain {1e6*(t2-t0)/200:6.1f} us")
