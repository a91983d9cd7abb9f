"""Reference-style predict_probabilities (7-frame windows, boosted) on 10 min of features: chunk size sweep."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from voice_activity_detection_amd import SelfAttentiveVAD, VADFromScratchPredictor, seeded_state_dict
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval()
feat = torch.from_numpy(np.random.default_rng(0).uniform(-13.8, 4.2, (60001, 80)).astype(np.float32)).cuda()
ref = None
for chunk in (1000, 4096, 16384, 65536):
    p = VADFromScratchPredictor(m, "cuda", chunk_size=chunk)
    for _ in range(2): probs, mean = p.predict_probabilities_device(feat)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): probs, mean = p.predict_probabilities_device(feat)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    if ref is None: ref = probs.clone()
    print(f"chunk {chunk:6d}: {dt*1e3:8.3f} ms for 600 s of audio (RTF {dt/600:.2e}); max |dp| vs chunk 1000 = {(probs-ref).abs().max().item():.2e}")
