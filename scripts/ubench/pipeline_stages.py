import os, sys, time, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from voice_activity_detection_amd import SelfAttentiveVAD, seeded_state_dict, _lib
from voice_activity_detection_amd.predictor import VADFromScratchPredictor
m = SelfAttentiveVAD(80, 3, 128, 0.5)
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(1234).items()})
m = m.cuda().eval()
p = VADFromScratchPredictor(m, torch.device("cuda"))
lib = _lib.load()
feat = torch.randn(1001, 80, device="cuda")
N, F = feat.shape
half, jump, W = p.context_window_half_frames, p.context_window_jump_frames, p.context_window_frames
n_items = N - 2 * half
dev = torch.device("cuda")
def stage(name, fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"  {name:34s} host issue {1e6*(t1-t0)/n:7.1f} us   incl. drain {1e6*(t2-t0)/n:7.1f} us")
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
logp = torch.empty((n_items, W, 2), dtype=torch.float32, device=dev)
pos = torch.empty((n_items, W), dtype=torch.int64, device=dev)
win = torch.empty((n_items, W, F), dtype=torch.float32, device=dev)
probs = torch.empty((N, W), dtype=torch.float32, device=dev); mean = torch.empty((N,), dtype=torch.float32, device=dev)
boosted = torch.empty((N, W, 2), dtype=torch.float32, device=dev)
stage("torch.as_tensor.to.contiguous", lambda: torch.as_tensor(feat, dtype=torch.float32).to(dev).contiguous())
stage("model.eval()", lambda: m.eval())
stage("6 x torch.empty", lambda: [torch.empty((n_items, W, 2), dtype=torch.float32, device=dev) for _ in range(6)])
stage("savad_gather_windows", lambda: lib.savad_gather_windows(ctypes.c_void_p(feat.data_ptr()), N, F, half, jump, 0, n_items, ctypes.c_void_p(win.data_ptr()), ctypes.c_void_p(pos.data_ptr()), stream))
stage("model(features=win, out=logp)", lambda: m(features=win, out=logp))
stage("savad_boost", lambda: lib.savad_boost(ctypes.c_void_p(logp.data_ptr()), ctypes.c_void_p(pos.data_ptr()), n_items, N, W, ctypes.c_void_p(boosted.data_ptr()), ctypes.c_void_p(probs.data_ptr()), ctypes.c_void_p(mean.data_ptr()), stream))
stage("predict_probabilities_device", lambda: p.predict_probabilities_device(feat))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): p.predict_probabilities_device(feat)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
