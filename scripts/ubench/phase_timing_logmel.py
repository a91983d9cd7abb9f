"""Phase cycles of logmel_fft_kernel (build: -DSAVAD_TIMING -> scripts/ubench/libsavad_timing.so); wave 0 of WG 0, summed over its tiles."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
os.environ["SAVAD_LIB"] = os.path.abspath("scripts/ubench/libsavad_timing.so")
import numpy as np, torch
from voice_activity_detection_amd import _lib
from voice_activity_detection_amd.features import log_mel
seconds = int(sys.argv[1]) if len(sys.argv) > 1 else 3600
y = torch.from_numpy(np.random.default_rng(0).standard_normal(16000 * seconds, dtype=np.float32) * 0.1).cuda()
for _ in range(3): log_mel(y)
torch.cuda.synchronize()
lib = _lib.load()
lib.savad_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
buf = (ctypes.c_longlong * 64)()
lib.savad_debug_stamps(buf, 64)
t = list(buf[24:32])
names = ["prologue (tables, first stage, first step 1)", "exchange write (waits for step 1's MFMAs)", "barrier 1", "DMA issue + step 3 + power + mel + partial write",
         "vmcnt(0): next stage landed", "barrier 2", "-", "next tile's step 1 + sums + log + store"]
tiles = (1 + 16000 * seconds // 160 + 31) // 32
per = -(-tiles // 256)
print(f"{seconds} s of audio: {tiles} tiles, ~{per} per workgroup; clock ticks of wave 0 / WG 0 (total, per tile)")
for i in (0, 1, 2, 3, 4, 5, 7): print(f"  {names[i]:44s} {t[i]:9d} {t[i]/per:9.0f}")
print("  sum", sum(t), sum(t) / per)
